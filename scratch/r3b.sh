# round 3: sweep of the chain pass (items per wave, waves per SIMD) + counters of k_stream_chain
O=gpurun_out/r3b; mkdir -p $O
for s in 3 4 5 6; do RL_ITEM_SHIFT=$s python scratch/ref_bench.py cbox 128; done 2>&1 | tee $O/shift.log
for v in cw6 cw8; do for s in 5 6; do RL_ITEM_SHIFT=$s python scratch/ref_bench.py cbox 128 scratch/variants/lib$v.so; done; done 2>&1 | tee $O/waves.log
python scratch/ref_bench.py cbox_medium 16 2>&1 | tee $O/other.log
RL_REF_SINGLE_PASS=1 python scratch/ref_bench.py cbox_medium 16 2>&1 | tee -a $O/other.log
python scratch/ref_bench.py living_room 16 2>&1 | tee -a $O/other.log
RL_REF_SINGLE_PASS=1 python scratch/ref_bench.py living_room 16 2>&1 | tee -a $O/other.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/p$i -o p -- python $R/scratch/ref_bench.py cbox 128 > $R/$O/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "chain" in k or "fused" in k: print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
find $O -name '*.csv' -size +2M -delete
