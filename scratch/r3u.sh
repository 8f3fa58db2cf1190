# counters of k_stream_chain on the 508 k-triangle scene (reference-order streams, 16 spp)
O=gpurun_out/r3u; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/p$i -o p -- python $R/scratch/ref_bench.py living_room 16 > $R/$O/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:36]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "chain" in k: print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
find $O -name '*.csv' -size +2M -delete
