cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcl; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/a -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene living_room --spp 32 > $O/a.log 2>&1
python $R/scratch/pmc_sum.py $(find $O/a -name '*counter_collection.csv' | head -1) "rl::k_"
python - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/pmcl/a/*kernel_trace.csv')[0]
t = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    t[r["Kernel_Name"].split("(")[0][:40]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(t.items(), key=lambda x: -x[1])[:6]: print(f"{k:42s} {v:9.2f} ms")
PY
find $O -name '*.csv' -size +4M -delete
