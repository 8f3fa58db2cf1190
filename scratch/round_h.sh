# r01h: rocprofv3 kernel-trace stats of the default bench command + the default bench line on the final tree
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r1h; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -o p -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/fused.log 2>&1
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
head -3 $O/fused/p_kernel_stats.csv; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['cpu_baseline']['value'])"
