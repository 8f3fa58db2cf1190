# Round-5 evidence on the GPU box: counters (separate --pmc passes, kernel-trace only) for the four workloads bench.py may quote, kernel-trace stats of the default
# bench command and of the reference-order run, the bench lines.   usage: RL_COMMIT=<sha> bash scratch/r5/profile_round5.sh
# (the counter passes of the reference-order workload run the two passes back to back, RL_NO_OVERLAP=1: rocprofv3 serialises dispatches while it collects counters)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5prof; mkdir -p $O
cd $R
bash scratch/pmc_collect.sh r5pmc_cbox k_path_fused > $O/pmc_cbox.log 2>&1
RL_NO_OVERLAP=1 bash scratch/pmc_collect.sh r5pmc_reference k_stream_spec --stream-mode reference > $O/pmc_reference.log 2>&1
bash scratch/pmc_collect.sh r5pmc_living k_path_fused --scene living_room > $O/pmc_living.log 2>&1
bash scratch/pmc_collect.sh r5pmc_medium k_path_fused --scene cbox_medium > $O/pmc_medium.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o p -- python $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_reference -o p -- python $R/bench.py --no-cpu-baseline --no-also --stream-mode reference --steps 3 --warmup 1 > $O/stats_reference.log 2>&1
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 2 > $O/bench_driver_style.json 2> /dev/null
python bench.py --stream-mode reference --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_reference.json 2>/dev/null
for t in cbox reference living medium; do cp gpurun_out/r5pmc_$t/pmc_summary.json $O/pmc_$t.json; done
find $O -name '*kernel_stats.csv' | head
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*agent_info.csv' -delete
tail -c 400 $O/bench_reference.json
