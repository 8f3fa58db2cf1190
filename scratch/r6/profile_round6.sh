# Round-6 evidence on the GPU box: counters (separate --pmc passes, kernel-trace only) for the SIX workloads bench.py may quote — the four of round 5 + the two slowest
# driver-timed chain passes (k_stream_chain on cbox + medium, k_stream_spec on 508 k triangles, reference-order streams) — kernel-trace stats of the headline, the default
# bench command and the reference-order run, the bench lines.   usage: RL_COMMIT=<sha> bash scratch/r6/profile_round6.sh
# (the counter passes of reference-order workloads run the two passes back to back, RL_NO_OVERLAP=1: rocprofv3 serialises dispatches while it collects counters)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6prof; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 2 > $O/bench_driver_style.json 2> /dev/null
python bench.py --stream-mode reference --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_reference.json 2>/dev/null
bash scratch/pmc_collect.sh r6pmc_cbox k_path_fused > $O/pmc_cbox.log 2>&1
RL_NO_OVERLAP=1 bash scratch/pmc_collect.sh r6pmc_reference k_stream_spec --stream-mode reference > $O/pmc_reference.log 2>&1
bash scratch/pmc_collect.sh r6pmc_living k_path_fused --scene living_room > $O/pmc_living.log 2>&1
bash scratch/pmc_collect.sh r6pmc_medium k_path_fused --scene cbox_medium > $O/pmc_medium.log 2>&1
RL_NO_OVERLAP=1 bash scratch/pmc_collect.sh r6pmc_medium_reference k_stream_chain --scene cbox_medium --stream-mode reference > $O/pmc_medium_reference.log 2>&1
RL_NO_OVERLAP=1 bash scratch/pmc_collect.sh r6pmc_living_reference k_stream_spec --scene living_room --stream-mode reference > $O/pmc_living_reference.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -o p -- python $R/bench.py --no-cpu-baseline --no-also --steps 5 > $O/stats_headline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o p -- python $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_reference -o p -- python $R/bench.py --no-cpu-baseline --no-also --stream-mode reference --steps 3 --warmup 1 > $O/stats_reference.log 2>&1
cd $R
for t in cbox reference living medium medium_reference living_reference; do cp gpurun_out/r6pmc_$t/pmc_summary.json $O/pmc_$t.json; done
find $O -name '*kernel_stats.csv' | head
find $O -name '*kernel_trace.csv' -size +2M -delete; find $O -name '*agent_info.csv' -delete
tail -c 600 $O/bench_default.json
