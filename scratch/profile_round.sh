# Collect the round's rocprofv3 evidence on the GPU box: kernel-trace stats for both pipelines, HBM PMC passes, bench lines.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND_TAG:-r1d}; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -o p -- $B --steps 3 --warmup 1 > $O/fused.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wavefront -o p -- $B --steps 3 --warmup 1 --pipeline wavefront > $O/wavefront.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/living -o p -- $B --steps 1 --warmup 1 --scene living_room > $O/living.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- $B --steps 1 --warmup 0 > $O/pmc_$c.log 2>&1
  python $R/scratch/pmc_sum.py $(find $O/pmc_$c -name '*counter_collection.csv' | head -1) k_path_fused
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcw_$c -o p -- $B --steps 1 --warmup 0 --pipeline wavefront > $O/pmcw_$c.log 2>&1
  python $R/scratch/pmc_sum.py $(find $O/pmcw_$c -name '*counter_collection.csv' | head -1) "rl::k_"
done
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python $R/bench.py --scene cbox_medium --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_medium.json 2>/dev/null
python $R/bench.py --scene living_room --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_living.json 2>/dev/null
find $O -name '*kernel_stats.csv' | head; tail -c 600 $O/bench_default.json
# keep the merge small: counter csvs are large
find $O -name '*counter_collection.csv' -size +8M -delete; find $O -name '*kernel_trace.csv' -size +8M -delete
