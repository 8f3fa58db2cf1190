cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5/a13_tests_full.txt 2>&1; grep -E "passed|failed" gpurun_out/r5/a13_tests_full.txt | tail -2 | tee gpurun_out/r5/a13_tests.txt
RL_COMMIT=$1 bash scratch/r5/profile_round5.sh 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5prof/stats_headline -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r5prof/stats_headline.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r5prof -name '*kernel_trace.csv' -size +2M -delete
