cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python scratch/r4/shard_scaling.py 2>&1 | tee gpurun_out/r5/a8_shards.txt
for cfg in "8 1024" "1 128"; do python scratch/r5/shard_stats.py scratch/variants/libspectimers.so $cfg; done 2>&1 | tee gpurun_out/r5/a8_timers.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlapped" 2>&1 | tail -8 | tee gpurun_out/r5/a8_tests.txt
