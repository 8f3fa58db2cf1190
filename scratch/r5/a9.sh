cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
echo "== timers lib, no overlap"; RL_NO_OVERLAP=1 timeout 60 python scratch/r5/shard_stats.py scratch/variants/libspectimers.so 8 1024 2>&1 | tail -12
echo "rc=$?"
echo "== timers lib, overlap"; timeout 60 python scratch/r5/shard_stats.py scratch/variants/libspectimers.so 8 1024 2>&1 | tail -12
echo "rc=$?"
echo "== default lib, overlap"; timeout 60 python scratch/r5/shard_stats.py rustlight_amd/lib/librustlight_amd.so 8 1024 2>&1 | tail -5
echo "rc=$?"
