cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 200 python scratch/r4/shard_scaling.py > /dev/null 2>&1; echo "scaling rc=$?"
for cfg in "8 1024" "1 128"; do timeout 60 python scratch/r5/shard_stats.py scratch/variants/libspectimers.so $cfg 2>&1 | tail -8; echo "rc=$?"; done
