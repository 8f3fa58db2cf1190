cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative_chain" 2>&1 | tail -12
for bw in 1 2 4; do echo "RL_SPEC_BLOCK_WAVES=$bw"; RL_SPEC_BLOCK_WAVES=$bw timeout 120 python scratch/r5/shard_stats.py rustlight_amd/lib/librustlight_amd.so 8 1024 2>&1 | tail -3; done
