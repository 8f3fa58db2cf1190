# r01f evidence: full GPU suite, rocprofv3 kernel-trace stats, HBM + SQ PMC passes, bench lines (see profile_round.sh / pmc_fused.sh)
cd $GRAFT_REPO_ROOT && python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ROUND_TAG=r1f bash scratch/profile_round.sh 2>&1 | tail -30
mkdir -p gpurun_out/pmcf && bash scratch/pmc_fused.sh 2>&1 | tail -12
