cd $GRAFT_REPO_ROOT
for sp in 1 4 8 16; do for sc in cbox cbox_medium; do echo "RL_EVAL_SPLIT=$sp"; RL_EVAL_SPLIT=$sp REPS=3 timeout 120 python scratch/ref_bench.py $sc 128 2>&1 | tail -1; done; done
REPS=3 timeout 120 python scratch/ref_bench.py living_room 128 2>&1 | tail -1
