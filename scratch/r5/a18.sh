cd $GRAFT_REPO_ROOT
for cfg in "64 3" "32 3" "128 3" "256 3" "64 2" "64 5" "64 8" "16 6"; do set -- $cfg; for sc in cbox cbox_medium; do echo -n "min $1 div $2: "; RL_EVAL_MIN=$1 RL_EVAL_DIV=$2 REPS=3 timeout 120 python scratch/ref_bench.py $sc 128 2>&1 | tail -1 | cut -c20-140; done; done
