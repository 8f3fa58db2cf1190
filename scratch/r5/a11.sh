cd $GRAFT_REPO_ROOT
for seed in 1 2 3 4; do timeout 150 python scratch/r5/stress_overlap.py $seed 25 2>&1 | tail -26; echo "rc=$?"; done
