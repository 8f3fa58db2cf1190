cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in noprio prio3; do for sc in cbox cbox_medium; do REPS=3 timeout 120 python scratch/ref_bench.py $sc 128 scratch/variants/lib$v.so 2>&1 | tail -1 | cut -c1-150; done; done; done
for v in noprio prio3; do REPS=2 timeout 120 python scratch/ref_bench.py living_room 128 scratch/variants/lib$v.so 2>&1 | tail -1 | cut -c1-150; done
