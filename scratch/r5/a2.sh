cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python scratch/variants.py run living_room 2 32 2>&1 | grep -v stats | tee gpurun_out/r5/a2_ab.txt
for v in twostats onestats; do echo "== $v"; python scratch/variants.py one scratch/variants/lib$v.so living_room 1 4 2>&1 | tail -4; done 2>&1 | tee gpurun_out/r5/a2_stats.txt
