# round 5, call 1: parity of traverse2 (two-level records) + same-box A/B against the one-level build on the 508 k-triangle scene
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "trace_batch or visible_batch or triangle_edge or small_scenes_through or non_finite or mixed_materials_parity or randomized" 2>&1 | tail -15 > gpurun_out/r5/a1_tests.txt
cat gpurun_out/r5/a1_tests.txt
( python scratch/variants.py run living_room 2 128; python scratch/variants.py run living_room 2 32 ) 2>&1 | tee gpurun_out/r5/a1_ab.txt
for lib in scratch/variants/libtwo.so scratch/variants/libone.so; do REPS=2 python scratch/ref_bench.py living_room 128 $lib; done 2>&1 | tee gpurun_out/r5/a1_ref.txt
