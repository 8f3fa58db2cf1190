cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for i in 1 2; do python scratch/variants.py run cbox 2 128; done 2>&1 | tee gpurun_out/r5/a5_ab.txt
VW=1080 VH=1080 python scratch/variants.py run cbox 2 128 2>&1 | tee -a gpurun_out/r5/a5_ab.txt
python scratch/variants.py run cbox_medium 2 32 2>&1 | tee -a gpurun_out/r5/a5_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cbox_render_parity or integrator_options or medium_parity or mixed_materials_parity or phong or emitters_parity or randomized or bench_frames" 2>&1 | tail -5 | tee gpurun_out/r5/a5_tests.txt
