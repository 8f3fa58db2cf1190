O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bvh4 or fast_numerics or small_scenes or randomized" 2>&1 | tail -5 | tee $O/tests.log
echo -n "default fast: "; NUMERICS=1 python scratch/variants.py one rustlight_amd/lib/librustlight_amd.so living_room 2 32 | tail -1
for v in vote11 vote21 vote12 w5 nosort; do echo -n "$v: "; NUMERICS=1 python scratch/variants.py one scratch/variants/lib$v.so living_room 2 32 | tail -1; done 2>&1 | tee $O/variants.log
