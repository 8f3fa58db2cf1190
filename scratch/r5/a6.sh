# the evaluation pass overlapped with the chain pass: parity tests, then same-box A/B (RL_NO_OVERLAP=1 = back to back)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cbox_render_parity or two_pass_equals or speculative_chain or cfg1_reference" 2>&1 | tail -8 | tee gpurun_out/r5/a6_tests.txt
for sc in cbox living_room cbox_medium; do
  for ov in 0 1; do
    if [ $ov = 1 ]; then export RL_NO_OVERLAP=1; else unset RL_NO_OVERLAP; fi
    echo "== $sc RL_NO_OVERLAP=$RL_NO_OVERLAP"
    REPS=3 timeout 300 python scratch/ref_bench.py $sc 128
  done
done 2>&1 | tee gpurun_out/r5/a6_ab.txt
unset RL_NO_OVERLAP
python scratch/variants.py one rustlight_amd/lib/librustlight_amd.so cbox 2 128 2>&1 | tee -a gpurun_out/r5/a6_ab.txt
