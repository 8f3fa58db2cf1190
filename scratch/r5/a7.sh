cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for sc in living_room; do
  for ov in 0 1; do
    if [ $ov = 1 ]; then export RL_NO_OVERLAP=1; else unset RL_NO_OVERLAP; fi
    echo "== $sc RL_NO_OVERLAP=$RL_NO_OVERLAP"
    REPS=3 timeout 120 python scratch/ref_bench.py $sc 128
  done
done 2>&1 | tee gpurun_out/r5/a7_ab.txt
unset RL_NO_OVERLAP
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or cfg4 or bench_frames or randomized or small_scenes" 2>&1 | tail -8 | tee gpurun_out/r5/a7_tests.txt
