# round 6, GPU call 10: the randomized differential test on the final tree (two seeds + the tolerance-build arm)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g10; mkdir -p $O
{ echo "== parity_fuzz.py 900 s seed 606"; timeout 1100 python tests/parity_fuzz.py 900 606 2>&1 | tail -4
  echo "== parity_fuzz.py 600 s seed 2606 fast"; timeout 800 python tests/parity_fuzz.py 600 2606 fast 2>&1 | tail -4; } > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
