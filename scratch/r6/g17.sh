cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g17; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
{ for seed in 9606 10606 11606; do echo "== parity_fuzz.py 850 s seed $seed"; timeout 1000 python tests/parity_fuzz.py 850 $seed 2>&1 | tail -2; done
  echo "== parity_fuzz.py 500 s seed 12606 fast"; timeout 700 python tests/parity_fuzz.py 500 12606 fast 2>&1 | tail -2; } > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
