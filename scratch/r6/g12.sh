cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g12; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_default_head.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench_default_head.json").read().strip().splitlines()[-1])
print("HEAD bench:", round(d["value"],1), d["reference_order_value"], d["reference_order_worst_step_value"], d["roofline"]["pmc_current"], d["roofline"]["traffic"], [ (a["workload"], round(a["value"],1), a.get("value_min")) for a in d["also"] if "reference_order" in a["workload"]])
PY
{ echo "== parity_fuzz.py 1000 s seed 7606"; timeout 1200 python tests/parity_fuzz.py 1000 7606 2>&1 | tail -3
  echo "== parity_fuzz.py 800 s seed 8606"; timeout 1000 python tests/parity_fuzz.py 800 8606 2>&1 | tail -3; } > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
