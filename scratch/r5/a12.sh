# the whole GPU suite + the default bench line on the tree with the overlapped evaluation pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5/a12_tests.txt
python bench.py > gpurun_out/r5/a12_bench.json 2> gpurun_out/r5/a12_bench.err; tail -c 300 gpurun_out/r5/a12_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/a12_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','reference_order_value','reference_order_in_flight_value','oracle_crc_match')})
for a in d['also']: print(a['workload'], round(a.get('ms_per_step',0),1), round(a.get('value',0),1), a.get('oracle_crc_match'), a.get('kernel_ms'))
PY
