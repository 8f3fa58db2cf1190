# final check of the committed tree on a fresh box: build check, smoke, the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5/a17_tests_full.txt 2>&1; grep -E "passed|failed" gpurun_out/r5/a17_tests_full.txt | tail -2
python bench.py > gpurun_out/r5/a17_bench.json 2> gpurun_out/r5/a17_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/a17_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','reference_order_value','oracle_crc_match')}, {k:d['roofline'].get(k) for k in ('frac','valu_issue_busy','lane_utilisation','valu_lane_slots_used','pmc_current')})
for a in d['also']: print(a['workload'], round(a.get('ms_per_step',0),1), round(a.get('value',0),1), a.get('oracle_crc_match'), a.get('kernel_ms'), (a.get('utilisation') or {}).get('valu_lane_slots_used'))
PY
