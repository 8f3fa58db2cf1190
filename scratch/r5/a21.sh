cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for i in 1 2 3 4 5 6; do
  t0=$(date +%s.%N)
  python bench.py --no-cpu-baseline > gpurun_out/r5/a21_bench_$i.json 2> gpurun_out/r5/a21_$i.err
  t1=$(date +%s.%N)
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5/a21_bench_$i.json').read().strip().splitlines()[-1])
print($i, 'wall %.1f s' % ($t1 - $t0), round(d['value']), round(d['reference_order_value']), [ (round(a.get('ms_per_step',0),1), a.get('oracle_crc_match')) for a in d['also']])
PY
done
