# dev: per-XCD item dispensers on the streaming fused kernel: time with / without, L2 hit rate, the GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/living_xcd; mkdir -p $O; cd $R
for x in 1 0; do
RL_XCD_DISPENSE=$x python bench.py --scene living_room --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null > $O/bench$x.json
python -c "
import json; d=json.loads(open('$O/bench$x.json').readline()); print('living xcd=$x', d['ms_per_step'], d['value'], d['distributed']['image_crc32'], d.get('oracle_crc_match'))"
RL_XCD_DISPENSE=$x python bench.py --scene living_room --tris 4000000 --steps 2 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null > $O/bench4m$x.json
python -c "
import json; d=json.loads(open('$O/bench4m$x.json').readline()); print('living4m xcd=$x', d['ms_per_step'], d['value'], d['distributed']['image_crc32'])"
done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
cd /tmp && export TMPDIR=/tmp
for x in 1; do
RL_XCD_DISPENSE=$x timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/p$x -o p -- python $R/bench.py --scene living_room --steps 1 --warmup 0 --no-cpu-baseline --no-also > $O/p$x.log 2>&1
python - <<PY
import csv,glob
c={}
for f in glob.glob('$O/p$x/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_path_fused' in r['Kernel_Name']: c[r['Counter_Name']]=c.get(r['Counter_Name'],0)+float(r['Counter_Value'])
print('xcd=$x', {k: '%.4g' % v for k,v in sorted(c.items())}, 'hit rate', c.get('TCC_HIT_sum',0)/max(1,c.get('TCC_HIT_sum',0)+c.get('TCC_MISS_sum',0)))
PY
done
find $O -name '*.csv' -size +1M -delete
