# dev: the streaming kernels with the ring stack (trace.hip.h: RL_STACK_RING): time, CRC, instruction counts, the GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/living_ring; mkdir -p $O; cd $R
python bench.py --scene living_room --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null > $O/bench.json
python -c "
import json; d=json.loads(open('$O/bench.json').readline()); print('living', d['ms_per_step'], d['value'], d['distributed']['image_crc32'], d.get('oracle_crc_match'))"
python bench.py --scene living_room --tris 4000000 --steps 2 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null > $O/bench4m.json
python -c "
import json; d=json.loads(open('$O/bench4m.json').readline()); print('living4m', d['ms_per_step'], d['value'], d['distributed']['image_crc32'])"
python bench.py --scene living_room --stream-mode reference --steps 1 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null > $O/bench_ref.json
python -c "
import json; d=json.loads(open('$O/bench_ref.json').readline()); print('living ref', d['ms_per_step'], d['value'], d['distributed']['image_crc32'], d['roofline']['kernels'].keys() if 'kernels' in d['roofline'] else '')"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -o p -- python $R/bench.py --scene living_room --steps 1 --warmup 0 --no-cpu-baseline --no-also > $O/p1.log 2>&1
python - <<PY
import csv,glob
c={}
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_path_fused' in r['Kernel_Name']: c[r['Counter_Name']]=c.get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in sorted(c.items()): print(f'{k:24s} {v:.4g}')
PY
find $O -name '*.csv' -size +1M -delete
