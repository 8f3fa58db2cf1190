# dev: the streaming fused kernel after the conditional / packed overflow pushes: time, CRC, store / load instruction counts, bytes written; then stage shares (timers build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/living_ac; mkdir -p $O; cd $R
python bench.py --scene living_room --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null > $O/bench.json
python -c "
import json; d=json.loads(open('$O/bench.json').readline()); print('living', d['ms_per_step'], d['value'], d['distributed']['image_crc32'], d.get('oracle_crc_match'))"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "mixed_materials or trace_batch or visible_batch or randomized" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "WRITE_SIZE" "FETCH_SIZE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -o p -- python $R/bench.py --scene living_room --steps 1 --warmup 0 --no-cpu-baseline --no-also > $O/p$i.log 2>&1
done
python - <<PY
import csv,glob
c={}
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_path_fused' in r['Kernel_Name']: c[r['Counter_Name']]=c.get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in sorted(c.items()): print(f'{k:24s} {v:.4g}')
PY
find $O -name '*.csv' -size +1M -delete
cd $R
RL_HIP_FLAGS="-DRL_STAGE_TIMERS" python -m rustlight_amd.build --force > /dev/null 2>&1
python bench.py --scene living_room --steps 1 --warmup 0 --no-cpu-baseline --no-also 2>&1 >/dev/null | grep "\[stage\]" | tail -8
