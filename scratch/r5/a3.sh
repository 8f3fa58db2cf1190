# round 5, call 3: the new tests, the default bench line, and the counter A/B of the two-level records (508 k triangles, 32 spp)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_level or cfg4 or bench_two_ranks or trace_batch" 2>&1 | tail -15 | tee gpurun_out/r5/a3_tests.txt
python bench.py > gpurun_out/r5/a3_bench.json 2> gpurun_out/r5/a3_bench.err; tail -c 600 gpurun_out/r5/a3_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/a3_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','reference_order_value','oracle_crc_match')}, {k:d['roofline'].get(k) for k in ('frac','valu_issue_busy','lane_utilisation','valu_lane_slots_used')})
for a in d['also']: print(a['workload'], round(a.get('ms_per_step',0),1), round(a.get('value',0),1), a.get('oracle_crc_match'), a.get('kernel_ms'))
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in one two; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r5/pmc_$v/p$i -o p -- python $R/scratch/variants.py one $R/scratch/variants/lib$v.so living_room 2 32 > $R/gpurun_out/r5/pmc_$v/p$i.log 2>&1
  done
  python $R/scratch/r5/pmc_ab_sum.py $R/gpurun_out/r5/pmc_$v k_path_fused > $R/gpurun_out/r5/pmc_ab_$v.json
  find $R/gpurun_out/r5/pmc_$v -name '*.csv' -size +1M -delete
done
head -c 1500 $R/gpurun_out/r5/pmc_ab_one.json; head -c 1500 $R/gpurun_out/r5/pmc_ab_two.json
