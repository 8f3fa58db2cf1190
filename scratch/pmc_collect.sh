# Collect SQ / TCC / HBM counters of one bench.py workload in separate rocprofv3 --pmc passes (never combined with other trace
# domains) and summarise them into gpurun_out/<tag>/pmc_summary.json (copy into profiles/ and merge into profiles/pmc_live.json
# with scratch/pmc_summary.py --merge).
#   usage (on the GPU box):  RL_COMMIT=<sha> bash scratch/pmc_collect.sh <tag> <kernel-filter> [bench.py args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KF=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also "$@" > $O/p$i.log 2>&1
done
python $R/scratch/pmc_summary.py $O "$KF" "$@" > $O/pmc_summary.json
cat $O/pmc_summary.json | head -c 3000
find $O -name '*counter_collection.csv' -size +4M -delete; find $O -name '*kernel_trace.csv' -size +4M -delete; find $O -name '*agent_info.csv' -delete
