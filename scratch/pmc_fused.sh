cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $grp | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcf/$tag -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/pmcf/$tag.log 2>&1
  f=$(find $R/gpurun_out/pmcf/$tag -name '*counter_collection.csv' | head -1)
  python $R/scratch/pmc_sum.py $f ${KFILTER:-k_path_fused}
done
