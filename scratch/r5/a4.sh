# counter A/B of the two-level records (508 k triangles, 1080p x 32 spp, per-sample streams): one = one-level records (default), two = -DRL_TWO_LEVEL=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in one two; do
  mkdir -p $R/gpurun_out/r5/pmc_$v
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/r5/pmc_$v/p$i -o p -- python $R/scratch/variants.py one $R/scratch/variants/lib$v.so living_room 2 32 > $R/gpurun_out/r5/pmc_$v/p$i.log 2>&1
  done
  python $R/scratch/r5/pmc_ab_sum.py $R/gpurun_out/r5/pmc_$v k_path_fused > $R/gpurun_out/r5/pmc_ab_$v.json
  find $R/gpurun_out/r5/pmc_$v -name '*.csv' -size +1M -delete
done
head -c 1800 $R/gpurun_out/r5/pmc_ab_one.json; head -c 1800 $R/gpurun_out/r5/pmc_ab_two.json
