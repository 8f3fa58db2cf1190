cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcl2; mkdir -p $O
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  tag=$(echo $grp | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene living_room --spp 16 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/scratch/pmc_sum.py $f "k_extend" || tail -3 $O/$tag.log
done
find $O -name '*.csv' -size +2M -delete
