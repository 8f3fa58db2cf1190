# memory-pipeline counters of the streaming fused kernel (living-room stand-in, 32 spp): is the vector-memory address path (TA / TCP tag lookups, TLB) the limiter?
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r2e; mkdir -p $O
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" "TCP_LFIFO_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene living_room --spp 32 > $O/p$i.log 2>&1
  python $R/scratch/pmc_sum.py $(find $O/p$i -name '*counter_collection.csv' | head -1) k_path_fused
done
find $O -name '*agent_info.csv' -delete
