# round 6, GPU call 3: the refactored render driver through the whole GPU suite; occupancy A/B of k_stream_spec on the 508 k-triangle scene; chains per wave on cbox + medium.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
{
run() { lib=$1; shift; echo "-- $lib $*"; env "$@" REPS=2 timeout 300 python scratch/ref_bench.py living_room 128 $lib 2>&1 | tail -1 | cut -c1-175; }
D=rustlight_amd/lib/librustlight_amd.so; W4=scratch/variants/libw4.so; W5=scratch/variants/libw5.so
echo "== 508 k triangles, reference-order, 1080p x 128 spp: k_stream_spec occupancy (waves per SIMD the registers are budgeted for x stack levels in LDS) and helpers"
for rep in 1 2; do
run $D X=0
run $D RL_SPEC_LDS_LEVELS=4
run $W4 RL_SPEC_LDS_LEVELS=4
run $W4 RL_SPEC_LDS_LEVELS=2
run $W4 RL_SPEC_LDS_LEVELS=0
run $W5 RL_SPEC_LDS_LEVELS=0
run $D RL_SPEC_DENSE=16
run $W4 RL_SPEC_LDS_LEVELS=4 RL_SPEC_DENSE=16
run $D RL_SPEC_LEAD=36 RL_SPEC_DENSE=16
done
echo "== cbox + medium: chains per wave (default now 4 per wave = RL_ITEM_SHIFT 4)"
for rep in 1 2; do for k in 4 5; do RL_ITEM_SHIFT=$k REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175; done; done
REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1 | cut -c1-175
REPS=3 timeout 300 python scratch/ref_bench.py cbox 128 2>&1 | tail -1 | cut -c1-175
} > $O/log.txt 2>&1
cat $O/log.txt
