# round 6, GPU call 2: knob sweeps (no rebuild) on the two slowest driver-timed chain passes.   usage: bash scratch/r6/g2.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g2; mkdir -p $O
{
echo "== cbox + medium, reference-order, 1080p x 128 spp: chains per wave (RL_ITEM_SHIFT: one chain per 2^k lanes; auto = 5)"
for k in 5 4 3 6 2; do RL_ITEM_SHIFT=$k REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1; done
echo "== ... with the evaluation pass after the chain pass (RL_NO_OVERLAP) at 4 and 3"
for k in 4 3; do RL_NO_OVERLAP=1 RL_ITEM_SHIFT=$k REPS=2 timeout 300 python scratch/ref_bench.py cbox_medium 128 2>&1 | tail -1; done
echo "== 508 k triangles, reference-order, 1080p x 128 spp: k_stream_spec's windows"
run() { echo "-- $*"; env "$@" REPS=2 timeout 300 python scratch/ref_bench.py living_room 128 2>&1 | tail -1; }
run RL_SPEC_STATS=1
run RL_SPEC_LEAD=12
run RL_SPEC_LEAD=36
run RL_SPEC_LEAD=48
run RL_SPEC_KS=3.5 RL_SPEC_KE=3.5
run RL_SPEC_KS=1.65 RL_SPEC_KE=1.65
run RL_SPEC_SUB=2
run RL_SPEC_SUB=8
run RL_SPEC_GROUP=32
run RL_SPEC_DENSE=16
run RL_SPEC_EXTRA=1
run RL_SPEC_LEAD_VAR=0
run RL_SPEC_PROBE_EVERY=1
} > $O/log.txt 2>&1
cat $O/log.txt
