# round 6, GPU call 1: evidence for the two slowest driver-timed kernels (k_stream_chain on cbox + medium, k_stream_spec on 508 k triangles, both in reference-order streams),
# and the shadow-stage occupancy of the headline kernel.   usage: RL_COMMIT=<sha> bash scratch/r6/g1.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g1; mkdir -p $O
T=scratch/variants/libtimers.so
{
echo "== baseline (shipped library), reference-order streams, 1080p x 128 spp"
for sc in cbox_medium living_room; do REPS=3 timeout 300 python scratch/ref_bench.py $sc 128 2>&1 | tail -1; done
echo "== timers build, overlap off (clean shares)"
for sc in cbox_medium living_room; do RL_NO_OVERLAP=1 RL_SPEC_STATS=1 RL_CHAIN_WAVE_TIMES=$O/chain_waves_$sc.txt RL_SPEC_WAVE_TIMES=$O/spec_waves_$sc.txt REPS=1 timeout 600 python scratch/ref_bench.py $sc 128 $T 2>&1 | grep -v "^\[stage\]" | tail -12; done
echo "== timers build, headline kernel (per-sample streams): stage shares and shadow-stage occupancy"
timeout 300 python scratch/variants.py one $T cbox 2 128 2>&1 | tail -8
VW=1080 VH=1080 timeout 300 python scratch/variants.py one $T cbox 2 128 2>&1 | tail -8
timeout 300 python scratch/variants.py one $T cbox_medium 2 32 2>&1 | tail -8
timeout 300 python scratch/variants.py one $T living_room 2 32 2>&1 | tail -8
} > $O/log.txt 2>&1
RL_NO_OVERLAP=1 bash scratch/pmc_collect.sh r6pmc_medium_reference k_stream_chain --scene cbox_medium --stream-mode reference > $O/pmc_medium_reference.log 2>&1
RL_NO_OVERLAP=1 bash scratch/pmc_collect.sh r6pmc_living_reference k_stream_spec --scene living_room --stream-mode reference > $O/pmc_living_reference.log 2>&1
cp gpurun_out/r6pmc_medium_reference/pmc_summary.json $O/pmc_medium_reference.json; cp gpurun_out/r6pmc_living_reference/pmc_summary.json $O/pmc_living_reference.json
cat $O/log.txt
