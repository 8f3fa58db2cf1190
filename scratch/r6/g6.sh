# round 6, GPU call 6: the serial walker of k_stream_spec through treelet blocks (508 k triangles): parity subset, A/B against the plain walker, wave lifetimes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g6; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "speculative or overlapped or full_size or randomized" > $O/pytest_gpu_subset.log 2>&1; tail -4 $O/pytest_gpu_subset.log
{
run() { sc=$1; shift; echo "-- $sc $*"; env "$@" timeout 400 python scratch/ref_bench.py $sc 128 2>&1 | tail -1 | cut -c1-175; }
echo "== 508 k triangles, reference-order, 1080p x 128 spp (4 waves per SIMD, 2 stack levels + the treelet cache in LDS)"
for rep in 1 2 3; do
run living_room REPS=3
run living_room REPS=3 RL_SPEC_NO_TREELETS=1
run living_room REPS=3 RL_SPEC_NO_TREELETS=1 RL_SPEC_LDS_LEVELS=3
done
run cbox REPS=4
run cbox_medium REPS=2
echo "== timers build: shares and wave lifetimes"
RL_NO_OVERLAP=1 RL_SPEC_STATS=1 RL_SPEC_WAVE_TIMES=$O/spec_waves_living_treelets.txt REPS=1 timeout 600 python scratch/ref_bench.py living_room 128 scratch/variants/libtimers.so 2>&1 | tail -8
RL_SPEC_NO_TREELETS=1 RL_NO_OVERLAP=1 RL_SPEC_STATS=1 RL_SPEC_WAVE_TIMES=$O/spec_waves_living_plain.txt REPS=1 timeout 600 python scratch/ref_bench.py living_room 128 scratch/variants/libtimers.so 2>&1 | tail -8
} > $O/log.txt 2>&1
cat $O/log.txt
