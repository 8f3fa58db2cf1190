# round 6, GPU call 5: the tests the round's changes touch; launch order by chain time (block_order = cost) on the 508 k-triangle scene and the Cornell box; WRITE_SIZE of the spill variants
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "speculative or overlapped or cfg4 or two_pass or full_size or in_flight" > $O/pytest_gpu_subset.log 2>&1; tail -4 $O/pytest_gpu_subset.log
{
run() { sc=$1; shift; echo "-- $sc $*"; env "$@" timeout 400 python scratch/ref_bench.py $sc 128 2>&1 | tail -1 | cut -c1-175; }
echo "== reference-order, 1080p x 128 spp; REPS renders per process, best reported (block_order = cost: the first render measures, the later ones are ordered)"
for rep in 1 2 3; do
run living_room REPS=4
run living_room REPS=4 RL_BLOCK_ORDER=cost
run living_room REPS=4 RL_BLOCK_ORDER=cost RL_SPEC_DENSE=16
run living_room REPS=4 RL_SPEC_DENSE=16
done
for rep in 1 2 3; do
run cbox REPS=5
run cbox REPS=5 RL_BLOCK_ORDER=cost
done
} > $O/log.txt 2>&1
cat $O/log.txt
cd /tmp && export TMPDIR=/tmp
for v in default cs13; do
  lib=$GRAFT_REPO_ROOT/scratch/variants/lib$v.so; [ $v = default ] && lib=$GRAFT_REPO_ROOT/rustlight_amd/lib/librustlight_amd.so
  for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${v}_$c -o p -- python $GRAFT_REPO_ROOT/scratch/variants.py one $lib living_room 2 32 > $GRAFT_REPO_ROOT/$O/pmc_${v}_$c.log 2>&1
  done
  python - <<PY
import csv, glob
c = {}
for f in glob.glob("$GRAFT_REPO_ROOT/$O/pmc_${v}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_path_fused" in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("$v (3 launches at 32 spp, KB):", c)
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/pmc_write_size.txt
find $GRAFT_REPO_ROOT/$O -name '*.csv' -size +2M -delete
