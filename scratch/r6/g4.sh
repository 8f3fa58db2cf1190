# round 6, GPU call 4: k_stream_spec occupancy x helpers combos on the 508 k-triangle scene (3 reps), and the spill A/Bs of k_path_fused<-1,false,false,..>
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6g4; mkdir -p $O
{
run() { lib=$1; shift; echo "-- $lib $*"; env "$@" REPS=3 timeout 300 python scratch/ref_bench.py living_room 128 $lib 2>&1 | tail -1 | cut -c1-175; }
D=rustlight_amd/lib/librustlight_amd.so; W4=scratch/variants/libw4.so
echo "== 508 k triangles, reference-order, 1080p x 128 spp"
for rep in 1 2; do
run $D X=0
run $W4 RL_SPEC_LDS_LEVELS=2
run $W4 RL_SPEC_LDS_LEVELS=3
run $W4 RL_SPEC_LDS_LEVELS=1
run $W4 RL_SPEC_LDS_LEVELS=2 RL_SPEC_DENSE=16
run $W4 RL_SPEC_LDS_LEVELS=3 RL_SPEC_DENSE=16
run $W4 RL_SPEC_LDS_LEVELS=2 RL_SPEC_DENSE=8
run $W4 RL_SPEC_LDS_LEVELS=2 RL_SPEC_DENSE=32
run $D RL_SPEC_LDS_LEVELS=2
done
echo "== spill A/Bs: k_path_fused<-1,false,false,..>, 508 k triangles, per-sample streams, 1080p x 128 spp (best of 3 each)"
for rep in 1 2; do for v in $D scratch/variants/libni.so scratch/variants/libcs13.so scratch/variants/libcs8.so; do timeout 300 python scratch/variants.py one $v living_room 2 128 2>&1 | tail -1; done; done
} > $O/log.txt 2>&1
cat $O/log.txt
cd /tmp && export TMPDIR=/tmp
for v in default ni cs13; do
  lib=$GRAFT_REPO_ROOT/scratch/variants/lib$v.so; [ $v = default ] && lib=$GRAFT_REPO_ROOT/rustlight_amd/lib/librustlight_amd.so
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$v -o p -- python $GRAFT_REPO_ROOT/scratch/variants.py one $lib living_room 2 32 > $GRAFT_REPO_ROOT/$O/pmc_$v.log 2>&1
  python - <<PY
import csv, glob
c = {}
for f in glob.glob("$GRAFT_REPO_ROOT/$O/pmc_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_path_fused" in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("$v (3 launches at 32 spp, KB):", c)
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/pmc_write_size.txt
find $GRAFT_REPO_ROOT/$O -name '*.csv' -size +2M -delete
