# dev: A/B on one box — current tree vs the generic-pointer traversal stack.  Before the call: git show dd30cc2:rustlight_amd/csrc/kernels/trace.hip.h > scratch/r4/ab/trace.hip.h.old (same for pathstate.hip.h); the copies are not kept in the tree
R=$GRAFT_REPO_ROOT; cd $R
one() {
  for a in "" "--numerics fast" "--tris 4000000" "--tris 4000000 --numerics fast"; do
    python bench.py --scene living_room $a --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1 [$a]', round(d['ms_per_step'],1), d['distributed']['image_crc32'])"
  done
}
one new; one new
cp scratch/r4/ab/trace.hip.h.old rustlight_amd/csrc/kernels/trace.hip.h; cp scratch/r4/ab/pathstate.hip.h.old rustlight_amd/csrc/kernels/pathstate.hip.h
python -m rustlight_amd.build --force > /dev/null 2>&1
one old; one old
