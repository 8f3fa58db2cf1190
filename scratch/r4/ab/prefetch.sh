# dev: leaf prefetch in the voted traversal loop of the streaming kernels (trace.hip.h: RL_LEAF_PREFETCH), same box A/B
R=$GRAFT_REPO_ROOT; cd $R
one() {
  for a in "" "--tris 4000000"; do
    python bench.py --scene living_room $a --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1 [$a]', round(d['ms_per_step'],1), d['distributed']['image_crc32'])"
  done
}
one off; one off
RL_HIP_FLAGS="-DRL_LEAF_PREFETCH=1" python -m rustlight_amd.build --force > /dev/null 2>&1
one on; one on
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "mixed_materials or trace_batch or visible_batch or randomized" 2>&1 | grep -E "passed|failed" | tail -2
