# build-flag variants of the streaming fused kernel on the 508 k-triangle scene, 1080p x 128 spp per-sample streams (built on the GPU box)
R=$GRAFT_REPO_ROOT; cd $R
for flags in "" "-DRL_FUSED_WAVES_STREAMING=5" "-DRL_FUSED_WAVES_STREAMING=4"; do
  RL_HIP_FLAGS="$flags" python -m rustlight_amd.build --force > /dev/null 2>&1
  python -m rustlight_amd.resources 2>/dev/null | grep "k_path_fused<-1, false, false, 1, 0>" | head -1 | cut -c1-260
  echo "== flags: [$flags]"
  python bench.py --scene living_room --steps 2 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('ms_per_step','value')}, d['distributed']['image_crc32'])"
done
