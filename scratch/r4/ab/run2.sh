# dev: streaming kernels after the stack flavours (typed: exact build, generic: BVH4 of the tolerance build) + the GPU suite + evidence
R=$GRAFT_REPO_ROOT; cd $R
for a in "" "--numerics fast" "--tris 4000000" "--tris 4000000 --numerics fast"; do
  python bench.py --scene living_room $a --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('[$a]', round(d['ms_per_step'],1), d['distributed']['image_crc32'])"
done
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
bash scratch/r4/profile_round4.sh > gpurun_out/r4prof_run.log 2>&1; tail -3 gpurun_out/r4prof_run.log | cut -c1-300
