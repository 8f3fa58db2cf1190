# round 3: BVH4 of the tolerance build on the 508 k-triangle scene
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bvh4 or fast_numerics or small_scenes" 2>&1 | tail -15 | tee $O/tests.log
L=rustlight_amd/lib/librustlight_amd.so
for nm in 0 1; do echo -n "numerics $nm: "; NUMERICS=$nm python scratch/variants.py one $L living_room 2 32 | tail -1; done 2>&1 | tee $O/bench32.log
python - <<'PY' 2>&1 | tee $O/steps.log
import numpy as np
from rustlight_amd import api, scenes
sd = scenes.living_room(1920, 1080)
ctx = api.Context(api.Scene(sd), 0)
rng = np.random.default_rng(3)
o = (rng.uniform(-0.95, 0.95, (400000, 3)) * 3.0).astype(np.float32); o[:, 1] += 4.0
d = rng.normal(size=(400000, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
t2, _, _, m2, tr2 = ctx.trace(o, d)
t4, m4, tr4, st = ctx.trace_fast(o, d)
print("same prim", ((m2 == m4) & (tr2 == tr4)).mean(), "hit rate", (m2 >= 0).mean(), "bvh4 node trips / ray", st.mean(), "max", st.max())
PY
