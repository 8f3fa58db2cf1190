"""Dev: extreme execution parameters must give the same image (or a clean error), never a fault."""
import sys; sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
sd = scenes.cbox(40, 24)
ctx = api.Context(api.Scene(sd), 0)
seeds = api.IndependentSampler(1).block_seeds(sd.width, sd.height)
ref = ctx.render(seeds, api.path_params(spp=3))[0]
cases = [dict(pool_slots=1), dict(pool_slots=3), dict(pool_slots=257), dict(pool_slots=(1 << 31) - 1), dict(pool_slots=0xFFFFFFFF), dict(sample_split=1000000), dict(sample_split=3, pipeline=1, pool_slots=7),
         dict(pipeline=2, pool_slots=5), dict(pipeline=3)]
for kw in cases:
    try:
        img = ctx.render(seeds, api.path_params(spp=3, **kw))[0]
        print(kw, "same" if np.array_equal(img, ref) else "DIFF", flush=True)
    except api.RustlightError as e:
        print(kw, "error:", str(e)[:80], flush=True)
parts = [ctx.render(seeds, api.path_params(spp=3, shard_index=r, shard_count=1000))[0] for r in range(0, 1000, 1)]
print("1000 shards sum", "same" if np.array_equal(sum(parts[1:], parts[0]), ref) else "DIFF")
img = ctx.render(seeds, api.path_params(spp=70000, max_depth=2, sample_split=0))[0]; print("spp 70000 ok", float(img.mean()))
