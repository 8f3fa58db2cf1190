"""Which BSDF makes numerics=fast drift?  One material at a time on the spheres of a small living-room scene: exact vs fast."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
if len(sys.argv) > 1: api.LIB_PATH = sys.argv[1]
mats = scenes.living_room_materials()
names = ["diffuse", "phong", "mirror", "ggx", "glass", "substrate"]
for k, (name, m) in enumerate(zip(names, mats)):
    sd = scenes.living_room(96, 64, n_spheres=27, tess=10)
    for mesh in sd.meshes:
        if mesh.name.startswith("sphere"):
            mesh.bsdf = m
    ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(3).block_seeds(96, 64)
    a, sa = ctx.render(seeds, api.path_params(spp=16, max_depth=8))
    b, sb = ctx.render(seeds, api.path_params(spp=16, max_depth=8, numerics=1))
    e = np.sum((a.astype(np.float64) - b) ** 2, -1)
    print(f"{name:10s} mean L2 {e.mean():.3e} p999 {np.quantile(e, 0.999):.3e} rel mean |d| {np.abs(a - b).mean() / a.mean():.4f} vertices {sa['vertices']} / {sb['vertices']}  draws {sa['rng_draws']} / {sb['rng_draws']}  finite {np.isfinite(b).all()} mean {a.mean():.4f} / {b.mean():.4f}", flush=True)
