import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
if len(sys.argv) > 1: api.LIB_PATH = sys.argv[1]
mats = scenes.living_room_materials()
sd = scenes.living_room(96, 64, n_spheres=27, tess=10)
for mesh in sd.meshes:
    if mesh.name.startswith("sphere"): mesh.bsdf = mats[2]
ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(3).block_seeds(96, 64)
out = {}
for md in (2, 3, 8):
    for num in (0, 1):
        img, st = ctx.render(seeds, api.path_params(spp=64, max_depth=md, numerics=num))
        out[f"d{md}_n{num}"] = img
        print(md, num, img.mean(), st["vertices"], st["shadow_rays"], st["extension_rays"], st["rng_draws"], flush=True)
    for strat in (1, 2):
        for num in (0, 1):
            img, st = ctx.render(seeds, api.path_params(spp=64, max_depth=md, numerics=num, strategy=strat))
            print("  strategy", strat, md, num, img.mean(), st["vertices"], flush=True)
np.savez(os.path.join(ROOT, "gpurun_out", "fast_diag2.npz"), **out)
