import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
mats = scenes.living_room_materials()
sd = scenes.living_room(96, 64, n_spheres=27, tess=10)
for mesh in sd.meshes:
    if mesh.name.startswith("sphere"): mesh.bsdf = mats[2]
ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(3).block_seeds(96, 64)
for kw in (dict(max_depth=3), dict(max_depth=3, rr_depth=None), dict(max_depth=3, rr_depth=None, strategy=1), dict(max_depth=4, rr_depth=None, strategy=1)):
    for num in (0, 1):
        img, st = ctx.render(seeds, api.path_params(spp=64, numerics=num, **kw))
        print(kw, num, img.mean(), st["vertices"], st["shadow_rays"], st["extension_rays"], st["rng_draws"], flush=True)
