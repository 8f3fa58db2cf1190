import os, time, sys
sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "cbox"
sd = scenes.cbox(1920,1080) if scene == "cbox" else scenes.cbox_medium(1920,1080,0.5)
ctx = api.Context(api.Scene(sd),0); seeds = api.IndependentSampler(0).block_seeds(1920,1080)
for pl in (1, 2):
    best = 1e9
    for r in range(3):
        t=time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=128, pipeline=pl)); dt=time.perf_counter()-t; best=min(best,dt)
    print(scene, "pipeline", pl, round(best*1e3,1), "ms", round(1920*1080*128/best/1e6), "Msamples/s", "mean", float(img.mean()))
