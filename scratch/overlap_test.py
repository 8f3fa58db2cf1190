import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
sd = scenes.cbox(1920, 1080)
scene = api.Scene(sd)
seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
def run(nshards):
    ctxs = [api.Context(scene, 0) for _ in range(nshards)]
    outs = [None]*nshards
    def work(i):
        outs[i] = ctxs[i].render(seeds, api.path_params(spp=128, shard_index=i, shard_count=nshards))
    for rep in range(2):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nshards)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
    img = sum(o[0] for o in outs)
    print(nshards, "concurrent shards:", round(dt*1e3,1), "ms ->", round(1920*1080*128/dt/1e6), "Msamples/s", float(img.mean()))
    return img
a = run(1); b = run(2); c = run(4)
print(np.array_equal(a,b), np.array_equal(a,c))
