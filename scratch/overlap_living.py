import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
W, H, SPP = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = scenes.living_room(W, H)
scene = api.Scene(sd)
seeds = api.IndependentSampler(0).block_seeds(W, H)
ref = None
for nshards in (1, 2, 4, 8):
    ctxs = [api.Context(scene, 0) for _ in range(nshards)]
    outs = [None] * nshards
    def work(i):
        outs[i] = ctxs[i].render(seeds, api.path_params(spp=SPP, shard_index=i, shard_count=nshards))
    for rep in range(2):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nshards)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
    img = sum(o[0] for o in outs[1:]) + outs[0][0] if nshards > 1 else outs[0][0]
    if ref is None: ref = img
    print(nshards, "concurrent shards:", round(dt * 1e3, 1), "ms ->", round(W * H * SPP / dt / 1e6), "Msamples/s", "same" if np.array_equal(ref, img) else "DIFF", flush=True)
    del ctxs
