import sys, time, zlib, os
sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
api.LIB_PATH = sys.argv[1]
W, H, SPP = 1920, 1080, int(sys.argv[2])
ctx = api.Context(api.Scene(scenes.living_room(W, H)), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
for split in [int(x) for x in sys.argv[3].split(",")]:
    for pool in [int(x) for x in sys.argv[4].split(",")]:
        best = 1e9
        for r in range(2):
            t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=SPP, pipeline=1, sample_split=split, pool_slots=pool)); best = min(best, time.perf_counter() - t)
        print(f"split {split:2d} pool {pool:8d}: {best*1e3:8.1f} ms {W*H*SPP/best/1e6:6.0f} Msamples/s iters {st['iterations']} crc {zlib.crc32(img.tobytes()):08x}", flush=True)
