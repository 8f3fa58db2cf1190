"""Dev: time rank 0 of an N-way shard at spp = 128 N (the weak-scaling workload of bench.py) for several sample_split values."""
import sys, time, zlib
sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
W, H = 1920, 1080
ctx = api.Context(api.Scene(scenes.cbox(W, H)), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
for N in (1, 2, 4, 8):
    for split in [int(x) for x in sys.argv[1].split(",")]:
        best = 1e9
        for r in range(3):
            t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=128 * N, shard_index=0, shard_count=N, sample_split=split)); best = min(best, time.perf_counter() - t)
        print(f"N={N} split={split:2d} {best*1e3:8.1f} ms  per-rank {W*H*128/best/1e6:7.0f} Msamples/s crc {zlib.crc32(img.tobytes()):08x}", flush=True)
