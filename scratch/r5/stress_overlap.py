"""Stress the overlapped evaluation pass: many reference-order renders of varied shape in fresh contexts; prints one line per render (a hang shows as a missing line under `timeout`)."""
import os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from rustlight_amd import api, scenes
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mk = [lambda w, h: scenes.cbox(w, h), lambda w, h: scenes.cbox_medium(w, h, 0.5), lambda w, h: scenes.living_room(w, h, n_spheres=27, tess=10)]
for it in range(n):
    k = int(rng.integers(0, 3))
    w, h = int(rng.integers(3, 60)) * 16 + int(rng.integers(0, 16)), int(rng.integers(3, 40)) * 16 + int(rng.integers(0, 16))
    spp = int(rng.choice([4, 16, 64, 128, 256]))
    if k == 1: spp = min(spp, 16)
    shard = (0, 1) if rng.random() < 0.6 else (int(rng.integers(0, 4)), 4)
    env = {}
    if rng.random() < 0.5: env["RL_SPEC_FORCE"] = "1"
    if k != 2 and rng.random() < 0.3: env["RL_FORCE_STREAMING"] = "1"
    for a, b in env.items(): os.environ[a] = b
    ctx = api.Context(api.Scene(mk[k](w, h)), 0)
    seeds = api.IndependentSampler(it).block_seeds(w, h)
    t = time.perf_counter()
    crcs = []
    for rep in range(3):
        img, st = ctx.render(seeds, api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER, shard_index=shard[0], shard_count=shard[1]))
        crcs.append(zlib.crc32(img.tobytes()))
    os.environ["RL_NO_OVERLAP"] = "1"
    img, st0 = ctx.render(seeds, api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER, shard_index=shard[0], shard_count=shard[1]))
    del os.environ["RL_NO_OVERLAP"]
    ok = all(c == zlib.crc32(img.tobytes()) for c in crcs)
    print(f"{it:3d} scene {k} {w}x{h}x{spp} shard {shard} {env} spec_group {st['spec_group']} {(time.perf_counter()-t)*1e3:8.1f} ms {'OK' if ok else 'MISMATCH'}", flush=True)
    ctx.close()
    for a in env: del os.environ[a]
