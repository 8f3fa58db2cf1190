"""reference-order streams, shard 0 of N on one GPU: lanes per block 64 vs 256 (a workgroup per block)"""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rustlight_amd import api, scenes
W, H = 1920, 1080
ctx = api.Context(api.Scene(scenes.cbox(W, H)), 0)
seeds = api.IndependentSampler(0).block_seeds(W, H)
for scaling, n in (('strong', 1), ('strong', 8), ('weak', 8), ('weak', 4), ('strong', 2)):
    spp = 128 * n if scaling == 'weak' else 128
    pp = api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER, shard_index=0, shard_count=n)
    ref = None
    for env in (dict(RL_SPEC_GROUP='64', RL_SPEC_SUB='2'), dict(RL_SPEC_GROUP='256', RL_SPEC_SUB='4'), dict(RL_SPEC_GROUP='256', RL_SPEC_SUB='8'), dict(RL_SPEC_GROUP='256', RL_SPEC_SUB='2')):
        if n == 1 and env['RL_SPEC_GROUP'] == '64': env = {}
        os.environ.update(env)
        best = None
        for rep in range(2):
            t0 = time.perf_counter(); img, st = ctx.render(seeds, pp); dt = (time.perf_counter() - t0) * 1e3
            if best is None or dt < best[0]: best = (dt, st)
        for k in env: del os.environ[k]
        crc = zlib.crc32(img.tobytes())
        if ref is None: ref = crc
        dt, st = best
        print(json.dumps({'scaling': scaling, 'n': n, 'spp': spp, 'env': env, 'ms': round(dt, 1), 'chain_ms': round(st['ms_prepass'], 1), 'eval_ms': round(st['ms_other'], 1), 'group': st['spec_group'],
                          'serial_per_pixel': round(st['spec_serial_samples'] / (W * H / n), 2), 'walked_x': round(st['spec_samples'] / max(1, st['camera_samples']), 2), 'same_image': crc == ref}), flush=True)
