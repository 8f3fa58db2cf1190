import json, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rustlight_amd import api, scenes
W, H = 1920, 1080
ctx = api.Context(api.Scene(scenes.cbox(W, H)), 0)
seeds = api.IndependentSampler(0).block_seeds(W, H)
for n, spp in ((8, 1024), (8, 128), (2, 256), (2, 128)):
    pp = api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER, shard_index=0, shard_count=n)
    for env in ({}, dict(RL_SPEC_KS='2.5', RL_SPEC_KE='2.5'), dict(RL_SPEC_GROUP='256', RL_SPEC_SUB='16'), dict(RL_SPEC_GROUP='256', RL_SPEC_SUB='16', RL_SPEC_KS='2.5', RL_SPEC_KE='2.5'), dict(RL_SPEC_GROUP='64', RL_SPEC_SUB='8')):
        os.environ.update(env); os.environ['RL_SPEC_STATS'] = '1'
        t0 = time.perf_counter(); img, st = ctx.render(seeds, pp); dt = (time.perf_counter() - t0) * 1e3
        for k in env: del os.environ[k]
        print(json.dumps({'n': n, 'spp': spp, 'env': env, 'ms': round(dt, 1), 'chain_ms': round(st['ms_prepass'], 1), 'group': st['spec_group'], 'serial_per_pixel': round(st['spec_serial_samples'] / (W * H / n), 2),
                          'walked_x': round(st['spec_samples'] / max(1, st['camera_samples']), 2), 'crc': '%08x' % zlib.crc32(img.tobytes())}), flush=True)
