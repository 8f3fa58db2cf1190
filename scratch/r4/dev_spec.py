"""dev: reference-order streams, k_stream_spec vs the serial chain (image equality, timings, counters)."""
import os, sys, time, zlib, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rustlight_amd import api, scenes
W, H, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = sys.argv[4] if len(sys.argv) > 4 else 'cbox'
sd = {'cbox': lambda: scenes.cbox(W, H), 'medium': lambda: scenes.cbox_medium(W, H, 0.5), 'living': lambda: scenes.living_room(W, H)}[scene]()
ctx = api.Context(api.Scene(sd), 0)
seeds = api.IndependentSampler(0).block_seeds(W, H)
pp = api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, spp=spp)
def run(env, reps=2):
    for k, v in env.items(): os.environ[k] = v
    best = None
    for _ in range(reps):
        t0 = time.time(); img, st = ctx.render(seeds, pp); dt = time.time() - t0
        if best is None or dt < best[0]: best = (dt, img, st)
    for k in env: del os.environ[k]
    dt, img, st = best
    print(json.dumps({'env': env, 'ms': round(dt * 1e3, 1), 'chain_ms': round(st['ms_prepass'], 1), 'eval_ms': round(st['ms_other'], 1), 'Msamples/s': round(W * H * spp / dt / 1e6, 1),
                      'crc': '%08x' % zlib.crc32(img.tobytes()), 'spec': [st['spec_samples'], st['spec_serial_samples'], st['spec_probe_samples'], st['spec_group']],
                      'draws': st['rng_draws']}), flush=True)
    return img
variants = [dict(RL_SPEC_STATS='1')] + [dict(a.split('=') for a in v.split(',')) for v in sys.argv[5:]]
if os.environ.get('RL_SPEC_ONLY'):
    run({}, reps=1); sys.exit(0)
ref = run(dict(RL_CHAIN_SERIAL='1'), reps=1)
for env in variants:
    img = run(env)
    print('  equal to the serial chain:', bool(np.array_equal(img, ref)), flush=True)
