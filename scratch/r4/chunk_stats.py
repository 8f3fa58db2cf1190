import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rustlight_amd import api, scenes
import numpy as np
sd = scenes.cbox(640, 400)
ctx = api.Context(api.Scene(sd), 0)
seeds = api.IndependentSampler(1).block_seeds(640, 400)
pp = api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, spp=128)
base = None
for env in ({}, dict(RL_STATE_BUDGET_MB='200'), dict(RL_STATE_BUDGET_MB='120'), dict(RL_STATE_BUDGET_MB='200', RL_CHAIN_SERIAL='1'), dict(RL_STATE_BUDGET_MB='60'), dict(RL_STATE_BUDGET_MB='200', RL_SPEC_GROUP='16')):
    os.environ.update(env)
    img, st = ctx.render(seeds, pp)
    for k in env: del os.environ[k]
    if base is None: base = (img, st)
    print(env, 'chunks', st['iterations'], 'same image', np.array_equal(img, base[0]), {k: st[k] - base[1][k] for k in ('camera_samples', 'vertices', 'extension_rays', 'shadow_rays', 'rng_draws')}, 'split?', st['kernel_launches'])
