"""Dev: extreme parameters of the ao / direct integrators vs the oracle."""
import sys; sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
from oracle import orc
sd = scenes.cbox(24, 16)
ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
seeds = api.IndependentSampler(2).block_seeds(sd.width, sd.height)
bad = 0
for name, kw in (("direct", dict(nb_bsdf_samples=0, nb_light_samples=0)), ("direct", dict(nb_bsdf_samples=37, nb_light_samples=0)), ("direct", dict(nb_bsdf_samples=0, nb_light_samples=41)),
                 ("ao", dict(max_distance=float("nan"))), ("ao", dict(max_distance=-1.0)), ("ao", dict(max_distance=float("inf"))), ("ao", dict(max_distance=0.0)), ("ao", dict(max_distance=1e-30, normal_correction=True))):
    for spp in (1, 5):
        try:
            img, st = (ctx.render_direct if name == "direct" else ctx.render_ao)(seeds, spp=spp, **kw)
        except api.RustlightError as e:
            print(name, kw, spp, "error:", str(e)[:80]); continue
        ref, ost = (osc.render_direct if name == "direct" else osc.render_ao)(seeds=seeds, spp=spp, **kw)
        same = np.array_equal(img, ref, equal_nan=True) and st["rng_draws"] == ost["rng_draws"]
        bad += not same
        print(name, kw, spp, "same" if same else "DIFF", flush=True)
print("failures", bad)
