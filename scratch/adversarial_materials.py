"""Dev: hostile BSDF parameters (0, negative, NaN, inf, 1e30, 1e-30 in exponents, roughness, weights, indices of refraction, colours)
— HIP path vs oracle, NaNs included."""
import sys; sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
from oracle import orc
S = scenes
H = [0.0, -1.0, float("nan"), float("inf"), 1e30, 1e-30, 1.0, 0.5, -0.0, 2.0]
bad = n = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    rng = np.random.default_rng(seed)
    pick = lambda: float(H[rng.integers(0, len(H))])
    col = lambda: S.const_color((pick(), pick(), pick())) if rng.random() < 0.5 else S.const_color((0.6, 0.5, 0.4))
    sd = S.cbox(20, 16)
    for m in sd.meshes:
        if m.emission: continue
        t = int(rng.integers(0, 5))
        m.bsdf = S.Bsdf(type=t, diffuse=col(), specular=col(), transmittance=col(), eta=col(), k=col(), exponent=pick(), weight_specular=pick(),
                        distribution=int(rng.integers(0, 3)), alpha_u=pick(), alpha_v=pick(), glass_eta=pick())
    kw = dict(spp=2, max_depth=int(rng.integers(2, 7)), strategy=int(rng.integers(0, 3)))
    try:
        scene = api.Scene(sd)
    except api.RustlightError as e:
        print(seed, "refused:", str(e)[:80]); continue
    osc = orc.Scene(sd)
    ref, ost = osc.render(master_seed=seed, eval_order=1, **kw)
    for pl in (1, 2):
        img, st = api.Context(scene, 0).render(api.IndependentSampler(seed).block_seeds(sd.width, sd.height), api.path_params(pipeline=pl, **kw))
        same = np.array_equal(img, ref, equal_nan=True) and all(st[k] == ost[k] for k in ("vertices", "rng_draws", "shadow_rays"))
        n += 1; bad += not same
        if not same:
            nan_g, nan_r = np.isnan(img), np.isnan(ref)
            print("DIFF seed", seed, "pl", pl, kw, "nan gpu/ref", nan_g.sum(), nan_r.sum(), "finite diffs", ((img != ref) & ~nan_g & ~nan_r).sum(), {k: (st[k], ost[k]) for k in ("vertices", "rng_draws", "shadow_rays")}, flush=True)
print("cases", n, "failures", bad)
