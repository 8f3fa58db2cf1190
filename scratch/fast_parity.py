"""Parity of the opt-in tolerance build (`numerics = fast`) at BASELINE size: per-pixel squared L2 against the exact build (which is
the oracle's image bit for bit) — mean / 99.9th percentile / max, pixels that differ at all, and the difference of the vertex / draw
counters (paths whose branch decisions flipped).  Run on the GPU box:  python scratch/fast_parity.py > profiles/r02_fast_numerics_parity.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes, provenance
W, H = 1920, 1080
out = {"what": "numerics=fast vs numerics=exact (= oracle bits), 1920x1080, same block seeds (master seed 0)", "kernel_src_hash": provenance.kernel_source_hash(), "commit": os.environ.get("RL_COMMIT"), "cases": {}}
for name, sd, spp in (("cbox", scenes.cbox(W, H), 128), ("living_room", scenes.living_room(W, H), 32), ("cbox_medium", scenes.cbox_medium(W, H, 0.5), 16)):
    ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
    res = {}
    for mode, num in (("exact", 0), ("fast", 1)):
        ctx.render(seeds, api.path_params(spp=1, numerics=num))
        best = 1e9
        for _ in range(2):
            t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=spp, numerics=num)); best = min(best, time.perf_counter() - t)
        res[mode] = (img, st, best)
    a, b = res["exact"][0].astype(np.float64), res["fast"][0].astype(np.float64)
    e = np.sum((a - b) ** 2, axis=-1)
    se, sf = res["exact"][1], res["fast"][1]
    out["cases"][name] = {"spp": spp, "ms_exact": res["exact"][2] * 1e3, "ms_fast": res["fast"][2] * 1e3, "kernel_ms_exact": se["ms_other"], "kernel_ms_fast": sf["ms_other"],
                          "speedup_kernel": se["ms_other"] / sf["ms_other"],
                          "per_pixel_l2_mean": float(e.mean()), "per_pixel_l2_p999": float(np.quantile(e, 0.999)), "per_pixel_l2_max": float(e.max()),
                          "pixels_differing_at_all": int((e > 0).sum()), "pixels_over_1e-3": int((e > 1e-3).sum()), "pixels": W * H,
                          "relative_mean_abs_diff": float(np.abs(a - b).mean() / max(a.mean(), 1e-30)),
                          "vertices_exact": se["vertices"], "vertices_fast": sf["vertices"], "rng_draws_exact": se["rng_draws"], "rng_draws_fast": sf["rng_draws"],
                          "shadow_rays_exact": se["shadow_rays"], "shadow_rays_fast": sf["shadow_rays"]}
print(json.dumps(out, indent=1))
