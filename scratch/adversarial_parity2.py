"""Dev: hostile shading inputs — NaN / inf / huge uv and normals on textured meshes (bitmap, checkerboard, grid), NaN / inf / negative
texels in bitmaps and in the environment map, NaN BSDF parameters — HIP path vs oracle (bits, NaNs included, + counters)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
from oracle import orc
S = scenes
def make(kind, seed):
    rng = np.random.default_rng(seed)
    sd = S.cbox(24, 20) if kind < 6 else S.sky_scene(24, 20, keep_area_light=bool(seed % 2))
    bm = rng.uniform(0, 1, (5, 7, 3)).astype(np.float32)
    if kind == 3: bm[1, 2] = np.nan; bm[3, 3] = np.inf; bm[0, 0] = -1.0
    sd.bitmaps = [(7, 5, bm)]
    texs = [{"type": S.TEX_BITMAP, "bitmap_id": 0, "color0": (1, 1, 1), "scale": (2.0, 3.0)}, {"type": S.TEX_CHECKERBOARD, "color0": (0.8, 0.8, 0.8), "color1": (0.1, 0.2, 0.3), "scale": (4.0, 4.0)},
            {"type": S.TEX_GRID, "color0": (0.9, 0.1, 0.1), "color1": (0.4, 0.4, 0.4), "line_width": 0.05, "scale": (3.0, 2.0)}]
    for i, m in enumerate(sd.meshes):
        if m.emission: continue
        n = len(m.vertices)
        m.uv = rng.uniform(-3, 3, (n, 2)).astype(np.float32)
        m.bsdf = S.Bsdf(type=S.DIFFUSE, diffuse=texs[i % 3]) if i % 2 else S.Bsdf(type=S.SUBSTRATE, diffuse=texs[i % 3], specular=S.const_color((0.04, 0.04, 0.04)), distribution=S.MF_GGX, alpha_u=0.2, alpha_v=0.3)
        if kind == 0: m.uv[rng.integers(0, n)] = np.nan
        if kind == 1: m.uv[rng.integers(0, n)] = (np.inf, -np.inf)
        if kind == 2: m.uv *= np.float32(1e30)
        if kind == 4 and m.normals is not None: m.normals = m.normals.copy(); m.normals[rng.integers(0, n)] = np.nan
        if kind == 5: m.bsdf = S.Bsdf(type=S.METAL, distribution=S.MF_GGX, alpha_u=float("nan") if i % 2 else 0.0, alpha_v=0.0 if i % 2 else float("inf"))
    if kind >= 6:
        em = S.sky_map().copy()
        if kind == 6: em[2, 3] = np.nan
        if kind == 7: em[4, 5] = np.inf
        if kind == 8: em[:, :] = 0.0
        if kind == 9: em[1, 1] = -5.0
        sd.environment_map = em
    return sd
bad = 0
gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
for kind in range(10):
    for seed in range(2):
        sd = make(kind, seed)
        try:
            osc = orc.Scene(sd); ref, ost = osc.render(master_seed=seed, spp=2, eval_order=1, max_depth=5)
        except Exception as e:
            print(f"kind {kind} seed {seed} oracle refused: {e}"); continue
        msg = f"kind {kind} seed {seed} oracle finite {np.isfinite(ref).mean():.3f}"
        if gpu:
            try:
                scene = api.Scene(sd)
            except api.RustlightError as e:
                print(msg + " refused: " + str(e)[:100], flush=True); continue
            for pl in (1, 2):
                img, st = api.Context(scene, 0).render(api.IndependentSampler(seed).block_seeds(sd.width, sd.height), api.path_params(spp=2, max_depth=5, pipeline=pl))
                same = np.array_equal(img, ref, equal_nan=True) and all(st[k] == ost[k] for k in ("vertices", "rng_draws", "shadow_rays"))
                msg += f" pl{pl} {'same' if same else 'DIFF'}"; bad += not same
            seeds = api.IndependentSampler(seed).block_seeds(sd.width, sd.height)
            ctx = api.Context(scene, 0)
            for name, kw in (("direct", dict(spp=2, nb_bsdf_samples=1, nb_light_samples=2)), ("ao", dict(spp=2, max_distance=0.7))):
                img, st = (ctx.render_direct if name == "direct" else ctx.render_ao)(seeds, **kw)
                ref2, ost2 = (osc.render_direct if name == "direct" else osc.render_ao)(seeds=seeds, **kw)
                same = np.array_equal(img, ref2, equal_nan=True) and st["rng_draws"] == ost2["rng_draws"]
                msg += f" {name} {'same' if same else 'DIFF'}"; bad += not same
        print(msg, flush=True)
print("failures", bad)
