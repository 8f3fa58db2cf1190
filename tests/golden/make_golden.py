"""Generates the committed golden fixtures from the CPU oracle (run from the repo root).

The reference cannot run here (Rust, no toolchain) and ships no fixtures for this path
(SURVEY.md F4/F5), so these vectors pin the *oracle* against silent drift and let the GPU box
self-check without it:  python tests/golden/make_golden.py
  cbox_64x64_4spp_seed0_{reference_order,per_sample}.npy   64x64x3 f32 renders (recursive eval order)
  draws_block0_seed0.npy                                   first 64 f32 draws of block 0's sampler
  trace_cbox_kat.npz                                       512 rays + (t, u, v, mesh, tri) hits
  pixel_kat.npz                                            per-sample radiance / draw counts of 32 camera samples
  features.npz                                             40x32 forward-order renders (eval_order = 1, what the kernels
                                                           reproduce bit for bit) of one scene per widened feature
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from rustlight_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def feature_cases():
    """name -> (SceneData, integrator, kwargs): shared with tests/test_gpu_parity.py::test_golden_feature_renders."""
    fog = scenes.cbox_medium(40, 32, 0.5, 0.05, g=0.4)
    return {
        "medium_hg": (fog, "path", dict(spp=2, max_depth=8)),
        "mixed_materials": (scenes.living_room(40, 32, n_spheres=12, tess=8), "path", dict(spp=2)),
        "other_emitters": (scenes.cbox_other_lights(40, 32), "path", dict(spp=2)),
        "environment_map": (scenes.sky_scene(40, 32, keep_area_light=True), "path", dict(spp=2)),
        "light_tree": (scenes.many_lights(40, 32, 3, glowing_spheres=2), "path", dict(spp=2)),
        "light_tree_direct": (scenes.many_lights(40, 32, 3, glowing_spheres=2), "direct", dict(spp=2)),
        "ao": (scenes.cbox(40, 32), "ao", dict(spp=2, max_distance=0.5)),
        "reference_order": (scenes.cbox(40, 32), "path", dict(spp=2, stream_mode=0)),
    }


def main():
    feats = {}
    for name, (fsd, integ, kw) in feature_cases().items():
        fsc = orc.Scene(fsd)
        if integ == "path":
            img, _ = fsc.render(master_seed=7, eval_order=1, **kw)
        elif integ == "direct":
            img, _ = fsc.render_direct(master_seed=7, **kw)
        else:
            img, _ = fsc.render_ao(master_seed=7, **kw)
        feats[name] = img
        print(name, float(img.mean()))
    np.savez_compressed(os.path.join(HERE, "features.npz"), **feats)
    sd = scenes.cbox(64, 64)
    sc = orc.Scene(sd)
    for mode, name in ((0, "reference_order"), (1, "per_sample")):
        img, st = sc.render(master_seed=0, spp=4, stream_mode=mode, eval_order=0)
        np.save(os.path.join(HERE, f"cbox_64x64_4spp_seed0_{name}.npy"), img)
        print(name, st)
    seeds = orc.block_seeds(0, 64, 64)
    r = orc.Rng(int(seeds[0]))
    np.save(os.path.join(HERE, "draws_block0_seed0.npy"), np.array([r.next_f32() for _ in range(64)], np.float32))
    rng = np.random.default_rng(7)
    o = rng.uniform(-0.9, 0.9, (512, 3)).astype(np.float32)
    o[:, 1] += 1.0
    d = rng.normal(size=(512, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    t, u, v, m, tr = sc.trace(o, d)
    np.savez(os.path.join(HERE, "trace_cbox_kat.npz"), o=o, d=d, t=t, u=u, v=v, mesh=m, tri=tr)
    rgb, draws, nv = [], [], []
    for i in range(32):
        rr = orc.Rng(1000 + i)
        c, nd, v_, _ = sc.compute_pixel(8 + i, 40, rr)
        rgb.append(c); draws.append(nd); nv.append(v_)
    np.savez(os.path.join(HERE, "pixel_kat.npz"), rgb=np.array(rgb, np.float32), draws=np.array(draws), vertices=np.array(nv))


if __name__ == "__main__":
    main()
