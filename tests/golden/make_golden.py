"""Generates the committed golden fixtures from the CPU oracle (run from the repo root).

The reference cannot run here (Rust, no toolchain) and ships no fixtures for this path
(SURVEY.md F4/F5), so these vectors pin the *oracle* against silent drift and let the GPU box
self-check without it:  python tests/golden/make_golden.py
  cbox_64x64_4spp_seed0_{reference_order,per_sample}.npy   64x64x3 f32 renders (recursive eval order)
  draws_block0_seed0.npy                                   first 64 f32 draws of block 0's sampler
  trace_cbox_kat.npz                                       512 rays + (t, u, v, mesh, tri) hits
  pixel_kat.npz                                            per-sample radiance / draw counts of 32 camera samples
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from rustlight_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
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
