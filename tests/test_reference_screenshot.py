"""A LOOSE external sanity check, labelled as such — it does not pin the oracle (parity stays "unpinned").

The only render of the Cornell-box fixture that the reference tree holds is `examples/web/assets/screenshot.png`: a browser screenshot of
the WASM demo (`examples/web/src/lib.rs`), i.e. of a STALE build — it calls the old three-argument `Camera::new` (lib.rs:156), so its image
is horizontally mirrored against `src/camera.rs:31-48` with `flip = false` — at a handful of samples per pixel, through a browser canvas,
tone-mapped `c^(1/2.2)` clamped to 1 (lib.rs:218-224).  `tests/golden/web_screenshot_canvas_64.npy` is its 512x512 canvas box-filtered to
64x64 (made by tests/golden/make_screenshot_fixture.py).

What this can check — and does: the gross restatement of scene and camera.  Wall colours on the right sides, the light where it is and
how big (17:12:4 emission clamps to white), the two boxes' silhouettes within a few pixels, the frame coverage that fov 19.5 deg gives at
aspect 1.  What it cannot check: seeds, draw order, RNG, any per-sample arithmetic."""
import os

import numpy as np

from oracle import orc
from rustlight_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_cbox_looks_like_the_references_own_screenshot(built):
    ref = np.load(os.path.join(HERE, "golden", "web_screenshot_canvas_64.npy"))
    img, _ = orc.Scene(scenes.cbox(512, 512)).render(master_seed=0, spp=8, stream_mode=1, eval_order=1, threads=max(1, min(8, os.cpu_count() or 1)))
    tm = np.minimum(1.0, np.maximum(img, 0.0)) ** (1.0 / 2.2)                     # Color::to_rgba / get_img
    ours = tm.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3))[:, ::-1]                 # mirrored: the demo's stale Camera::new
    plain = ours[:, ::-1]
    d_m, d_p = float(np.abs(ours - ref).mean()), float(np.abs(plain - ref).mean())
    assert d_m < 0.03 and d_p > 3 * d_m, (d_m, d_p)                               # measured: 0.011 mirrored, 0.099 unmirrored
    assert np.corrcoef(ours.ravel(), ref.ravel())[0, 1] > 0.98                    # measured 0.9966
    # wall colours: green on the screenshot's left, red on its right (after mirroring ours)
    for a in (ours, ref):
        left, right = a[20:50, 1:4].mean(axis=(0, 1)), a[20:50, 60:63].mean(axis=(0, 1))
        assert left[1] > 1.3 * left[0] and right[0] > 2.5 * right[1], (left, right)
    assert np.abs(ours[20:50, 1:4].mean(axis=(0, 1)) - ref[20:50, 1:4].mean(axis=(0, 1))).max() < 0.05
    assert np.abs(ours[20:50, 60:63].mean(axis=(0, 1)) - ref[20:50, 60:63].mean(axis=(0, 1))).max() < 0.05
    # the light: a white (clamped) rectangle in the same 64x64 cells, +- 1 cell
    def bbox(a):
        idx = np.argwhere(a.min(axis=-1) > 0.9)
        return np.array([idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max()])
    assert np.abs(bbox(ours) - bbox(ref)).max() <= 1, (bbox(ours), bbox(ref))
    # silhouettes and frame coverage: the luminance edges line up — shifting ours by 3 cells in any direction must fit clearly worse
    lum_o, lum_r = ours.mean(axis=-1), ref.mean(axis=-1)
    base = np.abs(lum_o[4:-4, 4:-4] - lum_r[4:-4, 4:-4]).mean()
    for dy, dx in ((3, 0), (-3, 0), (0, 3), (0, -3)):
        shifted = np.roll(lum_o, (dy, dx), axis=(0, 1))
        assert np.abs(shifted[4:-4, 4:-4] - lum_r[4:-4, 4:-4]).mean() > 1.5 * base, (dy, dx)
