"""Writes tests/golden/jpeg_fixture.npz: a few small JPEG files (bytes) made with Pillow (libjpeg-turbo) in the authoring container and the
RGB pixels Pillow decodes them to — the known answers for the from-scratch baseline JPEG reader (csrc/host/meshio.cpp: read_jpeg).
usage: python tests/golden/make_jpeg_fixture.py"""
import io, os
import numpy as np
from PIL import Image

rng = np.random.default_rng(0)
H, W = 37, 51
yy, xx = np.mgrid[0:H, 0:W]
img = np.stack([128 + 100 * np.sin(xx / 9.0), 128 + 100 * np.cos(yy / 7.0), (xx * 3 + yy * 2) % 256], -1)
img[10:22, 14:30] = [255, 0, 0]
img[3:9, 36:48] = [0, 0, 255]
img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
out = {}
for name, kw, src in (("yuv444_q90", dict(quality=90, subsampling=0), img), ("yuv422_q75_opt", dict(quality=75, subsampling=1, optimize=True), img),
                      ("yuv420_q60_rst", dict(quality=60, subsampling=2, restart_marker_blocks=2), img), ("gray_q80", dict(quality=80), img[:, :, 1]),
                      ("progressive_q80", dict(quality=80, subsampling=2, progressive=True), img)):
    b = io.BytesIO()
    Image.fromarray(src).save(b, "JPEG", **kw)
    out[name + "_file"] = np.frombuffer(b.getvalue(), np.uint8)
    out[name + "_rgb"] = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"), np.uint8)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_fixture.npz"), **out)
print({k: v.shape for k, v in out.items()})
