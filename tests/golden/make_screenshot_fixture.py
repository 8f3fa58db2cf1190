"""Makes tests/golden/web_screenshot_canvas_64.npy: the 512x512 canvas of the reference's only committed render of the Cornell-box fixture
(examples/web/assets/screenshot.png — a browser screenshot of the stale WASM demo at a few spp), cropped out of the page (rows 348..859,
columns 8..519) and box-filtered 8x8 down to 64x64 sRGB values in [0, 1].  It is image DATA from the reference tree, not source.
Run in the build container (needs /root/reference and Pillow):  python tests/golden/make_screenshot_fixture.py"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
im = np.asarray(Image.open("/root/reference/examples/web/assets/screenshot.png").convert("RGB")).astype(np.float32) / 255.0
canvas = im[348:860, 8:520]
assert canvas.shape == (512, 512, 3)
small = canvas.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3)).astype(np.float32)
np.save(os.path.join(HERE, "web_screenshot_canvas_64.npy"), small)
print("saved", small.shape, small.mean(axis=(0, 1)))
