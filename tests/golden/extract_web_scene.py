"""The only scene text in the reference tree that somebody else authored: the <textarea id='scene'> of examples/web/index.html (lines 9-43), the PBRT
Cornell box of rustlight's WASM demo (a pbrt-v3 exporter's formatting: `Integrator`, `Sampler "sobol"`, `PixelFilter`, `"string filename"`, trailing blanks,
`1.74846e-007` exponents, `-0`).  This script copies that text VERBATIM (it is scene data, not source) into tests/golden/web_cbox_verbatim.pbrt;
tests/test_loaders.py feeds the file to rl_scene_load_pbrt and demands the in-memory fixture's BVH, camera, emitter table and image.
Run in the build container:  python tests/golden/extract_web_scene.py"""
import os
import re

SRC = "/root/reference/examples/web/index.html"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "web_cbox_verbatim.pbrt")

html = open(SRC, encoding="utf-8").read()
m = re.search(r"<textarea id='scene'>(.*?)</textarea>", html, re.S)
assert m, "textarea not found"
with open(OUT, "w", encoding="utf-8", newline="") as f:
    f.write(m.group(1))
print(OUT, len(m.group(1)), "bytes")
