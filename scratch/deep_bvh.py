"""Dev: 20 000 coincident triangles (a BVH the SAH cannot split: very deep) — small image renders, a 1080p image must fail cleanly if the
overflow stack does not fit."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from rustlight_amd import api, scenes
nt = 20000
for (w, h) in ((16, 16), (1920, 1080)):
    sd = scenes.cbox(w, h)
    v = np.tile(np.asarray([[0.1, 0.5, 0.1], [0.2, 0.5, 0.1], [0.1, 0.6, 0.1]], np.float32), (nt, 1))
    sd.meshes.append(scenes.MeshData("deep", v, np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3), None, None, scenes.matte((0.5, 0.5, 0.5))))
    t = time.time(); ctx = api.Context(api.Scene(sd), 0); print("context", round(time.time() - t, 1), "s", flush=True)
    for pl in (1, 2):
        try:
            t = time.time(); img, st = ctx.render(api.IndependentSampler(0).block_seeds(w, h), api.path_params(spp=1, max_depth=3, pipeline=pl))
            print((w, h), "pipeline", pl, "ok", round(time.time() - t, 2), "s mean", float(img.mean()), flush=True)
        except api.RustlightError as e:
            print((w, h), "pipeline", pl, "error:", str(e)[:120], flush=True)
