import sys, time; sys.path.insert(0,'.')
import numpy as np
from rustlight_amd import scenes, api
to_world = np.asarray(scenes.CBOX_TO_WORLD, np.float32)
def run(name, sd, **kw):
    t=time.time(); ctx = api.Context(api.Scene(sd), 0); t1=time.time()
    img, st = ctx.render(api.IndependentSampler(0).block_seeds(sd.width, sd.height), api.path_params(**kw)); t2=time.time()
    print(f"{name:24s} ctx {t1-t:6.2f}s render {t2-t1:6.2f}s iters {st['iterations']} launches {st['kernel_launches']}", flush=True)
run("warm", scenes.cbox(16,16), spp=1)
run("empty", scenes.SceneData(20, 12, 40.0, 0, to_world, False, []), spp=2, strategy=1)
run("sky", scenes.SceneData(20, 12, 40.0, 0, to_world, False, [], environment=(0.25, 0.5, 1.0)), spp=2)
for (w,h) in ((1,1),(17,1),(1,33)):
    for pl in (1,2):
        run(f"cbox {w}x{h} pl{pl}", scenes.cbox(w,h), spp=1, pipeline=pl)
