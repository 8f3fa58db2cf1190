"""Dev: NaN / inf / zero-area / coincident / 1e30 / 1e-30 geometry added to the Cornell box, HIP path vs oracle (bits + counters)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from rustlight_amd import api, scenes
from oracle import orc
def adv_scene(kind, seed):
    rng = np.random.default_rng(seed)
    sd = scenes.cbox(24, 20)
    nt = 30
    c = rng.uniform(-0.8, 0.8, (nt, 1, 3)); c[:, :, 1] += 1.0
    v = (c + rng.uniform(-0.3, 0.3, (nt, 3, 3))).astype(np.float32).reshape(-1, 3)
    if kind == 0: v[rng.integers(0, len(v), 6)] = np.nan
    if kind == 1: v[rng.integers(0, len(v), 6), rng.integers(0, 3, 6)] = np.inf
    if kind == 2: v[:] = np.repeat(v[::3], 3, 0)                 # zero-area triangles
    if kind == 3: v *= np.float32(1e30)
    if kind == 4: v *= np.float32(1e-30)
    if kind == 5: v[:] = np.float32(0.5)                          # all coincident
    idx = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
    sd.meshes.append(scenes.MeshData("adv", v, idx, None, None, scenes.matte((0.6, 0.6, 0.6)), emission=(2.0, 2.0, 2.0) if seed % 2 else None))
    return sd
if __name__ == "__main__":
    gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
    bad = 0
    for kind in range(6):
        for seed in range(3):
            sd = adv_scene(kind, seed)
            t = time.time()
            ref, ost = orc.Scene(sd).render(master_seed=seed, spp=2, eval_order=1, max_depth=6)
            msg = f"kind {kind} seed {seed} oracle {time.time()-t:.2f}s finite {np.isfinite(ref).mean():.3f}"
            if gpu:
                try:
                    scene = api.Scene(sd)
                except api.RustlightError as e:
                    print(msg + " refused: " + str(e)[:90], flush=True); continue
                for pl in (1, 2):
                    img, st = api.Context(scene, 0).render(api.IndependentSampler(seed).block_seeds(sd.width, sd.height), api.path_params(spp=2, max_depth=6, pipeline=pl))
                    same = np.array_equal(img, ref, equal_nan=True) and all(st[k] == ost[k] for k in ("vertices", "rng_draws", "shadow_rays"))
                    msg += f" pl{pl} {'same' if same else 'DIFF'}"; bad += not same
            print(msg, flush=True)
    print("failures", bad)
