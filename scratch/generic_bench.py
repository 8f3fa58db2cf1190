"""Dev: mixed-material scenes through both pipelines for every lib in scratch/variants (or the product lib if none)."""
import glob, os, sys, time, zlib
sys.path.insert(0, '.')
from rustlight_amd import api, scenes
W, H = 1920, 1080
def mixed_cbox():
    sd = scenes.cbox(W, H)
    mats = scenes.living_room_materials()
    for i, m in enumerate(sd.meshes):
        if m.emission is None: m.bsdf = mats[i % len(mats)]
    return sd
libs = sorted(glob.glob('scratch/variants/*.so')) or [None]
pls = [int(x) for x in os.environ.get("PIPELINES", "1,2").split(",")]
for lib in libs:
    if os.fork() == 0:
        if lib: api.LIB_PATH = lib
        for name, sd, spp in (("living_room", scenes.living_room(W, H), 32), ("living_small", scenes.living_room(W, H, n_spheres=27, tess=10), 32), ("mixed_cbox", mixed_cbox(), 32)):
            ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
            for pl in pls:
                best = 1e9
                for r in range(3):
                    t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=spp, pipeline=pl)); best = min(best, time.perf_counter() - t)
                print(f"{os.path.basename(lib or 'product'):14s} {name:14s} tris {sd.n_triangles:7d} pl{pl} {best*1e3:8.1f} ms {W*H*spp/best/1e6:7.0f} Msamples/s crc {zlib.crc32(img.tobytes()):08x}", flush=True)
        os._exit(0)
    os.wait()
