"""Dev: time the `direct` and `ao` integrators (k_pixel_mc) for every lib in scratch/variants."""
import glob, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
W, H = 1920, 1080
libs = sorted(glob.glob(os.path.join(ROOT, 'scratch', 'variants', '*.so')))
for lib in libs:
    if len(libs) == 1 or os.fork() == 0:       # one process per library (a single one runs in place: profilers follow it)
        api.LIB_PATH = lib
        for name, sd in (("cbox", scenes.cbox(W, H)), ("living_room", scenes.living_room(W, H))):
            ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
            for integ, kw in (("direct", dict(spp=16, nb_bsdf_samples=1, nb_light_samples=1)), ("ao", dict(spp=16))):
                best = 1e9
                for r in range(3):
                    t = time.perf_counter(); img, st = (ctx.render_direct if integ == "direct" else ctx.render_ao)(seeds, **kw); best = min(best, time.perf_counter() - t)
                print(f"{os.path.basename(lib):16s} {name:12s} {integ:7s} {best*1e3:8.1f} ms crc {zlib.crc32(img.tobytes()):08x}", flush=True)
        if len(libs) == 1: break
        os._exit(0)
    os.wait()
