"""Dev: time the `direct` and `ao` integrators (k_pixel_mc) for every lib in scratch/variants."""
import glob, os, sys, time, zlib
sys.path.insert(0, '.')
from rustlight_amd import api, scenes
W, H = 1920, 1080
for lib in sorted(glob.glob('scratch/variants/*.so')):
    if os.fork() == 0:
        api.LIB_PATH = lib
        for name, sd in (("cbox", scenes.cbox(W, H)), ("living_room", scenes.living_room(W, H))):
            ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
            for integ, kw in (("direct", dict(spp=16, nb_bsdf_samples=1, nb_light_samples=1)), ("ao", dict(spp=16))):
                best = 1e9
                for r in range(3):
                    t = time.perf_counter(); img, st = (ctx.render_direct if integ == "direct" else ctx.render_ao)(seeds, **kw); best = min(best, time.perf_counter() - t)
                print(f"{os.path.basename(lib):16s} {name:12s} {integ:7s} {best*1e3:8.1f} ms crc {zlib.crc32(img.tobytes()):08x}", flush=True)
        os._exit(0)
    os.wait()
