"""Dev tool: `ao` / `direct` in reference-order streams at 1080p (two passes vs the single-pass walk)."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, sd in (("cbox", scenes.cbox(1920, 1080)), ("living_room", scenes.living_room(1920, 1080))):
    ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    for kind in ("ao", "direct"):
        for mode, mname in ((api.STREAM_PER_SAMPLE, "per_sample"), (api.STREAM_REFERENCE_ORDER, "reference")):
            best = None
            for r in range(2):
                t = time.perf_counter(); img, st = (ctx.render_ao if kind == "ao" else ctx.render_direct)(seeds, spp=spp, stream_mode=mode); dt = time.perf_counter() - t
                if best is None or dt < best[0]: best = (dt, st)
            print(f"{name:12s} {kind:6s} {mname:10s} {os.environ.get('RL_REF_SINGLE_PASS', ''):2s} {best[0]*1e3:9.1f} ms  chain {best[1]['ms_prepass']:8.1f}  eval {best[1]['ms_other']:8.1f}  {1920*1080*spp/best[0]/1e6:8.1f} Msamples/s crc {zlib.crc32(img.tobytes()):08x}", flush=True)
