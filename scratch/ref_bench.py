"""Dev tool: time RL_STREAM_REFERENCE_ORDER renders (two-pass form: ms_prepass = k_stream_chain, ms_other = the per-sample kernel).
  python scratch/ref_bench.py [scene] [spp] [lib]      env: RL_ITEM_SHIFT, RL_REF_SINGLE_PASS, VW / VH"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "cbox"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if len(sys.argv) > 3: api.LIB_PATH = sys.argv[3]
W, H = int(os.environ.get("VW", 1920)), int(os.environ.get("VH", 1080))
sd = {"cbox": lambda: scenes.cbox(W, H), "cbox_medium": lambda: scenes.cbox_medium(W, H, 0.5), "living_room": lambda: scenes.living_room(W, H)}[scene]()
ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
best = None
for r in range(int(os.environ.get("REPS", 2))):
    t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER, numerics=int(os.environ.get("NUMERICS", 0)))); dt = time.perf_counter() - t
    if best is None or dt < best[0]: best = (dt, st)
dt, st = best
tag = " ".join(f"{k}={os.environ[k]}" for k in ("RL_ITEM_SHIFT", "RL_REF_SINGLE_PASS", "NUMERICS") if k in os.environ)
print(f"{os.path.basename(api.LIB_PATH) if len(sys.argv) > 3 else 'default':24s} {scene} {W}x{H}x{spp} {tag:24s} total {dt*1e3:8.1f} ms  chain {st['ms_prepass']:8.1f}  eval {st['ms_other']:7.1f}  {W*H*spp/dt/1e6:7.1f} Msamples/s  V/sample {st['vertices']/st['camera_samples']:.3f} crc {zlib.crc32(img.tobytes()):08x}", flush=True)
