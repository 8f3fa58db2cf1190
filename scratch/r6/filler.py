"""Dev experiment (round 6): does ANY concurrent activity on another queue speed up the tail of the chain pass?  Renders cbox + medium in reference-order streams
while a host thread keeps a low-priority torch stream busy with small element-wise kernels (mode 'light') or large ones (mode 'heavy'), or not at all ('none')."""
import os, sys, threading, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rustlight_amd import api, scenes
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
W, H = 1920, 1080
ctx = api.Context(api.Scene(scenes.cbox_medium(W, H, 0.5)), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
stop = False
def fill():
    s = torch.cuda.Stream(priority=0)
    n = {"light": 1 << 16, "heavy": 1 << 26}.get(mode, 1 << 16)
    x = torch.zeros(n, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(64): x.add_(1.0)
            s.synchronize()
pp = api.path_params(spp=128, stream_mode=api.STREAM_REFERENCE_ORDER)
ctx.render(seeds, pp)
th = None
if mode != "none":
    th = threading.Thread(target=fill); th.start(); time.sleep(0.5)
for r in range(3):
    t = time.perf_counter(); img, st = ctx.render(seeds, pp); dt = time.perf_counter() - t
    print(f"filler={mode:6s} no_overlap={os.environ.get('RL_NO_OVERLAP','0')} total {dt*1e3:8.1f} ms chain {st['ms_prepass']:8.1f} eval_tail {st['ms_other']:6.1f} crc {zlib.crc32(img.tobytes()):08x}", flush=True)
stop = True
if th: th.join()
