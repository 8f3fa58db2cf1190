"""Repro harness for the intermittent stall of the overlapped evaluation pass: one fresh process, a few contexts, a few reference-order renders each, 300 ms patience.
Prints per render: wall ms, chain ms, exposed eval ms — a render whose chain pass cannot start beside the gate shows chain ms >> normal and '[queue] the gate ... gave up'."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ["RL_QUEUE_PATIENCE_MS"] = "300"; os.environ["RL_QUEUE_DEBUG"] = "1"
mode = sys.argv[1] if len(sys.argv) > 1 else "own"
from rustlight_amd import api, scenes
stream = 0
if mode == "torch":
    import torch
    ws = torch.cuda.Stream(); torch.cuda.set_stream(ws); stream = ws.cuda_stream
W, H, spp = 480, 270, 16
worst = 0.0
for c in range(4):
    sd = scenes.cbox_medium(W, H, 0.5) if c % 2 == 0 else scenes.cbox(W, H)
    ctx = api.Context(api.Scene(sd), 0)
    seeds = api.IndependentSampler(c).block_seeds(W, H)
    for r in range(3):
        t = time.perf_counter()
        img, st = ctx.render(seeds, api.path_params(spp=spp if c % 2 == 0 else 128, stream_mode=api.STREAM_REFERENCE_ORDER), stream=stream) if stream else ctx.render(seeds, api.path_params(spp=spp if c % 2 == 0 else 128, stream_mode=api.STREAM_REFERENCE_ORDER))
        dt = (time.perf_counter() - t) * 1e3
        worst = max(worst, dt)
        if dt > 350: print(f"  STALL? ctx {c} render {r}: {dt:.1f} ms chain {st['ms_prepass']:.1f} eval {st['ms_other']:.1f}", flush=True)
print(f"{mode}: worst render {worst:.1f} ms", flush=True)
