"""dev: the headline frame (cbox 1080p x 128 spp, per-sample streams) with 1 / 2 / 4 / 8 lanes per pixel (`sample_split`) — is the persistent kernel's tail worth more workgroups?"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from rustlight_amd import api, scenes
W, H, spp = 1920, 1080, 128
ctx = api.Context(api.Scene(scenes.cbox(W, H)), 0)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
for split in (0, 1, 2, 4, 8):
    pp = api.path_params(spp=spp, stream_mode=api.STREAM_PER_SAMPLE, sample_split=split)
    for s in (100, 101): ctx.render(api.IndependentSampler(s).block_seeds(W, H), pp, out_device_ptr=fb.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(5): _, st = ctx.render(api.IndependentSampler(s).block_seeds(W, H), pp, out_device_ptr=fb.data_ptr())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'sample_split {split}: {dt*1e3:.2f} ms per frame, kernel {st["ms_other"]:.2f} ms, crc {zlib.crc32(fb.cpu().numpy().tobytes()):08x}', flush=True)
