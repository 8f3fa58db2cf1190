"""dev: ONE frame rendered as K concurrent shards on one device (K contexts, one host thread each, blocks b % K == k; the framebuffers are summed: sums with zeros are exact) —
frame ms against the plain render, per stream mode.   usage: intraframe.py [reference|per_sample]"""
import os, sys, time, threading, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from rustlight_amd import api, scenes
mode = sys.argv[1] if len(sys.argv) > 1 else 'reference'
W, H, spp = 1920, 1080, 128
sm = api.STREAM_REFERENCE_ORDER if mode == 'reference' else api.STREAM_PER_SAMPLE
scene = api.Scene(scenes.cbox(W, H))
for K in (1, 2, 3, 4):
    ctxs = [api.Context(scene, 0) for _ in range(K)]
    fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device='cuda') for _ in range(K)]
    def work(k, seed):
        ctxs[k].render(api.IndependentSampler(seed).block_seeds(W, H), api.path_params(spp=spp, stream_mode=sm, shard_index=k, shard_count=K), out_device_ptr=fbs[k].data_ptr())
    def frame(seed):
        th = [threading.Thread(target=work, args=(k, seed)) for k in range(K)]
        for t in th: t.start()
        for t in th: t.join()
        out = fbs[0]
        for k in range(1, K): out = out + fbs[k]
        return out
    frame(100); frame(101)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(4): img = frame(s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print(f'{mode}: {K} concurrent shards: {dt*1e3:.1f} ms per frame, {W*H*spp/dt/1e6:.1f} Msamples/s, crc of frame 3 {zlib.crc32(img.cpu().numpy().tobytes()):08x}', flush=True)
    del ctxs, fbs
