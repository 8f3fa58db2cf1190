"""dev: frames in flight — K contexts of the same scene on one device, one host thread each, rendering independent frames concurrently (each context has its own stream):
aggregate Msamples/s against one context, per stream mode.   usage: inflight.py [reference|per_sample] [n_frames] [shard_count] [cbox|cbox_medium|living_room] [K,K,...]"""
import os, sys, time, threading, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from rustlight_amd import api, scenes
mode = sys.argv[1] if len(sys.argv) > 1 else 'reference'
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 1
scene_name = sys.argv[4] if len(sys.argv) > 4 else 'cbox'
Ks = [int(k) for k in sys.argv[5].split(',')] if len(sys.argv) > 5 else [1, 2, 3, 4]
W, H, spp = 1920, 1080, 128 * shards
sm = api.STREAM_REFERENCE_ORDER if mode == 'reference' else api.STREAM_PER_SAMPLE
scene = api.Scene({'cbox': lambda: scenes.cbox(W, H), 'cbox_medium': lambda: scenes.cbox_medium(W, H, 0.5), 'living_room': lambda: scenes.living_room(W, H)}[scene_name]())
for K in Ks:
    ctxs = [api.Context(scene, 0) for _ in range(K)]
    fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device='cuda') for _ in range(K)]
    pp = api.path_params(spp=spp, stream_mode=sm, shard_index=0, shard_count=shards)
    crcs = {}
    def work(k, frames):
        for f in frames:
            seeds = api.IndependentSampler(f).block_seeds(W, H)
            ctxs[k].render(seeds, pp, out_device_ptr=fbs[k].data_ptr())
            if f < 2: crcs[f] = '%08x' % zlib.crc32(fbs[k].cpu().numpy().tobytes())
    # warm-up: every context renders one frame alone
    for k in range(K): work(k, [100 + k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, list(range(k, n_frames, K)))) for k in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{scene_name} {mode} shards {shards} spp {spp}: {K} in flight: {n_frames} frames in {dt*1e3:.1f} ms = {dt*1e3/n_frames:.1f} ms per frame, {W*H*spp/shards*n_frames/dt/1e6:.1f} Msamples/s; crc of frames 0, 1: {crcs.get(0)} {crcs.get(1)}', flush=True)
    del ctxs, fbs
