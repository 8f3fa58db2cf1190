"""What one GPU of N would do in reference-order streams: shard 0 of N of the cbox 1080p frame (blocks b % N == 0), weak (128 N spp) and strong (128 spp) —
render ms, chain pass ms, evaluation ms, lanes per block and samples walked.  One GPU; the reduce is not part of it (a 25 MB device-to-device copy is timed as a stand-in)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from rustlight_amd import api, scenes
W, H = 1920, 1080
ctx = api.Context(api.Scene(scenes.cbox(W, H)), 0)
seeds = api.IndependentSampler(0).block_seeds(W, H)
fb = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
out = []
for scaling in ('weak', 'strong'):
    for n in (1, 2, 4, 8):
        spp = 128 * n if scaling == 'weak' else 128
        for mode, name in ((api.STREAM_REFERENCE_ORDER, 'reference'), (api.STREAM_PER_SAMPLE, 'per_sample')):
            pp = api.path_params(spp=spp, stream_mode=mode, shard_index=0, shard_count=n)
            best = None
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, st = ctx.render(seeds, pp, out_device_ptr=fb.data_ptr())
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
                if best is None or dt < best[0]: best = (dt, st)
            dt, st = best
            rec = {'scaling': scaling, 'n_gpus': n, 'mode': name, 'spp': spp, 'shard0_ms': round(dt, 1), 'chain_ms': round(st['ms_prepass'], 1), 'eval_ms': round(st['ms_other'], 1),
                   'spec_group': st['spec_group'], 'spec_x': round((st['spec_samples'] + st['spec_serial_samples'] + st['spec_probe_samples']) / max(1, st['camera_samples']), 2),
                   'serial_per_pixel': round(st['spec_serial_samples'] / (W * H / n), 2), 'Msamples_per_s_if_all_shards_alike': round(W * H * spp / dt / 1e3, 1)}
            out.append(rec); print(json.dumps(rec), flush=True)
a = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda'); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): a.copy_(fb)
torch.cuda.synchronize()
print(json.dumps({'d2d_copy_25MB_ms': round((time.perf_counter() - t0) / 20 * 1e3, 3)}))
