"""Dev tool (round 6): reference-order render of shard 0 of N at 128 N spp (the per-rank workload of bench.py --gpus N), or of a whole scene, timed; used with scratch/r6/spin.hip beside it.
   python scratch/r6/shard_ref.py <scene> <n_shards> [reps]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rustlight_amd import api, scenes
scene, n = sys.argv[1], int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W, H = 1920, 1080
sd = {"cbox": lambda: scenes.cbox(W, H), "cbox_medium": lambda: scenes.cbox_medium(W, H, 0.5), "living_room": lambda: scenes.living_room(W, H)}[scene]()
ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
pp = api.path_params(spp=128 * n, stream_mode=api.STREAM_REFERENCE_ORDER, shard_index=0, shard_count=n)
ctx.render(seeds, pp)
for r in range(reps):
    t = time.perf_counter(); img, st = ctx.render(seeds, pp); dt = time.perf_counter() - t
    print(f"{scene} shard 0 of {n} at {128 * n} spp: total {dt*1e3:8.1f} ms chain {st['ms_prepass']:8.1f} eval_tail {st['ms_other']:6.1f} spec_group {st['spec_group']} crc {zlib.crc32(img.tobytes()):08x}", flush=True)
