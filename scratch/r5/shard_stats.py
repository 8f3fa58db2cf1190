"""Cycle shares of k_stream_spec (timers build) on shard 0 of N of the cbox 1080p frame: python shard_stats.py <lib> <N> <spp>"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from rustlight_amd import api, scenes
api.LIB_PATH = sys.argv[1]
n, spp = int(sys.argv[2]), int(sys.argv[3])
ctx = api.Context(api.Scene(scenes.cbox(1920, 1080)), 0)
seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
pp = api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER, shard_index=0, shard_count=n)
ctx.render(seeds, pp)
os.environ["RL_SPEC_STATS"] = "1"
t = time.perf_counter(); _, st = ctx.render(seeds, pp); dt = time.perf_counter() - t
print(f"shard 0 of {n} at {spp} spp: {dt*1e3:.1f} ms, chain {st['ms_prepass']:.1f}, exposed eval {st['ms_other']:.1f}, group {st['spec_group']}", flush=True)
