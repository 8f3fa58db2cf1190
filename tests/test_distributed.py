"""N > 1 path on CPU: two `gloo` ranks run rustlight_amd.distributed's sharding + the single framebuffer
sum-reduce.  The per-rank renderer is the CPU oracle (this is a test of the host plumbing — block
dealing, identical master seeds on every rank, one reduce — not of the kernels)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import orc
    from rustlight_amd import api, scenes
    from rustlight_amd import distributed as rd

    r, w, _ = rd.init_from_env(world, backend="gloo")
    assert (r, w) == (rank, world) and rd.is_dist()
    sd = scenes.cbox(80, 48)
    seeds = api.IndependentSampler(21).block_seeds(80, 48)          # every rank draws the same master stream
    shard_index, shard_count = rd.shard_of(r, w)
    img, st = orc.Scene(sd).render(seeds=seeds, spp=2, shard_index=shard_index, shard_count=shard_count, threads=2)
    fb = torch.from_numpy(img)
    rd.reduce_framebuffer(fb, dst=0)
    t = rd.max_over_ranks(float(rank + 1))
    tot = rd.sum_over_ranks({"samples": st["camera_samples"]})
    rd.barrier()
    if rank == 0:
        np.savez(out_path, img=fb.numpy(), tmax=t, samples=tot["samples"])
    rd.finalize()


def test_two_rank_gloo_shard_and_reduce(built, tmp_path):
    from oracle import orc
    from rustlight_amd import scenes

    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    full, st = orc.Scene(scenes.cbox(80, 48)).render(seeds=orc.block_seeds(21, 80, 48), spp=2)
    np.testing.assert_array_equal(got["img"], full)                  # N-rank image == 1-rank image, bitwise
    assert got["tmax"] == 2.0 and got["samples"] == st["camera_samples"] == 80 * 48 * 2


def test_single_process_helpers_are_noops():
    from rustlight_amd import distributed as rd

    assert not rd.is_dist()
    t = torch.ones(4)
    assert rd.reduce_framebuffer(t) is t and rd.max_over_ranks(3.5) == 3.5 and rd.sum_over_ranks({"a": 2}) == {"a": 2}
    assert rd.shard_of(3, 8) == (3, 8)
