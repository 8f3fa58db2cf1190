"""torch.distributed plumbing for the multi-GPU path: one process per GPU; the
16x16 blocks are dealt round-robin to ranks (b % world == rank); every rank renders its blocks into
a zeroed W x H x 3 f32 buffer and ONE sum-reduce (RCCL over xGMI, backend "nccl") merges them.
Sums with zeros are exact, so the N-GPU image is bit-identical to the 1-GPU image.  There is no
other collective on the data path."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(expected_world: int = 1, backend: str | None = None, device: "torch.device | None" = None, timeout_s: float = 600.0):
    """Join the process group the launcher described (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  `device` (backend "nccl"): the GPU this
    rank renders on — the caller has already made it current (torch.cuda.set_device) and it is handed to the group as `device_id`, so RCCL
    binds its communicator to that device eagerly instead of to whatever is current at the first collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        import datetime

        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, world, local_rank


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_of(rank: int, world: int):
    """(shard_index, shard_count) handed to rl_path_params."""
    return rank, max(1, world)


def reduce_framebuffer(fb: torch.Tensor, dst: int = 0) -> torch.Tensor:
    """The single exchange step: sum the per-rank framebuffers onto `dst` (ncclReduce over xGMI with backend "nccl").
    gloo cannot reduce device tensors: in the shared-device plumbing mode (more ranks than GPUs) the buffer is staged
    through the host — same sum, same bits."""
    if is_dist():
        if fb.is_cuda and dist.get_backend() != "nccl":
            host = fb.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            if dist.get_rank() == dst:
                fb.copy_(host)
        else:
            dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb


def barrier():
    if is_dist():
        dist.barrier()


def max_over_ranks(x: float) -> float:
    if not is_dist():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(d: dict) -> dict:
    if not is_dist():
        return dict(d)
    keys = sorted(d)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(d[k]) for k in keys], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: float(v) for k, v in zip(keys, t.tolist())}


def gather_objects(obj):
    """Every rank's `obj`, in rank order, on every rank (rank / device report of bench.py)."""
    if not is_dist():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
