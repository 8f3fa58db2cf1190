#!/usr/bin/env python
"""bench.py — Msamples/s of the `path` hot path on BASELINE.json configs[1]:
cbox, 1920x1080, 128 spp per GPU, diffuse-only, per-sample stream mode, N x MI355X (pixel-tile shards +
one RCCL framebuffer reduce).  One "step" = one full render (W*H*spp camera samples).

Timed region: rl_render_path (tile seeding -> kernels -> framebuffer in HBM) + the RCCL reduce
(N > 1) + the framebuffer download to pinned host memory on rank 0.  Scene construction, BVH build and upload are
outside, as in the reference (integrators/mod.rs:280 vs 324-334).

`python bench.py --gpus N` starts its own N ranks (re-executes itself under torch.distributed.run) unless it already runs
under a launcher (WORLD_SIZE set).  On a box with fewer GPUs than ranks the ranks share devices and the reduce goes
through gloo — a plumbing mode, flagged in the JSON (`devices_shared`), never a scaling measurement.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_path_fused on this workload): algorithmic bytes
per launch / mean launch duration from HIP events on the render stream.  `cpu_baseline` times the
CPU oracle (a C++ restatement of rustlight's path integrator — NOT rustlight itself) on a bounded
sample of the same workload on the host cores."""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC; the runtime reads this when it initialises, i.e. before the first torch.cuda call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=128, help="samples per pixel PER GPU (weak scaling: the render uses spp x N)")
    ap.add_argument("--scene", default="cbox", choices=["cbox", "cbox_medium", "living_room"])
    ap.add_argument("--tris", type=int, default=0, help="living_room only: tessellate the spheres finer until the scene has at least this many triangles "
                    "(SURVEY 8(d): >= 4 M puts nodes + triangles past the 256 MB Infinity Cache)")
    ap.add_argument("--pool", type=int, default=0)
    ap.add_argument("--pipeline", default="auto", choices=["auto", "wavefront", "fused"])
    ap.add_argument("--stream-mode", default="per_sample", choices=["per_sample", "reference"],
                    help="per_sample: one lane per pixel / sample (throughput decomposition); reference: one lane per 16x16 block, rustlight's own stream assignment")
    ap.add_argument("--numerics", default="exact", choices=["exact", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed single-GPU re-render that checks the N-GPU image CRC")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _respawn_under_launcher(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same args>`."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    os.execvp(cmd[0], cmd)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_launcher(args)

    import torch
    import torch.distributed as dist

    from rustlight_amd import api, provenance, scenes
    from rustlight_amd import distributed as rd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback path exists)")
    n_dev = torch.cuda.device_count()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    # fewer devices than ranks (or RL_BENCH_SHARE_DEVICE=1): ranks share devices, RCCL cannot (one rank per device), so the reduce goes through gloo
    shared = env_world > n_dev or bool(os.environ.get("RL_BENCH_SHARE_DEVICE"))
    backend = os.environ.get("RL_BENCH_BACKEND") or ("gloo" if shared else "nccl")
    rank, world, local_rank = rd.init_from_env(args.gpus, backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} is running with WORLD_SIZE={world}: launch it with --nproc-per-node {args.gpus} (or without a launcher)")
    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)

    if args.scene == "cbox":
        sd = scenes.cbox(args.width, args.height)
        workload = f"cbox {args.width}x{args.height}x{args.spp}spp-per-GPU diffuse path (BASELINE configs[1]; configs[3] at 8 GPUs)"
    elif args.scene == "cbox_medium":
        sd = scenes.cbox_medium(args.width, args.height, 0.5)
        workload = f"cbox + homogeneous medium sigma_s=0.5 {args.width}x{args.height}x{args.spp}spp (BASELINE configs[4])"
    else:
        tess = 32
        while 256 * 2 * tess * (tess - 1) + 14 < args.tris: tess += 1
        sd = scenes.living_room(args.width, args.height, tess=tess)
        workload = (f"living-room-class synthetic stand-in ({sd.n_triangles} tris, 6 BSDF types; the real pbrt-v3 living-room is not available) "
                    f"{args.width}x{args.height}x{args.spp}spp (BASELINE configs[2])")
    scene = api.Scene(sd)
    ctx = api.Context(scene, device_index)          # BVH build + upload: untimed, like the reference (mod.rs:280)
    fb = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=dev)
    host_fb = torch.zeros((args.height, args.width, 3), dtype=torch.float32).pin_memory() if rank == 0 else None
    # ONE explicit stream for everything a step enqueues (render, reduce, download): torch's default stream has handle 0, which the C-ABI
    # reads as "use the context's own stream" — the next step's framebuffer memset would then race the previous step's reduce / download
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    assert stream != 0

    # weak scaling: the image is fixed and spp grows with the GPU count (128 spp at N=1 ... 1024 spp at N=8 =
    # BASELINE configs[3]), so every GPU always traces W*H*128 camera samples per step.
    spp_total = args.spp * world
    stream_mode = api.STREAM_PER_SAMPLE if args.stream_mode == "per_sample" else api.STREAM_REFERENCE_ORDER
    pipeline = {"auto": 0, "wavefront": 1, "fused": 2}[args.pipeline]
    numerics = {"exact": 0, "fast": 1}[args.numerics]

    def params(shard_index, shard_count):
        return api.path_params(spp=spp_total, shard_index=shard_index, shard_count=shard_count, pool_slots=args.pool, pipeline=pipeline,
                               stream_mode=stream_mode, numerics=numerics)

    def step(seed):
        seeds = api.IndependentSampler(seed).block_seeds(args.width, args.height)     # same master stream on every rank
        _, st = ctx.render(seeds, params(rank, world), out_device_ptr=fb.data_ptr(), stream=stream)
        if world > 1:
            rd.reduce_framebuffer(fb)                                                   # one RCCL reduce over xGMI
        if rank == 0:
            host_fb.copy_(fb, non_blocking=True)                                        # framebuffer download (inside the timed region, §8(d))
        return st

    for w in range(args.warmup):
        step(1000 + w)
    rd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = []
    for s in range(args.steps):
        stats.append(step(s))
    rd.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = rd.max_over_ranks(dt)
    host_img = host_fb.numpy() if rank == 0 else None

    samples_per_step = args.width * args.height * spp_total
    value = samples_per_step * args.steps / dt / 1e6
    agg = {k: sum(s[k] for s in stats) for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "iterations", "n_extend_launches", "kernel_launches")}
    ms = {k: sum(s[k] for s in stats) for k in ("ms_raygen", "ms_extend", "ms_shade", "ms_shadow", "ms_other")}
    agg_all = rd.sum_over_ranks(agg)
    ranks = rd.gather_objects({"rank": rank, "device": device_index, "name": torch.cuda.get_device_name(device_index), "pid": os.getpid(),
                               "kernel_ms_per_step": (ms["ms_other"] or sum(ms.values())) / max(1, args.steps)})

    if rank == 0:
        # ---- the N-GPU image must be the 1-GPU image, bit for bit (sums with zeros are exact): re-render the last step's
        # frame on this GPU alone, untimed, and compare CRCs
        crc = zlib.crc32(host_img.tobytes())
        crc_single = None
        if world > 1 and not args.no_verify:
            seeds = api.IndependentSampler(args.steps - 1).block_seeds(args.width, args.height)
            single, _ = ctx.render(seeds, params(0, 1))
            crc_single = zlib.crc32(single.tobytes())
            if crc_single != crc:       # reported in the JSON (crc_match: false) rather than raised: the other ranks are waiting in a barrier
                print(f"ERROR: {world}-GPU image CRC {crc:08x} != 1-GPU image CRC {crc_single:08x}", file=sys.stderr)
        # ---- roofline.  Algorithmic bytes per unit:
        #   k_raygen 108 B/camera sample, k_extend 44 B/ray, k_shade 280 B/vertex, k_shadow 72 B/shadow ray,
        #   k_path_fused (all four stages in one persistent launch): the whole-pipeline figure
        #   248 B/camera sample + 352 B/expanded vertex + 12 B/pixel.
        pix = args.width * args.height / world
        pipe_bytes = 248 * agg["camera_samples"] + 352 * agg["vertices"] + 12 * pix * args.steps
        fused = ms["ms_other"] > 0.0
        names = {"ms_raygen": "k_raygen", "ms_extend": "k_extend", "ms_shade": "k_shade", "ms_shadow": "k_shadow", "ms_other": "k_path_fused"}
        per_unit = {"ms_raygen": 108, "ms_extend": 44, "ms_shade": 280, "ms_shadow": 72}
        units = {"ms_raygen": agg["camera_samples"], "ms_extend": agg["extension_rays"], "ms_shade": agg["extension_rays"], "ms_shadow": agg["shadow_rays"]}
        n_launch = {k: max(1, agg["n_extend_launches"]) for k in per_unit}
        kernels = {}
        for k in per_unit:
            if ms[k] > 0:
                avg = ms[k] / n_launch[k]
                gbs = per_unit[k] * units[k] / n_launch[k] / (avg * 1e-3) / 1e9
                kernels[names[k]] = {"avg_launch_ms": avg, "launches": n_launch[k], "algorithmic_bytes_per_unit": per_unit[k], "achieved_GBps": gbs, "frac": gbs / 8000.0}
        if fused:
            avg = ms["ms_other"] / args.steps
            gbs = pipe_bytes / args.steps / (avg * 1e-3) / 1e9
            kernels["k_path_fused"] = {"avg_launch_ms": avg, "launches": args.steps, "algorithmic_bytes_per_unit": "248/sample + 352/vertex + 12/pixel",
                                       "achieved_GBps": gbs, "frac": gbs / 8000.0}
        dominant = max(kernels, key=lambda k: kernels[k]["avg_launch_ms"] * kernels[k]["launches"])
        # PMC numbers are NOT collected in this run (rocprofv3 --pmc needs its own passes: scratch/pmc_collect.sh).  They are quoted
        # only when they were collected on this very kernel source (hash of csrc/kernels + flags), with their provenance.
        src_hash = provenance.kernel_source_hash()
        traffic, pmc_entry = None, None
        try:
            live = json.load(open(os.path.join(ROOT, "profiles", "pmc_live.json")))
            key = f"{args.scene}{':tris' + str(args.tris) if args.tris else ''}:{args.width}x{args.height}x{args.spp}:{args.stream_mode}:{args.numerics}"
            e = live.get(key)
            if e and e.get("kernel") == dominant and world == 1:
                pmc_entry = dict(e, current=(e.get("kernel_src_hash") == src_hash))
                if pmc_entry["current"]:
                    traffic = e.get("hbm_bytes_per_launch")
        except Exception:
            pass
        roofline = {"bound": (pmc_entry or {}).get("bound", "valu"), "kernel": dominant,
                    "achieved": kernels[dominant]["achieved_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": kernels[dominant]["frac"],
                    "note": "achieved = ALGORITHMIC bytes (SURVEY §8(d): state a wavefront pipeline would stream) / kernel time; the fused kernel keeps that state in "
                            "registers / LDS, so measured HBM traffic (`traffic`) is far below it and the kernel is bound by VALU issue + latency, not by HBM",
                    "traffic": traffic, "avg_launch_ms": kernels[dominant]["avg_launch_ms"],
                    "launches": kernels[dominant]["launches"], "kernels": kernels,
                    "pipeline_algorithmic_GBps": pipe_bytes / dt / 1e9, "pipeline_frac": pipe_bytes / dt / 1e9 / 8000.0,
                    "rays_per_s": (agg_all["extension_rays"] + agg_all["shadow_rays"]) / dt,
                    "kernel_src_hash": src_hash, "pmc": pmc_entry}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import orc
            orc.use_timing_build()        # -O3 / libm / FMA build of the same restatement; never the checker
            cw, ch, cspp = args.width, args.height, max(1, args.spp // 4)   # bounded sample of the same workload: same scene and resolution, 1/4 of the spp (~10 s of CPU work over three passes)
            osc = orc.Scene(scenes.cbox(cw, ch) if args.scene == "cbox" else (scenes.cbox_medium(cw, ch, 0.5) if args.scene == "cbox_medium" else sd))
            # cores this process may really use: the GPU boxes report 256 hardware threads but run under a cgroup CPU quota
            # (cpu.max = 16 CPUs); more runnable threads than quota only adds throttling
            ncpu = len(os.sched_getaffinity(0))
            try:
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
                if quota != "max":
                    ncpu = max(1, min(ncpu, int(-(-int(quota) // int(period)))))
            except (OSError, ValueError):
                pass
            osc.render(master_seed=1, spp=1, stream_mode=0, threads=ncpu)             # warm up / page in
            runs = []
            for _ in range(3):      # the host is shared and noisy: report the best of three passes (all three listed)
                t1 = time.perf_counter()
                _, ost = osc.render(master_seed=0, spp=cspp, stream_mode=0, threads=ncpu)
                runs.append(cw * ch * cspp / (time.perf_counter() - t1) / 1e6)
                if sum(cw * ch * cspp / r / 1e6 for r in runs) > 30.0:
                    break
            cpu = {"value": max(runs), "unit": "Msamples/s", "cores": ost["threads"], "kind": "port", "runs": [round(r, 2) for r in runs],
                   "sample": f"{args.scene} {cw}x{ch}x{cspp}spp, reference-order streams, CPU restatement of rustlight `path` (C++, -O3 timing build), {ost['threads']} threads = CPUs available to the process (affinity / cgroup quota; the host has {os.cpu_count()} hardware threads), best of {len(runs)}"}
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        out = {"metric": "Msamples/s (paths/s) at 1080p x 128spp cbox", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "spp_total": spp_total, "stream_mode": args.stream_mode, "numerics": args.numerics,
                          "pipeline": "fused (k_path_fused)" if fused else "wavefront (raygen/extend/shade/shadow)",
                          "parallelism": f"tile-shard x{world} + 1 RCCL reduce",
                          "timed_region": "rl_render_path + framebuffer reduce (N > 1) + framebuffer download to pinned host memory (SURVEY §8(d))",
                          "mean_vertices_per_sample": agg_all["vertices"] / max(1, agg_all["camera_samples"]),
                          "image_mean": float(host_img.mean())},
               "distributed": {"world_size": world, "backend": backend if world > 1 else None, "rccl_version": rccl, "devices_visible": n_dev, "devices_shared": bool(shared and world > 1),
                               "ranks": ranks, "image_crc32": f"{crc:08x}", "single_gpu_image_crc32": None if crc_single is None else f"{crc_single:08x}",
                               "crc_match": None if crc_single is None else crc_single == crc},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
        sys.stdout.flush()
    rd.barrier()
    rd.finalize()


if __name__ == "__main__":
    main()
