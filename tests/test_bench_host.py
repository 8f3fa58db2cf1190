"""Host-side logic of bench.py that can be checked without a GPU: the CLI contract (defaults, knobs), the self-launch under
torch.distributed.run, and the provenance gate on stored counter summaries."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_are_the_baseline_workload_on_one_gpu():
    a = bench.parse_args([])
    assert (a.gpus, a.scene, a.width, a.height, a.spp) == (1, "cbox", 1920, 1080, 128)
    assert a.steps >= 1 and a.warmup >= 1 and a.steps * 0.06 < 60          # a default run is seconds, not minutes
    assert a.stream_mode == "per_sample" and a.numerics == "exact" and a.pipeline == "auto" and a.tris == 0


def test_knobs_parse():
    a = bench.parse_args(["--gpus", "8", "--steps", "5", "--warmup", "2", "--scene", "living_room", "--tris", "4000000", "--stream-mode", "reference", "--numerics", "fast"])
    assert (a.gpus, a.steps, a.warmup, a.scene, a.tris, a.stream_mode, a.numerics) == (8, 5, 2, "living_room", 4000000, "reference", "fast")
    with pytest.raises(SystemExit):
        bench.parse_args(["--scene", "nope"])


def test_self_launch_uses_the_local_rendezvous(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself as N ranks on 127.0.0.1 with the same arguments."""
    seen = {}
    monkeypatch.setattr(os, "execvp", lambda prog, argv: seen.update(prog=prog, argv=list(argv)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    bench._respawn_under_launcher(bench.parse_args(["--gpus", "4", "--steps", "2"]))
    argv = seen["argv"]
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(argv[argv.index("--master-port") + 1]) < 65536
    k = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "4", "--steps", "2"]
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"            # dmabuf IPC: RCCL across processes needs it on this pool


def test_stored_counter_keys_follow_the_workload():
    """profiles/pmc_live.json is keyed the way bench.py looks entries up (scene[:trisN]:WxHxSPP:stream:numerics)."""
    live = json.load(open(os.path.join(ROOT, "profiles", "pmc_live.json")))
    for key, e in live.items():
        scene, *rest = key.split(":")
        assert scene in ("cbox", "cbox_medium", "living_room")
        if rest[0].startswith("tris"): rest = rest[1:]
        dims, stream, numerics = rest
        w, h, spp = (int(x) for x in dims.split("x"))
        assert (w, h, spp) == (1920, 1080, 128) and stream in ("per_sample", "reference") and numerics in ("exact", "fast")
        assert e["key"] == key and len(e["kernel_src_hash"]) == 16 and e["commit"]


def test_also_records_only_on_the_default_single_gpu_run():
    """The `also` sub-records (the other BASELINE configs / stream modes, timed in the same process) ride on the run the driver makes — `python bench.py --gpus 1
    [--steps K --warmup W]` — and on nothing else: any other scene, size, stream mode, numerics, pipeline, pool, GPU count or --no-also is a single-workload run."""
    assert bench.is_default_workload(bench.parse_args([]))
    assert bench.is_default_workload(bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "3"]))
    assert bench.is_default_workload(bench.parse_args(["--no-cpu-baseline"]))
    for argv in (["--gpus", "2"], ["--scene", "living_room"], ["--stream-mode", "reference"], ["--numerics", "fast"], ["--width", "1080"], ["--spp", "64"],
                 ["--pipeline", "wavefront"], ["--pool", "4096"], ["--tris", "4000000"], ["--no-also"]):
        assert not bench.is_default_workload(bench.parse_args(argv)), argv
    a = bench.parse_args(["--scaling", "strong", "--gpus", "8"])
    assert a.scaling == "strong" and bench.parse_args([]).scaling == "weak" and a.init_timeout > 0


def test_pipeline_bytes_is_the_survey_formula():
    """algorithmic bytes = 248 B per camera sample + 352 B per expanded vertex + 12 B per pixel per pass."""
    assert bench.pipeline_bytes({"camera_samples": 10, "vertices": 7}, pixels=5, steps=2) == 248 * 10 + 352 * 7 + 12 * 5 * 2


def test_bench_golden_table_covers_the_timed_frames():
    """tests/golden/bench_crcs.json (oracle CRCs of the frames bench.py times; tests/golden/make_bench_golden.py): every frame of the default run is in it,
    keyed the way bench.oracle_crc looks it up — the headline at the driver's step count (--steps 20: seed 19) and at the default (seed 2), the `also` records
    at their three steps (seed 2), BASELINE configs[0]."""
    import bench
    for wl, w, h, spp, mode, seed in (("cbox", 1920, 1080, 128, "per_sample", 19), ("cbox", 1920, 1080, 128, "per_sample", 2), ("cbox", 1920, 1080, 128, "reference", 2),
                                      ("cbox", 1080, 1080, 128, "per_sample", 2), ("cbox_medium", 1920, 1080, 128, "per_sample", 2), ("living_room", 1920, 1080, 128, "per_sample", 2),
                                      ("cbox", 256, 256, 16, "reference", 0)):
        crc = bench.oracle_crc(wl, w, h, spp, mode, seed)
        assert crc is not None and len(crc) == 8, (wl, w, h, spp, mode, seed)
    assert bench.oracle_crc("cbox", 1920, 1080, 128, "per_sample", 7) is None        # a frame nobody rendered on the CPU: no claim
