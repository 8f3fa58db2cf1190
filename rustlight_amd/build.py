"""Build the product library IN-TREE: rustlight_amd/lib/librustlight_amd.so (HIP kernels for gfx950 +
host C++), so it travels to the GPU box with the repo snapshot.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "librustlight_amd.so")
BIN = os.path.join(LIB_DIR, "rustlight-amd")

HIP_SOURCES = ["kernels/wavefront.hip", "kernels/fused_lds.hip", "kernels/fused_stream.hip", "kernels/fusedq_lds.hip", "kernels/fusedq_stream.hip", "kernels/fused_lds_fast.hip", "kernels/fused_stream_fast.hip", "kernels/chain_lds.hip", "kernels/chain_stream.hip", "kernels/chain_lds_fast.hip", "kernels/chain_stream_fast.hip", "kernels/spec_lds.hip", "kernels/spec_stream.hip", "kernels/shade.hip", "kernels/mc.hip", "host/multigpu.hip"]     # multigpu.hip: N device contexts + the RCCL framebuffer reduce
CXX_SOURCES = ["host/frames.cpp", "host/scene.cpp", "host/bvh.cpp", "host/io.cpp", "host/pbrt.cpp", "host/meshio.cpp", "host/mitsuba.cpp", "host/lighttree.cpp"]
# -ffp-contract=off: rustlight's f32 arithmetic is never contracted into FMAs
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-value", "-Wno-unused-function"]
# no SLP vectorizer for the kernels: it turns 3-vector math into packed v_pk_{add,mul}_f32, which issue at half rate on gfx950
# and cost register-pair moves (measured: k_path_fused 65.9 -> 62.5 ms, same bits)
HIP_EXTRA = ["-fno-slp-vectorize", "-Wno-bitwise-instead-of-logical"]
HIPCC = os.environ.get("HIPCC") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
# where librccl / libamdhip64 live: ROCM_PATH, else next to the compiler (<rocm>/bin/hipcc -> <rocm>/lib)
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib")
if not os.path.isdir(ROCM_LIB):
    ROCM_LIB = "/opt/rocm/lib"
CXX = os.environ.get("CXX", "g++")


def extra_hip_flags():
    """Dev builds: RL_HIP_FLAGS="-DRL_STAGE_TIMERS ..." adds flags to every kernel translation unit."""
    return [f for f in os.environ.get("RL_HIP_FLAGS", "").split() if f]


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files]
    out.append(os.path.join(HERE, "..", "include", "rustlight_amd.h"))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    deps = _deps()
    if not force and not _stale(LIB, deps) and not _stale(BIN, deps):
        return LIB
    # every translation unit is independent: compile them side by side (the kernel families take ~1 min each)
    jobs = []
    for src in HIP_SOURCES:
        obj = os.path.join(LIB_DIR, os.path.basename(src) + ".o")
        jobs.append((obj, [HIPCC, "--offload-arch=gfx950", *COMMON, *HIP_EXTRA, *extra_hip_flags(), "-c", os.path.join(CSRC, src), "-o", obj]))
    for src in CXX_SOURCES:
        obj = os.path.join(LIB_DIR, os.path.basename(src) + ".o")
        jobs.append((obj, [CXX, *COMMON, "-c", os.path.join(CSRC, src), "-o", obj]))

    def run(job):
        if verbose:
            print(" ".join(job[1]))
        subprocess.check_call(job[1])
        return job[0]

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(run, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB, *objs, "-lz", "-L" + ROCM_LIB, "-lrccl", "-Wl,-rpath," + ROCM_LIB]
    subprocess.check_call(cmd)
    # the CLI (examples/cli.rs counterpart)
    cli = os.path.join(CSRC, "host", "cli.cpp")
    if os.path.exists(cli):
        subprocess.check_call([CXX, *COMMON, cli, "-o", BIN, "-L" + LIB_DIR, "-lrustlight_amd", "-Wl,-rpath,$ORIGIN",
                               "-Wl,-rpath," + ROCM_LIB, "-L" + ROCM_LIB, "-lamdhip64"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
