"""Dev tool: build kernel variants (extra -D flags) side by side and time them on the GPU box.
  build:  python scratch/variants.py build name1:-DX=1,-DY=2 name2:...      (every kernel translation unit is rebuilt per variant, in parallel)
  run:    python scratch/variants.py run [scene] [pipeline] [spp]           (on the GPU box; runs every built variant)"""
import os, subprocess, sys, time, glob
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "scratch", "variants")
LIBDIR = os.path.join(ROOT, "rustlight_amd", "lib")

def build(specs):
    from rustlight_amd import build as rb
    rb.build()
    os.makedirs(VDIR, exist_ok=True)
    if not os.environ.get("VKEEP"):
        for f in glob.glob(os.path.join(VDIR, "*.so")): os.remove(f)
    jobs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        only = [x for x in os.environ.get("VSRC", "").split(",") if x]      # VSRC=fused_stream.hip,wavefront.hip: rebuild only these units, the rest come from the default build
        for src in rb.HIP_SOURCES:
            obj = os.path.join(VDIR, f"{name}.{os.path.basename(src)}.o")
            if only and os.path.basename(src) not in only:
                jobs.append((name, os.path.join(LIBDIR, os.path.basename(src) + ".o"), ["true"]))
                continue
            jobs.append((name, obj, [rb.HIPCC, "--offload-arch=gfx950", *rb.COMMON, *rb.HIP_EXTRA, *flags, "-c", os.path.join(rb.CSRC, src), "-o", obj]))
    def run(j):
        r = subprocess.run(j[2], cwd="/tmp", stderr=subprocess.PIPE, text=True)
        return j[0], j[1], r.returncode, r.stderr
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex: res = list(ex.map(run, jobs))
    host = [o for o in glob.glob(os.path.join(LIBDIR, "*.cpp.o"))]
    for spec in specs:
        name = spec.partition(":")[0]
        mine = [r for r in res if r[0] == name]
        bad = [r for r in mine if r[2]]
        if bad: print(name, "FAILED\n", bad[0][3][-2000:]); continue
        subprocess.check_call([rb.HIPCC, "--offload-arch=gfx950", "-shared", "-o", os.path.join(VDIR, f"lib{name}.so"), *[r[1] for r in mine], *host, "-lz", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
        for r in mine:
            if r[1].startswith(VDIR): os.remove(r[1])
        print("built", name)

def run_one(libpath, scene, pipeline, spp):
    import numpy as np
    from rustlight_amd import api, scenes
    api.LIB_PATH = libpath
    W, H = int(os.environ.get("VW", 1920)), int(os.environ.get("VH", 1080))
    sd = {"cbox": lambda: scenes.cbox(W, H), "cbox_medium": lambda: scenes.cbox_medium(W, H, 0.5), "living_room": lambda: scenes.living_room(W, H)}[scene]()
    ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
    import zlib
    for split in [int(x) for x in os.environ.get("SPLITS", "0").split(",")]:
        best = 1e9
        for r in range(3):
            t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=spp, pipeline=pipeline, sample_split=split, numerics=int(os.environ.get("NUMERICS", 0)))); best = min(best, time.perf_counter() - t)
        print(f"{os.path.basename(libpath):28s} {scene} pl{pipeline} split{split} {best*1e3:8.1f} ms kernel {st['ms_other']:8.2f} ms {W*H*spp/best/1e6:8.0f} Msamples/s iters {st['iterations']} crc {zlib.crc32(img.tobytes()):08x}", flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "build": build(sys.argv[2:])
    elif sys.argv[1] == "one": run_one(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    else:
        scene = sys.argv[2] if len(sys.argv) > 2 else "cbox"; pl = sys.argv[3] if len(sys.argv) > 3 else "2"; spp = sys.argv[4] if len(sys.argv) > 4 else "128"
        for lib in sorted(glob.glob(os.path.join(VDIR, "*.so"))):
            subprocess.run([sys.executable, __file__, "one", lib, scene, pl, spp], timeout=900)
