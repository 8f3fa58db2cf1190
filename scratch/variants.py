"""Dev tool: build kernel variants (extra -D flags) side by side and time them on the GPU box.
  build:  python scratch/variants.py build name1:-DX=1,-DY=2 name2:...
  run:    python scratch/variants.py run [scene] [pipeline]      (on the GPU box; runs every built variant)"""
import os, subprocess, sys, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "scratch", "variants")
LIBDIR = os.path.join(ROOT, "rustlight_amd", "lib")

def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for f in glob.glob(os.path.join(VDIR, "*.so")): os.remove(f)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        obj = os.path.join(VDIR, name + ".o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", *flags, "-c",
               os.path.join(ROOT, "rustlight_amd/csrc/kernels/wavefront.hip"), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd, cwd="/tmp", stderr=subprocess.PIPE, text=True)))
    for name, obj, p in procs:
        err = p.communicate()[1]
        if p.returncode: print(name, "FAILED\n", err[-2000:]); continue
        host = [o for o in glob.glob(os.path.join(LIBDIR, "*.cpp.o"))]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", os.path.join(VDIR, f"lib{name}.so"), obj, *host])
        os.remove(obj)
        print("built", name)

def run_one(libpath, scene, pipeline, spp):
    sys.path.insert(0, ROOT)
    import numpy as np
    from rustlight_amd import api, scenes
    api.LIB_PATH = libpath
    W, H = 1920, 1080
    sd = {"cbox": lambda: scenes.cbox(W, H), "cbox_medium": lambda: scenes.cbox_medium(W, H, 0.5), "living_room": lambda: scenes.living_room(W, H)}[scene]()
    ctx = api.Context(api.Scene(sd), 0); seeds = api.IndependentSampler(0).block_seeds(W, H)
    import zlib
    for split in [int(x) for x in os.environ.get("SPLITS", "0").split(",")]:
        best = 1e9
        for r in range(3):
            t = time.perf_counter(); img, st = ctx.render(seeds, api.path_params(spp=spp, pipeline=pipeline, sample_split=split)); best = min(best, time.perf_counter() - t)
        print(f"{os.path.basename(libpath):28s} {scene} pl{pipeline} split{split} {best*1e3:8.1f} ms {W*H*spp/best/1e6:8.0f} Msamples/s iters {st['iterations']} crc {zlib.crc32(img.tobytes()):08x}", flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "build": build(sys.argv[2:])
    elif sys.argv[1] == "one": run_one(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    else:
        scene = sys.argv[2] if len(sys.argv) > 2 else "cbox"; pl = sys.argv[3] if len(sys.argv) > 3 else "2"; spp = sys.argv[4] if len(sys.argv) > 4 else "128"
        for lib in sorted(glob.glob(os.path.join(VDIR, "*.so"))):
            subprocess.run([sys.executable, __file__, "one", lib, scene, pl, spp], timeout=600)
