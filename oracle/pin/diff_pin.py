"""Step 3 of oracle/pin/README.md:  python oracle/pin/diff_pin.py ref_draws.txt ref.pfm"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); OUT = os.path.join(HERE, "out")

def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split()); scale = float(f.readline())
        a = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return a[::-1].astype(np.float32)            # PFM rows are bottom-up (Bitmap::save_pfm, src/structure.rs:471-500)

ok = True
ref = [l.split() for l in open(sys.argv[1]) if l.strip()]
exp = [l.split() for l in open(os.path.join(OUT, "expected_draws.txt")) if l.strip()]
if ref == exp:
    print("draws: bit-exact (seed_variant 0 = rand_core PCG32 fill)")
else:
    alt = [l.split() for l in open(os.path.join(OUT, "expected_draws_variant1.txt")) if l.strip()]
    if [r for r in ref if r[0] == "u64"][:64] == alt:
        print("draws: match seed_variant 1 (SplitMix64) — flip the default of seed_variant (README)")
    else:
        bad = [i for i, (a, b) in enumerate(zip(ref, exp)) if a != b]
        print(f"draws: MISMATCH at lines {bad[:8]} (first: got {ref[bad[0]]}, expected {exp[bad[0]]})")
    ok = False
img = read_pfm(sys.argv[2])
want = np.load(os.path.join(OUT, "expected_cbox_64x64x4_reference_order.npy"))
if img.shape != want.shape:
    print("image: shape", img.shape, "expected", want.shape); sys.exit(1)
e = np.sum((img.astype(np.float64) - want.astype(np.float64)) ** 2, axis=-1)
diff = np.argwhere(e > 0)
print(f"image: {len(diff)} of {e.size} pixels differ, mean per-pixel squared L2 = {e.mean():.3e}, max = {e.max():.3e}" + (" — bit-exact" if len(diff) == 0 else ""))
for (y, x) in diff[:16]:
    print(f"   pixel ({x}, {y}): rustlight {img[y, x]}  oracle {want[y, x]}")
ok = ok and e.max() < 1e-3
sys.exit(0 if ok else 1)
