"""Round-4 design check: how fast does a walk started at a WRONG stream offset merge with the block's true chain?
(oracle only; n_p(o) = draws a camera sample of pixel p takes when it starts at offset o of the block stream)"""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from oracle import orc
from rustlight_amd import scenes as S

W, H, spp = 1920, 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 128
which = sys.argv[1] if len(sys.argv) > 1 else 'cbox'
sd = S.cbox(W, H) if which == 'cbox' else getattr(S, which)(W, H)
sc = orc.Scene(sd)
seeds = orc.block_seeds(0, W, H)
nby = (H + 15) // 16
bxi, byi = int(sys.argv[3]) if len(sys.argv) > 3 else 60, int(sys.argv[4]) if len(sys.argv) > 4 else 34
b = bxi * nby + byi
L = orc.lib()
pp = orc.path_params(spp=spp)
NMAX = 256 * spp * 40
r = orc.Rng(int(seeds[b]))
states = np.zeros((NMAX, 4), np.uint64)
st = r.state
for o in range(NMAX):
    states[o] = st[:]
    L.orc_rng_next_u64(st)
rgb = (C.c_float * 3)(); nv = C.c_uint64(); ns = C.c_uint64()
cache = {}
def n_of(p, o):
    k = (p, o)
    if k in cache: return cache[k]
    s = (C.c_uint64 * 4)(*[int(x) for x in states[o]])
    ix, iy = bxi * 16 + p % 16, byi * 16 + p // 16
    d = int(L.orc_compute_pixel(sc.h, C.byref(pp), ix, iy, s, rgb, C.byref(nv), C.byref(ns)))
    cache[k] = d
    return d
# truth
t = [0]
starts = {}
o = 0
for p in range(256):
    for s in range(spp):
        starts[(p, o)] = s
        o += n_of(p, o)
    t.append(o)
lens = np.diff(t)
print('block', bxi, byi, 'total draws', o, 'mean/sample', o / (256 * spp), 'per-pixel len mean', lens.mean(), 'std', lens.std())
alln = np.array([v for v in cache.values()])
print('n histogram', np.unique(alln, return_counts=True))
rng = np.random.default_rng(1)
for delta in (30, 300, 3000):
    merge_truth_idx, merge_track_cnt = [], []
    for p in range(1, 256):
        g = max(0, t[p] - delta - int(rng.integers(0, 17)))
        # spec walk from g with pixel p's function until it lands on a truth boundary of pixel p
        o = g; cnt = 0
        while (p, o) not in starts and o < t[p + 1] and cnt < 4 * spp:
            o += n_of(p, o); cnt += 1
        merge_track_cnt.append(cnt)
        merge_truth_idx.append(starts.get((p, o), spp))
    a = np.array(merge_truth_idx); c = np.array(merge_track_cnt)
    print(f'delta {delta}: truth sample index at merge: mean {a.mean():.1f} median {np.median(a)} p90 {np.percentile(a,90)} max {a.max()} never {np.sum(a>=spp)}; track samples before merge mean {c.mean():.1f}')
