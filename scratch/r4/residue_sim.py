"""Prototype (oracle only): several full-window tracks per near-constant pixel at shifted starts (residue coverage) vs one track.
Reports serial samples per pixel for one block, lanes per block GROUP, R tracks per pixel (R = 1: today's S = 1 layout)."""
import sys, numpy as np, ctypes as C, bisect
sys.path.insert(0, '.')
from oracle import orc
from rustlight_amd import scenes as S
W, H, spp = 1920, 1080, 128
bxi, byi = int(sys.argv[1]), int(sys.argv[2])
sc = orc.Scene(S.cbox(W, H)); seeds = orc.block_seeds(0, W, H); nby = (H + 15) // 16
L = orc.lib(); pp = orc.path_params(spp=spp)
NMAX = 256 * spp * 40
r = orc.Rng(int(seeds[bxi * nby + byi])); states = np.zeros((NMAX, 4), np.uint64); st = r.state
for o in range(NMAX):
    states[o] = st[:]; L.orc_rng_next_u64(st)
rgb = (C.c_float * 3)(); nv = C.c_uint64(); ns = C.c_uint64(); cache = {}
def n_of(p, o):
    k = (p, o)
    if k in cache: return cache[k]
    s = (C.c_uint64 * 4)(*[int(x) for x in states[o]])
    d = int(L.orc_compute_pixel(sc.h, C.byref(pp), bxi * 16 + p % 16, byi * 16 + p // 16, s, rgb, C.byref(nv), C.byref(ns)))
    cache[k] = d; return d
truth = []; o = 0
for p in range(256):
    for s in range(spp): truth.append(o); o += n_of(p, o)
def run(NPIX, R, var_thr, shift_mode, ks=1.65, ke=1.65, lead=24, NP=32):
    est_L = [None] * NPIX; est_V = [None] * NPIX; mode_c = [None] * NPIX; mode_d = [None] * NPIX
    anchor = 0; out = []; slow = 0; spec = 0; spec_max = 0
    for q0 in range(0, 256, NPIX):
        if q0 == 0:
            for l in range(NPIX):
                oo = anchor + l * 997; a = []
                for k in range(NP): n = n_of(l, oo); a.append(n); oo += n
                a = np.array(a, float); est_L[l] = a.mean() * spp; est_V[l] = a.var() * spp * (1 + spp / NP)
                vals, cnts = np.unique(a, return_counts=True); mode_c[l] = int(vals[np.argmax(cnts)]); rest = [(c, v) for v, c in zip(vals, cnts) if v != mode_c[l]]
                mode_d[l] = int(max(rest)[1] - mode_c[l]) if rest else 1
        that = np.concatenate([[0], np.cumsum(est_L)]); Svar = np.concatenate([[0], np.cumsum([2 * v for v in est_V])])
        tracks = []; mx = 0
        for l in range(NPIX):
            p = q0 + l; nbar = est_L[l] / spp
            lo = 0 if l == 0 else max(0, int(that[l] - ks * np.sqrt(Svar[l]) - lead * nbar)); hi = int(that[l + 1] + ke * np.sqrt(Svar[l + 1]))
            lowvar = est_V[l] / spp < var_thr
            starts = [lo]
            if R > 1 and lowvar and l > 0:
                d = mode_d[l] if shift_mode == 'dev' else max(1, round(nbar / R))
                starts = [lo + k * abs(d) for k in range(R)]
            tl = []; walked = 0; allns = []
            for s0 in starts:
                offs = []; oo = anchor + s0
                while oo - anchor < hi and len(offs) < 4 * spp:
                    offs.append(oo); n = n_of(p, oo); allns.append(n); oo += n
                offs.append(oo); tl.append(offs); walked += len(offs) - 1
            tracks.append((tl, allns)); spec += walked; mx = max(mx, walked)
        spec_max += mx
        cur = anchor
        for l in range(NPIX):
            p = q0 + l; tl, allns = tracks[l]; i = 0; o = cur
            def find(o):
                for offs in tl:
                    j = bisect.bisect_left(offs, o)
                    if j < len(offs) and offs[j] == o: return offs, j
                return None
            while i < spp:
                f = find(o)
                if f is not None:
                    offs, j = f; k = min(spp - i, len(offs) - 1 - j); 
                    if k > 0: out.extend(offs[j:j + k]); i += k; o = offs[j + k]; continue
                out.append(o); o += n_of(p, o); i += 1; slow += 1
            est_L[l] = o - cur; a = np.array(allns, float)
            if len(a) > 8:
                est_V[l] = a.var() * spp; vals, cnts = np.unique(a, return_counts=True); mode_c[l] = int(vals[np.argmax(cnts)]); rest = [(c, v) for v, c in zip(vals, cnts) if v != mode_c[l]]
                mode_d[l] = int(max(rest)[1] - mode_c[l]) if rest else 1
            cur = o
        anchor = cur
    assert out == truth
    print(f'pixels/batch {NPIX} tracks {R} var_thr {var_thr} shift {shift_mode}: walked/truth {spec/(256*spp):.2f} max-lane/batch per pixel {spec_max/256:.1f} serial/pixel {slow/256:.2f}')
for NPIX in (16,):
    run(NPIX, 1, 0, 'dev')
    for R in (2, 4):
        for thr in (40, 1e9):
            for sm in ('dev', 'frac'): run(NPIX, R, thr, sm)
