"""Round-4 design prototype (oracle only): speculative windows + merge for one block's reference-order stream.
Reports, per GROUP size, the work the scheme would do: spec samples (sum and per-batch max lane) and slow (serial) samples."""
import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from oracle import orc
from rustlight_amd import scenes as S

W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bxi, byi = int(sys.argv[2]), int(sys.argv[3])
scene = sys.argv[4] if len(sys.argv) > 4 else 'cbox'
sd = S.cbox(W, H) if scene == 'cbox' else (S.cbox_medium(W, H) if scene == 'medium' else S.living_room(W, H, n_spheres=16, tess=8))
sc = orc.Scene(sd)
seeds = orc.block_seeds(0, W, H)
nby = (H + 15) // 16
b = bxi * nby + byi
L = orc.lib()
pp = orc.path_params(spp=spp)
NMAX = 256 * spp * 80
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
truth = []
o = 0
tstart = [0]
for p in range(256):
    for s in range(spp):
        truth.append(o); o += n_of(p, o)
    tstart.append(o)
print(f'block ({bxi},{byi}) {scene}: {o} draws, {o/(256*spp):.2f} per sample; per-pixel length mean {np.diff(tstart).mean():.0f} std {np.diff(tstart).std():.0f}')

def run(GROUP, ks=1.65, ke=1.65, lead=24, NP=32, probe_every=False, lead_k=0.0, lead_max=256):
    spec_sum = 0; spec_max = 0; slow = 0; probe = 0; slow_pre = 0; slow_post = 0; late = []
    est_L = [None] * GROUP; est_V = [None] * GROUP
    anchor = 0
    out = []
    miss_start = 0; perb = []
    for q0 in range(0, 256, GROUP):
        # probe (batch 0)
        if q0 == 0 or probe_every:
            for l in range(GROUP):
                oo = anchor + l * 997; ns_ = []
                for k in range(NP):
                    n = n_of(q0 + l, oo); ns_.append(n); oo += n
                a = np.array(ns_, float)
                if q0 == 0 or abs(a.mean() * spp - est_L[l]) > 3 * np.sqrt(spp * spp * a.var() / NP + est_V[l]):
                    est_L[l] = a.mean() * spp; est_V[l] = a.var() * spp * (1 + spp / NP)
            probe += NP
        that = np.concatenate([[0], np.cumsum(est_L)])
        Svar = np.concatenate([[0], np.cumsum([2 * v for v in est_V])])
        tracks = []
        mx = 0
        for l in range(GROUP):
            p = q0 + l
            nbar = est_L[l] / spp
            lead_l = lead
            if lead_k > 0:
                var_l = est_V[l] / spp if q0 > 0 else est_V[l] / spp / (1 + spp / NP)
                lead_l = min(lead_max, max(lead, lead_k / max(var_l, 1e-3)))
            lo = 0 if l == 0 else max(0, int(that[l] - ks * np.sqrt(Svar[l]) - lead_l * nbar))
            hi = int(that[l + 1] + ke * np.sqrt(Svar[l + 1]))
            offs = []; oo = anchor + lo; ns_ = []
            while oo - anchor < hi and len(offs) < 4 * spp + 64:
                offs.append(oo); n = n_of(p, oo); ns_.append(n); oo += n
            offs.append(oo)    # frontier
            tracks.append((offs, ns_))
            spec_sum += len(ns_); mx = max(mx, len(ns_))
        spec_max += mx
        # resolve
        cur = anchor
        for l in range(GROUP):
            p = q0 + l
            offs, ns_ = tracks[l]
            M = len(offs) - 1
            i = 0; o = cur
            import bisect
            j = bisect.bisect_left(offs, o)
            if o < offs[0]: miss_start += 1
            late.append((cur - anchor) - that[l])
            while i < spp and not (j <= M and offs[j] == o):
                out.append(o); o += n_of(p, o); i += 1; slow += 1; slow_pre += 1
                while j <= M and offs[j] < o: j += 1
            if i < spp:
                k = min(spp - i, M - j)
                out.extend(offs[j:j + k]); i += k; o = offs[j + k]
                while i < spp:
                    out.append(o); o += n_of(p, o); i += 1; slow += 1; slow_post += 1
            est_L[l] = o - cur
            a = np.array(ns_, float)
            est_V[l] = a.var() * spp if len(a) > 8 else est_V[l]
            cur = o
        anchor = cur
        perb.append((slow, spec_sum))
    assert out == truth, 'chain differs'
    nb = 256 // GROUP
    print(f'GROUP {GROUP:3d} ks {ks} ke {ke} lead {lead}: spec/truth {spec_sum/(256*spp):.2f}  wave iterations-as-samples per pixel (max lane/batch) {spec_max*GROUP/256/ (GROUP) :.1f}  slow/pixel {slow/256:.2f}  start-misses {miss_start}  probe/pixel {probe*GROUP/256/GROUP:.1f}'
          f'   => cost/pixel (3.3/64 per lane-sample) {spec_max/nb*nb/256*3.3*GROUP/64*  (64/GROUP) /1:.1f}')
    print('      slow pre-merge/pixel %.2f  post-track/pixel %.2f  rms(t - that) %.0f draws' % (slow_pre/256, slow_post/256, np.sqrt(np.mean(np.square(late)))))
    print('      per-batch cumulative (slow, spec):', perb[:6])
    return spec_sum, spec_max, slow

for G in (16, 32, 64):
    for ks, ke, lead in ((1.65, 1.65, 24), (2.5, 1.0, 32), (3.0, 3.0, 48)):
        run(G, ks, ke, lead)
print('---- adaptive lead')
for G in (16, 32):
    for lk, lm in ((0, 256), (2400, 128), (2400, 256), (4800, 256), (4800, 400)):
        print('lead_k', lk, 'max', lm); run(G, 1.65, 1.65, 24, lead_k=lk, lead_max=lm)
