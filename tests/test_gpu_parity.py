"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json): RNG sequence bit-exact; per-pixel squared L2 < 1e-3 at matched seeds.
Because both sides use IEEE f32 without contraction and the same deterministic transcendentals, the
comparisons below are in fact bit-exact against the oracle's forward evaluation order, and within
f32 rounding of the reference's recursive order."""
import os

import numpy as np
import pytest

from oracle import orc
from rustlight_amd import api, scenes

pytestmark = pytest.mark.gpu

L2_TOL = 1e-3     # per-pixel squared L2 tolerance stated by BASELINE.json


def per_pixel_l2(a, b):
    return np.sum((a.astype(np.float64) - b.astype(np.float64)) ** 2, axis=-1)


@pytest.fixture(scope="module")
def ctx_cbox(built, cbox64):
    return api.Context(api.Scene(cbox64), 0)


def test_numerics_contract_on_device(built):
    """IEEE divide / sqrt, no FMA contraction, denormals kept, transcendentals == oracle bit-for-bit."""
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-6.3, 6.3, 60000), rng.normal(0, 1e-3, 2000), [1e-39, 3e-45, 0.0]]).astype(np.float32)
    b = np.concatenate([rng.uniform(0.01, 50, 60000), rng.normal(0, 1e3, 2000), [0.5, 0.5, 1.0]]).astype(np.float32)
    out = api.numerics_probe(a, b)
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(out[0], a / b)
        np.testing.assert_array_equal(out[1], np.sqrt(np.abs(a)))
        np.testing.assert_array_equal(out[2], (a * b).astype(np.float32) + a)     # two roundings, not an FMA
    from tests.test_oracle_math import batch
    np.testing.assert_array_equal(out[3], batch(0, a))
    np.testing.assert_array_equal(out[4], batch(1, a))
    np.testing.assert_array_equal(out[5], batch(2, a))
    np.testing.assert_array_equal(out[6], batch(3, np.abs(a)))
    np.testing.assert_array_equal(out[7], batch(4, np.abs(a), b))
    np.testing.assert_array_equal(out[8], batch(5, a * np.float32(0.15)))
    np.testing.assert_array_equal(out[9], batch(6, a, b))
    assert out[0][-3] != 0.0          # 1e-39 / 0.5 stays a denormal


def _random_rays(sd, n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(0.05, 1.95, n)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("maker", [lambda: scenes.cbox(64, 64), lambda: scenes.living_room(64, 64, n_spheres=27, tess=12)])
def test_trace_batch_matches_oracle(built, maker):
    sd = maker()
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    o, d = _random_rays(sd, 200000, 1)
    if sd.n_triangles > 100:
        o = o * 3.0
        o[:, 1] += 4.0
    cam = np.array([osc.camera_generate(x + 0.5, y + 0.5) for x in range(0, 64, 2) for y in range(0, 64, 2)])
    o = np.concatenate([o, cam[:, 0]]).astype(np.float32)
    d = np.concatenate([d, cam[:, 1]]).astype(np.float32)
    got, ref = ctx.trace(o, d), osc.trace(o, d)
    for g, r, name in zip(got, ref, ["t", "u", "v", "mesh", "tri"]):
        np.testing.assert_array_equal(g, r, err_msg=name)
    assert (got[3] >= 0).mean() > 0.5
    # the reference's own cross-check: BVHAccel == NaiveAcceleration (accel.rs:14-77)
    brute = osc.trace(o[:20000], d[:20000], brute=True)
    np.testing.assert_array_equal(got[0][:20000], brute[0])


def test_two_level_records_find_the_reference_hits(built):
    """traverse2 (trace.hip.h): two levels of the reference's recursion (src/accel.rs:256-287) per fetched record — the grandchildren's exact boxes ride with the node,
    AABB::intersect (src/structure.rs:849-869) never reads its.t, so one fetch decides both levels with the identical comparisons.  Must be the oracle's hit for every
    ray, bit for bit (t, u, v, mesh, triangle) — random rays, camera rays, rays ALONG the axes from origins ON box planes (1 / d = inf, 0 * inf = NaN planes: the
    `unsafe` path that unions planes instead of distances), non-finite geometry — in about half the node trips.  (The build traverses the one-level records by
    default: profiles/NEGATIVES.md round 5 — fewer trips, more instructions; this keeps the construction honest.)"""
    def spoiled(kind):
        rng = np.random.default_rng(kind)
        sd = scenes.cbox(24, 20)
        c = rng.uniform(-0.8, 0.8, (30, 1, 3)); c[:, :, 1] += 1.0
        v = (c + rng.uniform(-0.3, 0.3, (30, 3, 3))).astype(np.float32).reshape(-1, 3)
        if kind == 0: v[rng.integers(0, len(v), 6)] = np.nan
        if kind == 1: v[rng.integers(0, len(v), 6), rng.integers(0, 3, 6)] = np.inf
        if kind == 2: v *= np.float32(1e30)
        sd.meshes.append(scenes.MeshData("adv", v, np.arange(90, dtype=np.uint32).reshape(-1, 3), None, None, scenes.matte((0.6, 0.6, 0.6))))
        return sd
    for n_case, maker in enumerate([lambda: scenes.cbox(64, 64), lambda: scenes.living_room(64, 64, n_spheres=27, tess=12), lambda: scenes.living_room(64, 64, n_spheres=64, tess=20),
                                    lambda: spoiled(0), lambda: spoiled(1), lambda: spoiled(2), lambda: scenes.single_triangle()]):
        sd = maker()
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        o, d = _random_rays(sd, 150000, 11 + n_case)
        if sd.n_triangles > 1000:
            o = o * 3.0
            o[:, 1] += 4.0
        # rays along the axes and in the axis planes (zero direction components), from origins snapped onto the planes of the BVH's boxes
        boxes = osc.bvh()[0].reshape(-1, 2, 3)
        rng = np.random.default_rng(5 + n_case)
        n_ax = 30000
        oa = rng.uniform(-1.0, 1.0, (n_ax, 3)).astype(np.float32); oa[:, 1] += 1.0
        if len(boxes):
            fin = boxes[np.isfinite(boxes).all(axis=(1, 2))]
            if len(fin):
                pick = fin[rng.integers(0, len(fin), n_ax)]
                for k in range(3):
                    snap = rng.random(n_ax) < 0.5
                    oa[snap, k] = pick[snap, rng.integers(0, 2, n_ax)[snap], k]
        da = np.zeros((n_ax, 3), np.float32)
        ax = rng.integers(0, 3, n_ax)
        da[np.arange(n_ax), ax] = rng.choice([-1.0, 1.0], n_ax)
        planar = rng.random(n_ax) < 0.4            # a second non-zero component: still one zero, 1 / d = inf on one axis
        ang = rng.uniform(0, 2 * np.pi, n_ax)
        da[planar, (ax[planar] + 1) % 3] = np.sin(ang[planar]); da[planar, ax[planar]] = np.cos(ang[planar])
        o = np.concatenate([o, oa]).astype(np.float32); d = np.concatenate([d, da]).astype(np.float32)
        got, ref, one = ctx.trace_two_level(o, d), osc.trace(o, d), ctx.trace(o, d)
        for g, r, w, name in zip(got, ref, one, ["t", "u", "v", "mesh", "tri"]):
            np.testing.assert_array_equal(g, r, err_msg=f"case {n_case} {name}")
            np.testing.assert_array_equal(g, w, err_msg=f"case {n_case} {name} (one-level)")
        # any-hit form against Acceleration::visible's inner traversal: segment o -> o + d * len
        seg = rng.uniform(0.2, 3.0, len(o)).astype(np.float32)
        found, _ = ctx.trace_two_level(o, d, segment_lengths=seg)
        ref_t = ref[0]
        # (a closest hit strictly inside the segment means the any-hit query finds something; one beyond it means it cannot)
        assert not (found & (ref[3] < 0)).any() and found[(ref[3] >= 0) & (ref_t < seg * 0.999)].all() and not found[(ref[3] >= 0) & (ref_t > seg * 1.001)].any()
    sd = scenes.living_room(64, 64, n_spheres=64, tess=20)
    ctx = api.Context(api.Scene(sd), 0)
    o, d = _random_rays(sd, 100000, 3); o = o * 3.0; o[:, 1] += 4.0
    steps2 = ctx.trace_two_level(o, d)[5]
    if not ctx.debug_sizes()["lds_scene"]:
        steps4 = ctx.trace_fast(o, d)[3]
        assert 0 < steps2.mean() < 1.1 * steps4.mean() * 1.6      # (the BVH4 collapses by area, the two-level records by depth: same order of magnitude)


def test_bvh4_of_the_tolerance_build_finds_the_same_hits(built, monkeypatch):
    """`numerics = fast` on scenes that stream their BVH traverses the same tree collapsed into quantised BVH4 nodes (build_bvh4 / traverse4:
    conservative 8-bit child boxes, a superset of the leaves).  Ray by ray against the exact BVH2 traversal (which equals the oracle and the
    brute-force loop, test_trace_batch_matches_oracle): same triangle for all but a handful of grazing rays, distances within 1e-5 relative
    (the tolerance build's 1-ulp divide), never a hit lost or invented away from a silhouette — and about half the node trips."""
    for maker, forced in ((lambda: scenes.living_room(64, 64, n_spheres=27, tess=12), False), (lambda: scenes.cbox(64, 64), True), (lambda: scenes.living_room(64, 64, n_spheres=64, tess=20), False)):
        if forced: monkeypatch.setenv("RL_FORCE_STREAMING", "1")
        else: monkeypatch.delenv("RL_FORCE_STREAMING", raising=False)
        sd = maker()
        ctx = api.Context(api.Scene(sd), 0)
        assert not ctx.debug_sizes()["lds_scene"]
        o, d = _random_rays(sd, 300000, 7)
        if sd.n_triangles > 100:
            o = o * 3.0
            o[:, 1] += 4.0
        t2, _, _, m2, tr2 = ctx.trace(o, d)
        t4, m4, tr4, steps = ctx.trace_fast(o, d)
        same = (m2 == m4) & (tr2 == tr4)
        # a ray through a shared edge may take the neighbouring triangle (the tolerance build decides the barycentric verdict from 1-ulp square roots):
        # same surface point, other primitive — counted as agreement when the distances agree
        close = same | ((m2 >= 0) & (m4 >= 0) & (np.abs(t4 - t2) <= 1e-4 * np.maximum(1.0, np.abs(t2))))
        assert same.mean() > 0.999 and close.mean() > 0.9998, (same.mean(), close.mean())
        hit = (m2 >= 0) & same
        assert hit.mean() > 0.3
        assert np.abs(t4[hit] - t2[hit]).max() <= 2e-5 * np.abs(t2[hit]).max() + 1e-6
        assert ((m2 >= 0) != (m4 >= 0)).mean() < 2e-4                       # hit / miss flips only on silhouettes
        assert steps.mean() > 0.5
    monkeypatch.delenv("RL_FORCE_STREAMING", raising=False)


def _triangle_soup(seed=5, n=600):
    """Slivers, tiny and huge triangles at scattered positions: stresses the barycentric verdict of the triangle test (the device
    decides it from a 1-ulp sqrt estimate outside a 1e-5 band around u + v = 1 and recomputes u, v for the final hit only)."""
    rng = np.random.default_rng(seed)
    V, I = [], []
    for k in range(n):
        scale = 10.0 ** rng.uniform(-4, 2)
        c = rng.uniform(-3, 3, 3) * (50.0 if k % 7 == 0 else 1.0)
        a = rng.normal(size=3); b = rng.normal(size=3)
        a /= np.linalg.norm(a); b -= a * (a @ b); b /= np.linalg.norm(b)
        aspect = 10.0 ** rng.uniform(-4, 0) if k % 3 == 0 else 1.0       # every third one is a sliver
        V += [c, c + a * scale, c + (a * rng.uniform(0, 1) + b * aspect) * scale]
        I.append([3 * k, 3 * k + 1, 3 * k + 2])
    m = scenes.MeshData("soup", np.asarray(V, np.float32), np.asarray(I, np.uint32), None, None, scenes.matte((0.5, 0.5, 0.5)))
    return scenes.SceneData(32, 32, 40.0, 0, np.asarray(scenes.CBOX_TO_WORLD, np.float32), False, [m])


def test_triangle_edge_cases_match_oracle(built):
    sd = _triangle_soup()
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    V = sd.meshes[0].vertices.astype(np.float64).reshape(-1, 3, 3)
    rng = np.random.default_rng(11)
    o, d = [], []
    for tri in V:
        v0, e1, e2 = tri[0], tri[1] - tri[0], tri[2] - tri[0]
        nrm = np.cross(e1, e2); nrm /= max(np.linalg.norm(nrm), 1e-300)
        size = max(np.linalg.norm(e1), np.linalg.norm(e2))
        for _ in range(40):
            kind = rng.integers(0, 5)
            if kind == 0:   u = rng.uniform(0, 1); v = 1.0 - u                                  # on the hypotenuse: u + v = 1
            elif kind == 1: u = rng.uniform(0, 1); v = (1.0 - u) * (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-8, -3))
            elif kind == 2: u, v = rng.choice([(0, 0), (1, 0), (0, 1)])                          # a vertex
            elif kind == 3: u = rng.uniform(0, 1); v = rng.choice([0.0, 1e-7, -1e-7])           # on / next to the edge v = 0
            else:           u, v = rng.uniform(0, 1, 2) * 0.5                                    # inside
            target = v0 + u * e1 + v * e2
            side = rng.normal(size=3) + nrm * rng.choice([-2, 2])
            side /= np.linalg.norm(side)
            org = target + side * size * 10.0 ** rng.uniform(-1, 2)
            o.append(org); d.append((target - org) / np.linalg.norm(target - org))
    o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    got, ref = ctx.trace(o, d), osc.trace(o, d)
    for g, r, name in zip(got, ref, ["t", "u", "v", "mesh", "tri"]):
        np.testing.assert_array_equal(g, r, err_msg=name)
    assert 0.2 < (got[3] >= 0).mean() < 0.99
    hit = got[3] >= 0
    near_edge = hit & (np.abs(got[1] + got[2] - 1.0) < 1e-4)
    assert near_edge.sum() > 500                      # the band around u + v = 1 really is exercised
    p1 = o + d * np.float32(1e3)
    np.testing.assert_array_equal(ctx.visible(o, p1), osc.visible(o, p1))


def test_visible_batch_matches_oracle(built, cbox64, ctx_cbox, orc_cbox64):
    o, _ = _random_rays(cbox64, 100000, 2)
    p1, _ = _random_rays(cbox64, 100000, 3)
    p1[::3] = np.float32([0.0, 1.98, -0.03])                 # many segments towards the light
    p1[5::7] = np.float32([0.0, 5.0, 0.0])                   # and some leaving the scene box
    got, ref = ctx_cbox.visible(o, p1), orc_cbox64.visible(o, p1)
    np.testing.assert_array_equal(got, ref)
    assert 0.05 < got.mean() < 0.95


def _render_pair(sd, ctx=None, osc=None, seed=0, **kw):
    ctx = ctx or api.Context(api.Scene(sd), 0)
    osc = osc or orc.Scene(sd)
    okw = {k: v for k, v in kw.items() if k not in ("pool_slots", "pipeline", "sample_split")}
    img, st = ctx.render(api.IndependentSampler(seed, kw.get("seed_variant", 0)).block_seeds(sd.width, sd.height), api.path_params(**kw))
    ref_fwd, ost = osc.render(master_seed=seed, eval_order=1, **okw)
    ref_rec, _ = osc.render(master_seed=seed, eval_order=0, **okw)
    return img, st, ref_fwd, ref_rec, ost


def _assert_parity(img, st, ref_fwd, ref_rec, ost):
    assert np.isfinite(img).all()
    for k in ("camera_samples", "vertices", "extension_rays", "rng_draws"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert st["shadow_rays"] == ost["shadow_rays"]
    np.testing.assert_array_equal(img, ref_fwd)                            # bit-exact vs the forward-order oracle
    e = per_pixel_l2(img, ref_rec)                                         # vs the reference's recursion order
    assert e.max() < L2_TOL and e.mean() < 1e-9, (e.max(), e.mean())


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
@pytest.mark.parametrize("mode", [api.STREAM_PER_SAMPLE, api.STREAM_REFERENCE_ORDER])
def test_cbox_render_parity(built, cbox64, ctx_cbox, orc_cbox64, mode, pipeline):
    """Both pipelines (wavefront stage kernels / persistent fused kernel) x both stream modes, bit-exact."""
    out = _render_pair(cbox64, ctx_cbox, orc_cbox64, spp=8, stream_mode=mode, pipeline=pipeline)
    _assert_parity(*out)
    assert out[0].mean() > 0.05
    assert (out[1]["iterations"] == 1) == (pipeline == api.PIPELINE_FUSED)


@pytest.mark.parametrize("kw", [dict(strategy=api.STRATEGY_BSDF), dict(strategy=api.STRATEGY_EMITTER), dict(max_depth=2), dict(max_depth=3, min_depth=1),
                                dict(rr_depth=None), dict(rr_depth=4, max_depth=8), dict(max_depth=1), dict(seed_variant=1), dict(single_scattering=True)])
def test_cbox_integrator_options(built, cbox64, ctx_cbox, orc_cbox64, kw):
    if kw.get("rr_depth", 0) is None:
        kw = dict(kw, max_depth=12)
    _assert_parity(*_render_pair(cbox64, ctx_cbox, orc_cbox64, spp=4, **kw))


def test_ragged_image_and_pool_smaller_than_image(built):
    sd = scenes.cbox(70, 41)          # edge blocks of 6x9 pixels
    osc = orc.Scene(sd)
    ctx = api.Context(api.Scene(sd), 0)
    for mode in (api.STREAM_PER_SAMPLE, api.STREAM_REFERENCE_ORDER):
        _assert_parity(*_render_pair(sd, ctx, osc, spp=3, stream_mode=mode, pool_slots=512))
    a = _render_pair(sd, ctx, osc, spp=3, pool_slots=256)[0]
    b = _render_pair(sd, ctx, osc, spp=3, pool_slots=0)[0]
    np.testing.assert_array_equal(a, b)       # launch geometry never changes results


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
@pytest.mark.parametrize("split", [2, 3, 8, 64])
def test_sample_parallel_pixels_keep_the_sum_order(built, split, pipeline):
    """`sample_split` lanes per pixel: samples run concurrently, their radiances are parked and added in sample order, so the
    image (and every counter) is the one-lane-per-pixel image bit for bit; spp not a multiple of the split, split > spp,
    ragged blocks and a pool smaller than the item count included."""
    sd = scenes.cbox(70, 41)
    out = _render_pair(sd, spp=7, pipeline=pipeline, sample_split=split, pool_slots=0 if pipeline == api.PIPELINE_FUSED else 1024)
    _assert_parity(*out)


def test_shards_sum_to_full_image(built, cbox64, ctx_cbox):
    seeds = api.IndependentSampler(5).block_seeds(64, 64)
    full, _ = ctx_cbox.render(seeds, api.path_params(spp=4))
    acc = np.zeros_like(full)
    for r in range(3):
        part, st = ctx_cbox.render(seeds, api.path_params(spp=4, shard_index=r, shard_count=3))
        assert np.count_nonzero(part.sum(-1)) <= np.count_nonzero(full.sum(-1))
        acc += part
    np.testing.assert_array_equal(acc, full)   # sums with zeros are exact: N-GPU image == 1-GPU image bitwise


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
def test_medium_parity(built, pipeline):
    for sd in (scenes.cbox_medium(48, 48, 0.5), scenes.cbox_medium(32, 32, 0.8, 0.2, g=0.6)):
        out = _render_pair(sd, spp=4, pipeline=pipeline)
        _assert_parity(*out)
        out = _render_pair(sd, spp=2, single_scattering=True, pipeline=pipeline)
        _assert_parity(*out)


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
def test_mixed_materials_parity(built, pipeline):
    """Six BSDF types in one scene: the wavefront pipeline sorts the live slots by material, the fused kernel switches per vertex."""
    sd = scenes.living_room(64, 48, n_spheres=27, tess=10)
    out = _render_pair(sd, spp=4, max_depth=10, pipeline=pipeline)
    _assert_parity(*out)
    assert (out[1]["iterations"] == 1) == (pipeline == api.PIPELINE_FUSED)
    _assert_parity(*_render_pair(sd, spp=5, max_depth=6, pipeline=pipeline, sample_split=2))


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
def test_small_scenes_through_the_streaming_kernels(built, monkeypatch, pipeline):
    """The kernels that stream the BVH from L2 / HBM (wave-voted node / leaf trips, uniform trips through the scalar cache, traversal stack
    overflowing from LDS to global memory) on scenes the oracle finishes in seconds: RL_FORCE_STREAMING keeps small scenes out of LDS."""
    monkeypatch.setenv("RL_FORCE_STREAMING", "1")
    for sd, kw in ((scenes.cbox(64, 48), dict(spp=6)),
                   (scenes.living_room(56, 40, n_spheres=27, tess=12), dict(spp=4, max_depth=8)),        # BVH deeper than the 6 LDS stack levels
                   (scenes.cbox_medium(40, 32, 0.5), dict(spp=3))):
        ctx = api.Context(api.Scene(sd), 0)
        assert not ctx.debug_sizes()["lds_scene"]
        _assert_parity(*_render_pair(sd, ctx=ctx, pipeline=pipeline, **kw))
        _assert_parity(*_render_pair(sd, ctx=ctx, pipeline=pipeline, stream_mode=api.STREAM_REFERENCE_ORDER, **kw))
        if pipeline == api.PIPELINE_FUSED:
            # the chain pass of reference-order streams reads the BVH through 16-node treelet blocks (traverse_treelet): against the plain per-node fetches
            a = ctx.render(api.IndependentSampler(0).block_seeds(sd.width, sd.height), api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, **kw))[0]
            with ctx.options(chain_no_treelets=1):
                b = ctx.render(api.IndependentSampler(0).block_seeds(sd.width, sd.height), api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, **kw))[0]
            np.testing.assert_array_equal(a, b)
    sd = scenes.living_room(48, 32, n_spheres=20, tess=8)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    seeds = api.IndependentSampler(4).block_seeds(sd.width, sd.height)
    img, st = ctx.render_direct(seeds, spp=3); ref, ost = osc.render_direct(seeds=seeds, spp=3)
    np.testing.assert_array_equal(img, ref)
    assert all(st[k] == ost[k] for k in ("camera_samples", "extension_rays", "shadow_rays", "rng_draws"))
    img, st = ctx.render_ao(seeds, spp=3, max_distance=1.0); ref, ost = osc.render_ao(seeds=seeds, spp=3, max_distance=1.0)
    np.testing.assert_array_equal(img, ref)


def test_phong_beckmann_textures_parity(built):
    sd = scenes.cbox(48, 48)
    S = scenes
    sd.meshes[0].bsdf = S.Bsdf(type=S.DIFFUSE, diffuse={"type": S.TEX_CHECKERBOARD, "color0": (0.8, 0.8, 0.8), "color1": (0.1, 0.1, 0.1), "scale": (4.0, 4.0)})
    sd.meshes[2].bsdf = S.Bsdf(type=S.PHONG, diffuse=S.const_color((0.3, 0.3, 0.3)), specular=S.const_color((0.5, 0.5, 0.5)), exponent=30.0, weight_specular=0.6)
    sd.meshes[5].bsdf = S.Bsdf(type=S.METAL, distribution=S.MF_BECKMANN, alpha_u=0.2, alpha_v=0.2)
    sd.meshes[6].bsdf = S.Bsdf(type=S.SUBSTRATE, diffuse={"type": S.TEX_GRID, "color0": (0.9, 0.1, 0.1), "color1": (0.4, 0.4, 0.4), "line_width": 0.05, "scale": (3.0, 0.0)},
                               specular=S.const_color((0.04, 0.04, 0.04)), distribution=S.MF_NONE)
    for pipeline in (api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED):
        _assert_parity(*_render_pair(sd, spp=4, max_depth=8, pipeline=pipeline))


@pytest.mark.parametrize("combo", [dict(), dict(keep_area_light=False), dict(point=False, directional=False), dict(environment=False, keep_area_light=False)])
def test_point_directional_environment_emitters_parity(built, combo):
    """PointEmitter, DirectionalLight, EnvironmentLight::Constant next to / instead of the mesh light."""
    sd = scenes.cbox_other_lights(48, 48, **combo)
    for kw in (dict(spp=4), dict(spp=2, strategy=api.STRATEGY_EMITTER), dict(spp=2, strategy=api.STRATEGY_BSDF, max_depth=6), dict(spp=2, min_depth=1)):
        _assert_parity(*_render_pair(sd, **kw))


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
@pytest.mark.parametrize("keep_area_light", [False, True])
def test_textured_environment_parity(built, keep_area_light, pipeline):
    """SURVEY.md a24 remainder: EnvironmentLightColor::Texture (lat-long image, Distribution2D importance sampling, MIS
    through direct_pdf of the hit direction), alone and next to the mesh light, all three strategies."""
    sd = scenes.sky_scene(48, 40, keep_area_light=keep_area_light)
    for kw in (dict(spp=4), dict(spp=2, strategy=api.STRATEGY_EMITTER), dict(spp=2, strategy=api.STRATEGY_BSDF, max_depth=6), dict(spp=2, min_depth=1, stream_mode=api.STREAM_REFERENCE_ORDER)):
        out = _render_pair(sd, pipeline=pipeline, **kw)
        _assert_parity(*out)
        assert out[0].mean() > 0.02


@pytest.mark.parametrize("pipeline", [api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED])
def test_light_tree_sampling_parity(built, pipeline):
    """`-x ats` — NEE draws (emitter, triangle) by descending the light tree with
    importance_point(p, Some(n_s) | None), MIS evaluates the tree pdf of the triangle a BSDF ray hit (n = None): bit-exact
    against the oracle for path (all strategies, with a medium) and direct."""
    sd = scenes.many_lights(48, 40, 3, glowing_spheres=2)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    for kw in (dict(spp=4), dict(spp=2, strategy=api.STRATEGY_EMITTER), dict(spp=2, strategy=api.STRATEGY_BSDF, max_depth=5), dict(spp=2, stream_mode=api.STREAM_REFERENCE_ORDER, max_depth=4)):
        out = _render_pair(sd, ctx, osc, pipeline=pipeline, **kw)
        _assert_parity(*out)
        assert out[0].mean() > 0.05
    fog = scenes.many_lights(32, 32, 2, glowing_spheres=1)
    fog.medium = scenes.Medium((0.05, 0.05, 0.05), (0.4, 0.4, 0.4))
    _assert_parity(*_render_pair(fog, spp=3, pipeline=pipeline, max_depth=6))
    if pipeline == api.PIPELINE_WAVEFRONT:
        seeds = api.IndependentSampler(4).block_seeds(sd.width, sd.height)
        for kw in (dict(), dict(nb_bsdf_samples=2, nb_light_samples=2)):
            img, st = ctx.render_direct(seeds, spp=3, **kw)
            ref, ost = osc.render_direct(seeds=seeds, spp=3, **kw)
            np.testing.assert_array_equal(img, ref)
            assert all(st[k] == ost[k] for k in ("camera_samples", "extension_rays", "shadow_rays", "rng_draws"))


@pytest.mark.parametrize("mode", [api.STREAM_PER_SAMPLE, api.STREAM_REFERENCE_ORDER])
def test_ao_and_direct_integrators_parity(built, mode):
    """`ao` and `direct` through the same tiling driver, bit-exact vs the oracle."""
    for sd in (scenes.cbox(48, 40), scenes.cbox_other_lights(40, 40), scenes.sky_scene(40, 40, keep_area_light=True), scenes.living_room(48, 32, n_spheres=20, tess=8)):
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        seeds = api.IndependentSampler(4).block_seeds(sd.width, sd.height)
        for kw in (dict(max_distance=1.0), dict(max_distance=None), dict(max_distance=0.3, normal_correction=True)):
            img, st = ctx.render_ao(seeds, spp=3, stream_mode=mode, **kw)
            ref, ost = osc.render_ao(seeds=seeds, spp=3, stream_mode=mode, **kw)
            np.testing.assert_array_equal(img, ref)
            assert all(st[k] == ost[k] for k in ("camera_samples", "extension_rays", "rng_draws"))
        for kw in (dict(), dict(nb_bsdf_samples=2, nb_light_samples=3), dict(nb_bsdf_samples=0, nb_light_samples=1), dict(nb_bsdf_samples=1, nb_light_samples=0)):
            img, st = ctx.render_direct(seeds, spp=3, stream_mode=mode, **kw)
            ref, ost = osc.render_direct(seeds=seeds, spp=3, stream_mode=mode, **kw)
            np.testing.assert_array_equal(img, ref)
            assert all(st[k] == ost[k] for k in ("camera_samples", "extension_rays", "shadow_rays", "rng_draws"))
            assert img.mean() > 0
        if mode == api.STREAM_REFERENCE_ORDER:
            # the mode runs in two passes (k_mc_chain records where each camera sample starts in its block's stream, the per-pixel form evaluates them);
            # the single-pass walk (one lane per block) must give the same images
            two_ao, sa = ctx.render_ao(seeds, spp=3, stream_mode=mode, max_distance=0.3, normal_correction=True)
            two_di, sd2 = ctx.render_direct(seeds, spp=3, stream_mode=mode, nb_bsdf_samples=2, nb_light_samples=3)
            assert sa["ms_prepass"] > 0.0 and sd2["ms_prepass"] > 0.0
            with ctx.options(ref_single_pass=1):
                one_ao, sb = ctx.render_ao(seeds, spp=3, stream_mode=mode, max_distance=0.3, normal_correction=True)
                one_di, _ = ctx.render_direct(seeds, spp=3, stream_mode=mode, nb_bsdf_samples=2, nb_light_samples=3)
            assert sb["ms_prepass"] == 0.0
            np.testing.assert_array_equal(two_ao, one_ao)
            np.testing.assert_array_equal(two_di, one_di)


@pytest.mark.parametrize("fmt", ["obj", "serialized"])
def test_scene_files_render_like_the_fixture(built, tmp_path, fmt):
    """a mixed-material scene written as Mitsuba XML (+ OBJ / serialized meshes) and read back by
    rl_scene_load renders the image of the in-memory fixture bit for bit (all five BSDF families, area lights, a point light)."""
    from rustlight_amd import export
    sd = scenes.living_room(48, 32, n_spheres=12, tess=8)
    sd.flip = True
    sd.lights.append({"type": "point", "a": (0.5, 3.0, 1.0), "intensity": (4.0, 4.0, 3.0)})
    f = np.float32
    lum = lambda c: (f(c[0]) * f(0.212671) + f(c[1]) * f(0.715160)) + f(c[2]) * f(0.072169)
    for m in sd.meshes:                      # bsdf_mts derives Phong's lobe weight from the two reflectances (bsdfs/mod.rs:521-526)
        if m.bsdf.type == scenes.PHONG:
            d, s = lum(m.bsdf.diffuse["color0"]), lum(m.bsdf.specular["color0"])
            m.bsdf.weight_specular = float(s / (d + s))
    p = str(tmp_path / "room.xml")
    export.write_mitsuba(sd, p, fmt)
    seeds = api.IndependentSampler(2).block_seeds(48, 32)
    a, sta = api.Context(api.Scene.load(p), 0).render(seeds, api.path_params(spp=4))
    b, stb = api.Context(api.Scene(sd), 0).render(seeds, api.path_params(spp=4))
    np.testing.assert_array_equal(a, b)
    assert sta["vertices"] == stb["vertices"] and a.mean() > 0.01


def test_golden_feature_renders_on_gpu(built):
    """The committed golden renders (tests/golden/features.npz, made by the oracle in the authoring container) are reproduced
    bit for bit by the kernels — no oracle call in this test."""
    import os
    from tests.golden.make_golden import feature_cases
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features.npz"))
    for name, (sd, integ, kw) in feature_cases().items():
        ctx = api.Context(api.Scene(sd), 0)
        seeds = api.IndependentSampler(7).block_seeds(sd.width, sd.height)
        kw = dict(kw)
        spp = kw.pop("spp")
        if integ == "path":
            img = ctx.render(seeds, api.path_params(spp=spp, **kw))[0]
            np.testing.assert_array_equal(ctx.render(seeds, api.path_params(spp=spp, pipeline=api.PIPELINE_WAVEFRONT, **kw))[0], gold[name], err_msg=name + " (wavefront)")
        elif integ == "direct":
            img = ctx.render_direct(seeds, spp=spp, **kw)[0]
        else:
            img = ctx.render_ao(seeds, spp=spp, **kw)[0]
        np.testing.assert_array_equal(img, gold[name], err_msg=name)


def test_cli_renders_what_the_api_renders(built, tmp_path):
    """`rustlight-amd` (the C++ mirror of examples/cli.rs) end to end: scene file -> loader -> emitters (-x ats) -> integrator ->
    image file, equal to the Python mirror's render with the same master seed; `-m` adds the medium, `direct` / `ao` run too."""
    import os
    import subprocess
    from rustlight_amd import export
    cli = os.path.join(os.path.dirname(api.LIB_PATH), "rustlight-amd")
    sd = scenes.many_lights(40, 32, 2)
    sd.flip, sd.fov_axis = True, 0
    xml = str(tmp_path / "lights.xml")
    export.write_mitsuba(sd, xml, "ply")
    def run(*args):
        out = str(tmp_path / "out.pfm")
        r = subprocess.run([cli, xml, "-n", "3", "-r", "independent:11", "-o", out, "--stream-mode", "per-sample", *args], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return api.load_pfm(out)
    seeds = lambda: api.IndependentSampler(11).block_seeds(40, 32)
    ctx = api.Context(api.Scene(sd), 0)
    np.testing.assert_array_equal(run("-x", "ats", "path", "-m", "5"), ctx.render(seeds(), api.path_params(spp=3, max_depth=5))[0])
    np.testing.assert_array_equal(run("-x", "ats", "direct", "-b", "2", "-l", "1"), ctx.render_direct(seeds(), spp=3, nb_bsdf_samples=2, nb_light_samples=1)[0])
    sd.use_ats = False
    flat = api.Context(api.Scene(sd), 0)
    np.testing.assert_array_equal(run("path", "-s", "emitter"), flat.render(seeds(), api.path_params(spp=3, strategy=api.STRATEGY_EMITTER))[0])
    np.testing.assert_array_equal(run("ao", "-d", "0.5"), flat.render_ao(seeds(), spp=3, max_distance=0.5)[0])
    # the CLI's default stream mode is rustlight's own (one SmallRng per 16x16 block): a later --stream-mode wins, so this is the default path
    np.testing.assert_array_equal(run("--stream-mode", "reference", "path"), flat.render(seeds(), api.path_params(spp=3, stream_mode=api.STREAM_REFERENCE_ORDER))[0])
    # --option name=value: an execution option of the device context (rl_context_set_option through the C++ integrator mirror) — another form of the same render; an unknown
    # name is an error, not a silently ignored flag
    np.testing.assert_array_equal(run("--stream-mode", "reference", "--option", "ref_single_pass=1", "--option", "no_events", "path"),
                                  flat.render(seeds(), api.path_params(spp=3, stream_mode=api.STREAM_REFERENCE_ORDER))[0])
    bad = subprocess.run([cli, xml, "-n", "1", "-o", str(tmp_path / "bad.pfm"), "--option", "no_such_option=1", "path"], capture_output=True, text=True)
    assert bad.returncode != 0 and "no_such_option" in (bad.stderr + bad.stdout)
    # --gpus N: N device contexts (round-robin over the visible devices — three on the one GPU here), one host thread each,
    # per-shard framebuffers added on the host: the single-context image again
    np.testing.assert_array_equal(run("--gpus", "3", "path", "-s", "emitter"), flat.render(seeds(), api.path_params(spp=3, strategy=api.STRATEGY_EMITTER))[0])
    sd.medium = scenes.Medium((0.1, 0.1, 0.1), (0.3, 0.3, 0.3), scenes.PHASE_HG, 0.5)
    fog = api.Context(api.Scene(sd), 0)
    np.testing.assert_array_equal(run("-m", "0.3:0.1:0.5", "path", "-m", "6"), fog.render(seeds(), api.path_params(spp=3, max_depth=6))[0])


def test_progressive_wrappers(built, cbox64, tmp_path):
    """IntegratorAverage / IntegratorEqualTime (avg.rs, equal_time.rs): every pass draws fresh block seeds from the same,
    advancing master sampler; pass k of the wrapper equals a plain render with the k-th batch of seeds."""
    scene = api.Scene(cbox64)
    inner = api.IntegratorPathTracing(stream_mode=api.STREAM_PER_SAMPLE)
    avg = api.IntegratorAverage(inner, max_iterations=3)
    out = str(tmp_path / "avg.pfm")
    img = avg.compute(api.IndependentSampler(9), scene, nb_samples=2, output_img_path=out)
    s = api.IndependentSampler(9)
    passes = [api.Context(scene, 0).render(s.block_seeds(64, 64), api.path_params(spp=2))[0] for _ in range(3)]
    want = passes[0]
    for k, p in enumerate(passes[1:], start=2):
        # avg.rs:59-61 weighs the old image by k and divides by k + 1 (the counter is already one ahead); kept as is
        want = ((want * np.float32(k)) + p) * (np.float32(1.0) / np.float32(k + 1))
    np.testing.assert_array_equal(img, want)
    import os
    assert all(os.path.exists(str(tmp_path / f"avg_{k}.pfm")) for k in (1, 2, 3)) and os.path.exists(str(tmp_path / "avg_time.csv"))
    eq = api.IntegratorEqualTime(api.IntegratorPathTracing(stream_mode=api.STREAM_PER_SAMPLE), target_time_ms=0.0)
    one = eq.compute(api.IndependentSampler(9), scene, nb_samples=2)
    assert eq.iterations == 1
    np.testing.assert_array_equal(one, passes[0])


def test_frames_in_flight_render_the_same_frames(built, cbox64, tmp_path):
    """Frames in flight: several contexts of one scene rendering independent frames from their own host threads (api.render_in_flight,
    IntegratorPathTracing.compute_frames, IntegratorAverage on top) give the images of one frame after the other — both stream modes, the
    default mode through its speculative chain pass — and the progressive average folds them in pass order."""
    scene = api.Scene(cbox64)
    for mode, spp in ((api.STREAM_REFERENCE_ORDER, 48), (api.STREAM_PER_SAMPLE, 4)):
        jobs = [(api.IndependentSampler(100 + f).block_seeds(64, 64), api.path_params(spp=spp, stream_mode=mode)) for f in range(7)]
        one = api.Context(scene, 0)
        want = [one.render(*j) for j in jobs]
        flying = [api.Context(scene, 0) for _ in range(3)]
        for c in flying: c.set_option("spec_force", 1)           # (64 x 64 x 48 spp would pick the serial chain pass)
        got = api.render_in_flight(flying, jobs)
        for (wi, ws), (gi, gs) in zip(want, got):
            np.testing.assert_array_equal(gi, wi)
            assert gs["rng_draws"] == ws["rng_draws"] and gs["vertices"] == ws["vertices"]
        # the same behind one C-ABI call (the library's own threads), and its argument checks
        cs = [api.Context(scene, 0) for _ in range(2)]
        imgs, sts = api.render_frames(cs, [j[0] for j in jobs], jobs[0][1])
        for (wi, ws), gi, gs in zip(want, imgs, sts):
            np.testing.assert_array_equal(gi, wi)
            assert gs["rng_draws"] == ws["rng_draws"]
        with pytest.raises(api.RustlightError):
            api.render_frames([cs[0], cs[0]], [j[0] for j in jobs[:2]], jobs[0][1])      # one frame at a time per context
    seq = api.IntegratorAverage(api.IntegratorPathTracing(), max_iterations=5, dump_all=False)
    a = seq.compute(api.IndependentSampler(9), scene, nb_samples=2, output_img_path=str(tmp_path / "a.pfm"))
    s2 = api.IndependentSampler(9)
    par = api.IntegratorAverage(api.IntegratorPathTracing(frames_in_flight=2), max_iterations=5, dump_all=False)
    b = par.compute(s2, scene, nb_samples=2, output_img_path=str(tmp_path / "b.pfm"))
    np.testing.assert_array_equal(b, a)
    assert par.iterations == 5
    eq = api.IntegratorEqualTime(api.IntegratorPathTracing(frames_in_flight=3), target_time_ms=0.0)      # stops after the first pass of its first batch
    np.testing.assert_array_equal(eq.compute(api.IndependentSampler(9), scene, nb_samples=2),
                                  api.IntegratorEqualTime(api.IntegratorPathTracing(), target_time_ms=0.0).compute(api.IndependentSampler(9), scene, nb_samples=2))
    # the C++ mirror through the CLI: -a 0 stops after the first pass of the first batch; its image is the plain render's
    import subprocess
    from rustlight_amd import export
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "rustlight-amd")
    scn = str(tmp_path / "cbox.xml")
    export.write_mitsuba(cbox64, scn, "ply")
    outs = []
    for extra in ([], ["--frames-in-flight", "3", "-a", "0"]):
        out = str(tmp_path / f"cli{len(outs)}.pfm")
        r = subprocess.run([exe, scn, "-n", "2", "-r", "independent:5", "-o", out, *extra, "path"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(api.load_pfm(out))
    np.testing.assert_array_equal(outs[1], outs[0])


def test_degenerate_inputs(built):
    """Empty scene (no mesh at all, with and without an environment), a 1 x 1 image, a 17 x 1 strip, spp = 1, a mesh-less scene lit
    by a point light only: no crash, same bits as the oracle."""
    to_world = np.asarray(scenes.CBOX_TO_WORLD, np.float32)
    empty = scenes.SceneData(20, 12, 40.0, 0, to_world, False, [])
    out = _render_pair(empty, spp=2, strategy=api.STRATEGY_BSDF)
    _assert_parity(*out)
    assert out[0].max() == 0.0
    sky = scenes.SceneData(20, 12, 40.0, 0, to_world, False, [], environment=(0.25, 0.5, 1.0))
    out = _render_pair(sky, spp=2)
    _assert_parity(*out)
    np.testing.assert_array_equal(out[0][3, 4], np.asarray([0.25, 0.5, 1.0], np.float32))
    for (w, h) in ((1, 1), (17, 1), (1, 33)):
        for pl in (api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED):
            _assert_parity(*_render_pair(scenes.cbox(w, h), spp=1, pipeline=pl))
    floor_only = scenes.SceneData(24, 16, 30.0, 0, np.asarray([1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 3, 0, 1], np.float32), False,
                                  [scenes._quad_mesh("Floor", [-5, 0, -5, -5, 0, 5, 5, 0, 5, 5, 0, -5], [0, 1, 0], scenes.matte((0.5, 0.5, 0.5)))],
                                  lights=[{"type": "point", "a": (0.0, 2.0, 0.0), "intensity": (3.0, 3.0, 3.0)}])
    out = _render_pair(floor_only, spp=2)
    _assert_parity(*out)
    assert out[0].mean() > 0.01


def test_randomized_parity(built):
    """A short run of the differential fuzzer (tests/parity_fuzz.py): random scenes, emitters, materials, media, integrator options,
    stream modes, pipelines, lanes per pixel and pool sizes — image bits and counters must equal the oracle's in every case."""
    from tests.parity_fuzz import run
    n, bad = run(budget=25.0, seed=3)
    assert bad == 0 and n >= 100, (n, bad)
    n, bad = run(budget=12.0, seed=4, fast=True)            # the same generator, eligible cases also through the tolerance build (statistical bar)
    assert bad == 0 and n >= 20, (n, bad)


def test_non_finite_and_degenerate_geometry_parity(built):
    """NaN / infinite vertices, zero-area and coincident triangles, 1e30 / 1e-30 coordinates in a non-emissive mesh inside the Cornell box:
    no fault, no hang, same bits and counters as the oracle in both pipelines (such triangles are simply never hit).  The same mesh as
    an emitter is refused when the scene is built (tests/test_abi.py)."""
    def spoiled(kind, seed):
        rng = np.random.default_rng(seed)
        sd = scenes.cbox(24, 20)
        nt = 30
        c = rng.uniform(-0.8, 0.8, (nt, 1, 3)); c[:, :, 1] += 1.0
        v = (c + rng.uniform(-0.3, 0.3, (nt, 3, 3))).astype(np.float32).reshape(-1, 3)
        if kind == 0: v[rng.integers(0, len(v), 6)] = np.nan
        if kind == 1: v[rng.integers(0, len(v), 6), rng.integers(0, 3, 6)] = np.inf
        if kind == 2: v[:] = np.repeat(v[::3], 3, 0)
        if kind == 3: v *= np.float32(1e30)
        if kind == 4: v *= np.float32(1e-30)
        if kind == 5: v[:] = np.float32(0.5)
        sd.meshes.append(scenes.MeshData("adv", v, np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3), None, None, scenes.matte((0.6, 0.6, 0.6))))
        return sd
    for kind in range(6):
        sd = spoiled(kind, kind)
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        ref, ost = osc.render(master_seed=kind, spp=2, eval_order=1, max_depth=6)
        for pipeline in (api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED):
            img, st = ctx.render(api.IndependentSampler(kind).block_seeds(sd.width, sd.height), api.path_params(spp=2, max_depth=6, pipeline=pipeline))
            np.testing.assert_array_equal(img, ref, err_msg=f"kind {kind} pipeline {pipeline}")
            assert all(st[k] == ost[k] for k in ("vertices", "rng_draws", "shadow_rays", "extension_rays")), (kind, pipeline)
        o, d = _random_rays(sd, 20000, kind)
        for g, r in zip(ctx.trace(o, d), osc.trace(o, d)):
            np.testing.assert_array_equal(g, r)


def test_hostile_shading_inputs_parity(built):
    """NaN / infinite / huge uv, NaN normals, NaN / infinite / negative texels, NaN or infinite roughness, an all-black or negative
    environment map: no fault, and the image equals the oracle's bit for bit *including its NaNs* — a NaN throughput poisons the sample
    exactly as the reference's `weight * evaluate(next)` does, zero terms are added as beta * 0 rather than skipped."""
    S = scenes

    def make(kind, seed):
        rng = np.random.default_rng(seed)
        sd = S.cbox(24, 20) if kind < 6 else S.sky_scene(24, 20, keep_area_light=bool(seed % 2))
        bm = rng.uniform(0, 1, (5, 7, 3)).astype(np.float32)
        if kind == 3:
            bm[1, 2] = np.nan; bm[3, 3] = np.inf; bm[0, 0] = -1.0
        sd.bitmaps = [(7, 5, bm)]
        texs = [{"type": S.TEX_BITMAP, "bitmap_id": 0, "color0": (1, 1, 1), "scale": (2.0, 3.0)},
                {"type": S.TEX_CHECKERBOARD, "color0": (0.8, 0.8, 0.8), "color1": (0.1, 0.2, 0.3), "scale": (4.0, 4.0)},
                {"type": S.TEX_GRID, "color0": (0.9, 0.1, 0.1), "color1": (0.4, 0.4, 0.4), "line_width": 0.05, "scale": (3.0, 2.0)}]
        for i, m in enumerate(sd.meshes):
            if m.emission:
                continue
            n = len(m.vertices)
            m.uv = rng.uniform(-3, 3, (n, 2)).astype(np.float32)
            m.bsdf = (S.Bsdf(type=S.DIFFUSE, diffuse=texs[i % 3]) if i % 2 else
                      S.Bsdf(type=S.SUBSTRATE, diffuse=texs[i % 3], specular=S.const_color((0.04, 0.04, 0.04)), distribution=S.MF_GGX, alpha_u=0.2, alpha_v=0.3))
            if kind == 0: m.uv[rng.integers(0, n)] = np.nan
            if kind == 1: m.uv[rng.integers(0, n)] = (np.inf, -np.inf)
            if kind == 2: m.uv *= np.float32(1e30)
            if kind == 4 and m.normals is not None:
                m.normals = m.normals.copy(); m.normals[rng.integers(0, n)] = np.nan
            if kind == 5: m.bsdf = S.Bsdf(type=S.METAL, distribution=S.MF_GGX, alpha_u=float("nan") if i % 2 else 0.0, alpha_v=0.0 if i % 2 else float("inf"))
        if kind >= 6:
            em = S.sky_map().copy()
            if kind == 6: em[:, :] = 0.0
            if kind == 7: em[1, 1] = -5.0
            sd.environment_map = em
        return sd
    saw_nan = False
    for kind in range(8):
        for seed in range(2):
            sd = make(kind, seed)
            ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
            for kw in (dict(spp=2, max_depth=5), dict(spp=2, max_depth=3, strategy=api.STRATEGY_BSDF)):
                ref, ost = osc.render(master_seed=seed, eval_order=1, **kw)
                saw_nan |= bool(np.isnan(ref).any())
                for pipeline in (api.PIPELINE_WAVEFRONT, api.PIPELINE_FUSED):
                    img, st = ctx.render(api.IndependentSampler(seed).block_seeds(sd.width, sd.height), api.path_params(pipeline=pipeline, **kw))
                    np.testing.assert_array_equal(img, ref, err_msg=f"kind {kind} seed {seed} pipeline {pipeline} {kw}")      # NaN == NaN here
                    assert all(st[k] == ost[k] for k in ("vertices", "rng_draws", "shadow_rays")), (kind, seed, pipeline)
    assert saw_nan


def test_extreme_execution_parameters(built):
    """Tuning parameters are caller input: absurd values give the same image (pool sizes are clamped to the work there is, lanes per
    pixel to the samples there are), more shards than tiles still sum to the full image, an unknown pipeline is an error."""
    sd = scenes.cbox(40, 24)
    ctx = api.Context(api.Scene(sd), 0)
    seeds = api.IndependentSampler(1).block_seeds(sd.width, sd.height)
    ref = ctx.render(seeds, api.path_params(spp=3))[0]
    for kw in (dict(pool_slots=1), dict(pool_slots=3), dict(pool_slots=257), dict(pool_slots=(1 << 31) - 1), dict(pool_slots=0xFFFFFFFF), dict(sample_split=1000000),
               dict(sample_split=3, pipeline=1, pool_slots=7), dict(pipeline=2, pool_slots=5)):
        np.testing.assert_array_equal(ctx.render(seeds, api.path_params(spp=3, **kw))[0], ref, err_msg=str(kw))
    with pytest.raises(api.RustlightError, match="pipeline"):
        ctx.render(seeds, api.path_params(spp=3, pipeline=3))
    parts = [ctx.render(seeds, api.path_params(spp=3, shard_index=r, shard_count=50))[0] for r in range(50)]
    np.testing.assert_array_equal(sum(parts[1:], parts[0]), ref)


def test_furnace_invariant_on_gpu(built):
    sd = scenes.furnace(albedo=0.5, le=1.0, width=16, height=16)
    ctx = api.Context(api.Scene(sd), 0)
    img, _ = ctx.render(api.IndependentSampler(3).block_seeds(16, 16), api.path_params(spp=256))
    assert abs(img.mean() - 2.0) < 0.05, img.mean()          # Le / (1 - albedo)


def test_full_size_properties(built):
    """BASELINE cfg 2 at reduced spp: determinism, shard additivity and counters at 1920x1080."""
    sd = scenes.cbox(1920, 1080)
    ctx = api.Context(api.Scene(sd), 0)
    seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    a, st = ctx.render(seeds, api.path_params(spp=4))
    b, _ = ctx.render(seeds, api.path_params(spp=4, pool_slots=1 << 19))
    c, stc = ctx.render(seeds, api.path_params(spp=4, pipeline=api.PIPELINE_WAVEFRONT))
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)                      # fused (auto) == wavefront, bitwise
    assert all(st[k] == stc[k] for k in ("vertices", "extension_rays", "shadow_rays", "rng_draws"))
    assert st["camera_samples"] == 1920 * 1080 * 4 and np.isfinite(a).all() and (a >= 0).all()
    parts = [ctx.render(seeds, api.path_params(spp=4, shard_index=r, shard_count=8, pipeline=1 + r % 2))[0] for r in range(8)]
    np.testing.assert_array_equal(sum(parts[1:], parts[0]), a)
    # a 64x64 crop of blocks rendered by the oracle with the same block seeds agrees bit-exactly
    osc = orc.Scene(sd)
    crop_blocks = [bx * 68 + by for bx in range(40, 42) for by in range(30, 32)]
    for bidx in crop_blocks:
        ref, _ = osc.render(seeds=seeds, spp=4, stream_mode=1, eval_order=1, shard_index=bidx, shard_count=120 * 68)
        x0, y0 = (bidx // 68) * 16, (bidx % 68) * 16
        np.testing.assert_array_equal(a[y0:y0 + 16, x0:x0 + 16], ref[y0:y0 + 16, x0:x0 + 16])


def _oracle_blocks_match(sd, osc, img, seeds, blocks, stream_mode=1, **okw):
    """`blocks`: indices b = (x/16) * ceil(H/16) + y/16 — each is rendered alone by the oracle (a shard of one block) and compared bit for bit."""
    nby = (sd.height + 15) // 16
    nb = ((sd.width + 15) // 16) * nby
    verts = 0
    for b in blocks:
        ref, ost = osc.render(seeds=seeds, stream_mode=stream_mode, eval_order=1, shard_index=int(b), shard_count=nb, **okw)
        x0, y0 = (b // nby) * 16, (b % nby) * 16
        np.testing.assert_array_equal(img[y0:y0 + 16, x0:x0 + 16], ref[y0:y0 + 16, x0:x0 + 16], err_msg=f"block {b} at ({x0}, {y0})")
        verts += ost["vertices"]
    return verts


def _busiest_blocks(img, n):
    """The n 16x16 blocks with the largest pixel variance (geometry edges, glass, caustic noise) — 'blocks over geometry'."""
    H, W = img.shape[:2]
    nby = (H + 15) // 16
    lum = img.sum(-1)
    scores = []
    for bx in range((W + 15) // 16):
        for by in range(nby):
            t = lum[by * 16:by * 16 + 16, bx * 16:bx * 16 + 16]
            scores.append((float(np.var(t)) if np.isfinite(t).all() else 0.0, bx * nby + by))
    scores.sort(reverse=True)
    return [b for _, b in scores[:n]]


def test_full_size_mixed_materials(built):
    """BASELINE cfg 3 at full size (1920x1080, 508 k triangles, 6 BSDF types; 4 spp): deep BVH streamed from L2 / HBM, global stack
    overflow levels, the dynamic pixel dispenser and the run-time BSDF switch at scale.  fused == wavefront (material sort) bitwise
    with equal counters, and 6 blocks over geometry equal the oracle's bits (src/accel.rs:243-343)."""
    sd = scenes.living_room(1920, 1080)
    assert sd.n_triangles > 500000
    ctx = api.Context(api.Scene(sd), 0)
    assert not ctx.debug_sizes()["lds_scene"]
    seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    a, st = ctx.render(seeds, api.path_params(spp=4))
    assert st["iterations"] == 1                                   # auto = the persistent fused kernel
    c, stc = ctx.render(seeds, api.path_params(spp=4, pipeline=api.PIPELINE_WAVEFRONT))
    np.testing.assert_array_equal(a, c)
    assert all(st[k] == stc[k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
    assert st["camera_samples"] == 1920 * 1080 * 4 and np.isfinite(a).all()
    osc = orc.Scene(sd)
    blocks = _busiest_blocks(a, 4) + [60 * 68 + 34, 30 * 68 + 50]
    assert _oracle_blocks_match(sd, osc, a, seeds, blocks, spp=4) > 4000
    _full_size_default_mode(sd, ctx, osc, seeds, blocks, spp=8, min_verts=8000)


def _full_size_default_mode(sd, ctx, osc, seeds, blocks, spp, min_verts):
    """The same full-size frame in the drop-in default, RL_STREAM_REFERENCE_ORDER (rustlight's own stream assignment, integrators/mod.rs:403-450), in two
    passes: the blocks equal the oracle's walk of the same block streams bit for bit, and the two forms of the chain pass — the speculative one
    (k_stream_spec, forced: `spp` is too small for it to be chosen) and the one-lane-per-block walk (k_stream_chain) — give the same frame and counters."""
    keys = ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws")
    with ctx.options(spec_force=1):
        r, st = ctx.render(seeds, api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER))
    assert st["ms_prepass"] > 0.0 and st["spec_group"] > 0 and st["camera_samples"] == sd.width * sd.height * spp and np.isfinite(r).all()
    assert _oracle_blocks_match(sd, osc, r, seeds, blocks, stream_mode=0, spp=spp) > min_verts
    with ctx.options(chain_serial=1):
        r2, st2 = ctx.render(seeds, api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER))
    assert st2["spec_group"] == 0
    np.testing.assert_array_equal(r, r2)
    assert all(st[k] == st2[k] for k in keys)


def test_full_size_medium(built):
    """BASELINE cfg 5 at full size (cbox + homogeneous medium sigma_s = 0.5, 1920x1080, 2 spp; mean path length ~19 vertices):
    fused (pixel dispenser) == wavefront bitwise, counters equal, 6 blocks equal the oracle's bits (src/volume.rs:95-141)."""
    sd = scenes.cbox_medium(1920, 1080, 0.5)
    ctx = api.Context(api.Scene(sd), 0)
    seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    a, st = ctx.render(seeds, api.path_params(spp=2))
    c, stc = ctx.render(seeds, api.path_params(spp=2, pipeline=api.PIPELINE_WAVEFRONT))
    np.testing.assert_array_equal(a, c)
    assert all(st[k] == stc[k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
    assert st["vertices"] > 10 * st["camera_samples"] and np.isfinite(a).all()
    osc = orc.Scene(sd)
    blocks = _busiest_blocks(a, 4) + [60 * 68 + 34, 10 * 68 + 5]
    assert _oracle_blocks_match(sd, osc, a, seeds, blocks, spp=2) > 20000
    _full_size_default_mode(sd, ctx, osc, seeds, blocks[:4], spp=8, min_verts=40000)


def test_cfg1_reference_order_whole_frame(built):
    """BASELINE cfg 1 exactly: cbox 256x256x16 spp in RL_STREAM_REFERENCE_ORDER — the one mode that is rustlight's own stream
    assignment (one SmallRng per 16x16 block consumed over (iy, ix, sample), src/integrators/mod.rs:357-371,420-435) — whole frame
    against the oracle, both pipelines."""
    sd = scenes.cbox(256, 256)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    ref_fwd, ost = osc.render(master_seed=0, spp=16, stream_mode=0, eval_order=1)
    ref_rec, _ = osc.render(master_seed=0, spp=16, stream_mode=0, eval_order=0)
    for pipeline in (api.PIPELINE_WAVEFRONT, api.PIPELINE_AUTO):
        img, st = ctx.render(api.IndependentSampler(0).block_seeds(256, 256), api.path_params(spp=16, stream_mode=api.STREAM_REFERENCE_ORDER, pipeline=pipeline))
        _assert_parity(img, st, ref_fwd, ref_rec, ost)
    assert st["camera_samples"] == 256 * 256 * 16


def test_reference_order_two_pass_equals_single_pass(built, monkeypatch):
    """RL_STREAM_REFERENCE_ORDER through the persistent kernel runs in two passes (k_stream_chain records the sampler state at the start of
    every camera sample with the radiance half of the integrator left out; the per-sample kernel evaluates them): the image and every counter
    must be those of the single-pass walk (RL_REF_SINGLE_PASS, the form rounds 1-2 shipped) and of the oracle, bit for bit — on every scene
    feature that changes how many draws a sample takes (smooth BSDFs skip the light sample, media add distance / phase draws, strategies,
    depth limits, Russian roulette off), with the states cut into several chunks (RL_STATE_BUDGET_MB) and through the streaming kernels."""
    ref_mode = api.STREAM_REFERENCE_ORDER
    cases = [(scenes.cbox(70, 41), dict(spp=5)),
             (scenes.cbox(48, 48), dict(spp=3, strategy=api.STRATEGY_BSDF, max_depth=6)),
             (scenes.cbox(48, 48), dict(spp=3, strategy=api.STRATEGY_EMITTER)),
             (scenes.cbox(48, 48), dict(spp=2, rr_depth=None, max_depth=9)),
             (scenes.cbox(48, 48), dict(spp=3, rr_depth=3, min_depth=1, max_depth=7)),
             (scenes.cbox(32, 32), dict(spp=2, max_depth=1)),
             (scenes.cbox_medium(40, 40, 0.8, 0.2, g=0.6), dict(spp=3)),
             (scenes.cbox_medium(32, 32, 0.5), dict(spp=2, single_scattering=True)),
             (scenes.living_room(64, 48, n_spheres=27, tess=10), dict(spp=4, max_depth=10)),          # glass / mirror / phong / substrate
             (scenes.cbox_other_lights(48, 48), dict(spp=3)),
             (scenes.sky_scene(48, 48), dict(spp=3, min_depth=1)),
             (scenes.many_lights(48, 48, n=5, use_ats=True), dict(spp=2, max_depth=4))]
    for sd, kw in cases:
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        two = _render_pair(sd, ctx, osc, seed=3, stream_mode=ref_mode, **kw)
        _assert_parity(*two)
        assert two[1]["ms_prepass"] > 0.0 and two[1]["ms_other"] > 0.0          # both passes ran
        with ctx.options(ref_single_pass=1):
            one = _render_pair(sd, ctx, osc, seed=3, stream_mode=ref_mode, **kw)
        assert one[1]["ms_prepass"] == 0.0
        np.testing.assert_array_equal(two[0], one[0])
        assert all(two[1][k] == one[1][k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
        for split in (0, 3):
            img, st = ctx.render(api.IndependentSampler(3).block_seeds(sd.width, sd.height), api.path_params(stream_mode=ref_mode, sample_split=split, **kw))
            np.testing.assert_array_equal(img, two[0])
    # tiny scenes: the chain pass precomputes its ray's node / triangle records on the idle lanes of the chain's group (32 or 64 lanes; 42 triangles need
    # two passes of a 32-lane group, and the 36 of the Cornell box too) — against the plain one-lane traversal (RL_CHAIN_NO_PRE) and the oracle
    for sd, kw in ((scenes.cbox(96, 80), dict(spp=6)), (scenes.many_lights(64, 48, n=3, use_ats=False), dict(spp=4, max_depth=6)), (scenes.cbox_other_lights(48, 48), dict(spp=3))):
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        assert ctx.debug_sizes()["lds_scene"] and sd.n_triangles <= 64
        base = _render_pair(sd, ctx, osc, seed=9, stream_mode=ref_mode, **kw)
        _assert_parity(*base)
        for env in (dict(item_shift=5), dict(item_shift=6), dict(chain_no_pre=1), dict(item_shift=4)):
            with ctx.options(**env):
                img, st = ctx.render(api.IndependentSampler(9).block_seeds(sd.width, sd.height), api.path_params(stream_mode=ref_mode, **kw))
            np.testing.assert_array_equal(img, base[0], err_msg=str(env))
            assert all(st[k] == base[1][k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
    # several chunks: 1 MB of states = a few block cursors per chunk on this frame (130 blocks x 24 spp x 32 B per cursor)
    sd = scenes.cbox(160, 200)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    whole = _render_pair(sd, ctx, osc, seed=1, stream_mode=ref_mode, spp=24)
    _assert_parity(*whole)
    ctx.set_option("state_budget_mb", 1)
    img, st = ctx.render(api.IndependentSampler(1).block_seeds(160, 200), api.path_params(stream_mode=ref_mode, spp=24))
    assert st["iterations"] > 10 and st["chunks"] == st["iterations"]             # chunks
    assert st["overlapped"] == 1                                                  # (round 6: the evaluation pass runs beside the chain pass chunk by chunk)
    np.testing.assert_array_equal(img, whole[0])
    assert all(st[k] == whole[1][k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
    # a budget that not even one cursor position of every block fits (130 blocks x 300 spp x 32 B > 1 MB): the single-pass walk takes over, same image
    big, stb = ctx.render(api.IndependentSampler(2).block_seeds(160, 200), api.path_params(stream_mode=ref_mode, spp=300))
    assert stb["ms_prepass"] == 0.0 and stb["chunks"] == 0
    ctx.set_option("state_budget_mb", None)
    big2, stb2 = ctx.render(api.IndependentSampler(2).block_seeds(160, 200), api.path_params(stream_mode=ref_mode, spp=300))
    assert stb2["ms_prepass"] > 0.0 and stb2["rng_draws"] == stb["rng_draws"]
    np.testing.assert_array_equal(big, big2)
    # shards: the chains of a shard's blocks only
    parts = [ctx.render(api.IndependentSampler(1).block_seeds(160, 200), api.path_params(stream_mode=ref_mode, spp=24, shard_index=r, shard_count=3))[0] for r in range(3)]
    np.testing.assert_array_equal(parts[0] + parts[1] + parts[2], whole[0])
    # two UNEVEN chunks (140 + 116 block cursors of a 1280 x 800 frame): the smaller chunk asks for more lanes per pixel (9 against 7), i.e. more slots and statistics rows than
    # the larger one — the counters must not lose them (round 4: sized from the largest chunk only, 1 % of the draws of a 1080p render went missing)
    sd = scenes.cbox(1280, 800)
    ctx = api.Context(api.Scene(sd), 0)
    seeds = api.IndependentSampler(2).block_seeds(1280, 800)
    a, sa = ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=8))
    with ctx.options(state_budget_mb=143):
        b, sb = ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=8))
    assert sa["iterations"] == 1 and sb["iterations"] == 2 and sa["chunks"] == 1 and sb["chunks"] == 2 and sb["overlapped"] == 1
    np.testing.assert_array_equal(a, b)
    assert all(sa[k] == sb[k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))


def test_overlapped_evaluation_pass_equals_back_to_back(built, monkeypatch):
    """Reference-order streams, round 5: the evaluation pass (k_path_fused from the recorded sampler states) runs BESIDE the chain pass — the chain kernels flag every
    block whose sample states are all recorded (a release fence, then the render's tag into the block's word in mapped host memory), the host thread inside rl_render_path
    polls those words and launches k_path_fused<.., QUEUE = true> over explicit lists of complete blocks on the context's low-priority streams; nothing on the device
    waits — instead of after it (option no_overlap: the two passes back to back, the form rounds 3-4 shipped).  Same samples from the same states, folded per pixel in
    sample order: the image and every counter must be identical, and identical to the oracle's — on LDS-staged and streamed scenes (the streamed ones keep a second set
    of overflow stack levels for the kernel that runs beside the chain kernel), with a medium, through both chain kernels (speculative, forced; serial), ragged frames,
    shards with several lanes per pixel, a frame cut into several chunks (round 6: every chunk's evaluation runs beside its chain pass), and repeatedly on one context
    (the flags are re-tagged per render).  rl_render_stats.overlapped / .chunks say which form ran.  src/integrators/mod.rs:420-448."""
    ref_mode = api.STREAM_REFERENCE_ORDER
    keys = ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws")
    cases = [(scenes.cbox(70, 41), dict(spp=24), {}),
             (scenes.cbox(96, 64), dict(spp=40), dict(spec_force=1)),
             (scenes.cbox_medium(40, 40, 0.8, 0.2, g=0.6), dict(spp=6), {}),
             (scenes.living_room(64, 48, n_spheres=27, tess=10), dict(spp=12, max_depth=10), {}),
             (scenes.living_room(64, 48, n_spheres=27, tess=10), dict(spp=16, max_depth=10), dict(spec_force=1, RL_FORCE_STREAMING="1")),
             (scenes.cbox(48, 48), dict(spp=9), dict(RL_FORCE_STREAMING="1")),
             (scenes.cbox(160, 200), dict(spp=24), dict(state_budget_mb=1)),          # several chunks: the evaluation pass beside the chain pass chunk by chunk (round 6)
             (scenes.many_lights(48, 48, n=5, use_ats=True), dict(spp=6, max_depth=4), {}),
             (scenes.cbox(160, 200), dict(spp=32, sample_split=4), {}),
             (scenes.cbox(160, 200), dict(spp=32, shard_index=1, shard_count=3), dict(spec_force=1))]
    for n_case, (sd, kw, env) in enumerate(cases):
        # (RL_* entries: the creation-time options, which a context takes from the environment; the others: rl_context_set_option)
        for k, v in env.items():
            if k.startswith("RL_"): monkeypatch.setenv(k, v)
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        for k, v in env.items():
            if not k.startswith("RL_"): ctx.set_option(k, v)
        seeds = api.IndependentSampler(5).block_seeds(sd.width, sd.height)
        with ctx.options(no_overlap=1):
            base, st0 = ctx.render(seeds, api.path_params(stream_mode=ref_mode, **kw))
        assert st0["overlapped"] == 0 and st0["chunks"] >= 1
        for rep in range(3):
            img, st = ctx.render(seeds, api.path_params(stream_mode=ref_mode, **kw))
            np.testing.assert_array_equal(img, base, err_msg=f"case {n_case} rep {rep}")
            assert all(st[k] == st0[k] for k in keys), (n_case, rep)
            assert st["ms_prepass"] > 0.0 and st["overlapped"] == 1 and st["chunks"] == st0["chunks"], (n_case, rep, st["overlapped"], st["chunks"])
            assert st["ms_eval_span"] >= st["ms_other"] > 0.0 or st["ms_eval_span"] > 0.0
        okw = {k: v for k, v in kw.items() if k != "sample_split"}
        ref, ost = osc.render(seeds=seeds, stream_mode=0, eval_order=1, **okw)
        np.testing.assert_array_equal(base, ref, err_msg=f"case {n_case} vs the oracle")
        assert all(st0[k] == ost[k] for k in ("vertices", "rng_draws", "shadow_rays", "extension_rays")), n_case
        for k in env:
            if k.startswith("RL_"): monkeypatch.delenv(k)


def test_rng_advance_equals_stepping(built):
    """rng_advance (csrc/kernels/rngjump.h: the jump polynomials x^(2^b) mod P of Xoshiro256's state transition, derived by scratch/r4/xoshiro_jump.py and
    checked there against the generator's published JUMP / LONG_JUMP constants) against plain stepping of the oracle's sampler, on the device."""
    import ctypes as C
    L = api.lib()
    n = 192
    rng = np.random.default_rng(5)
    states = rng.integers(1, 2**63, size=(n, 4), dtype=np.uint64)
    counts = np.concatenate([[0, 1, 2, 255, 256, 257, 511, 512, 1023, 65536, 65537, 1000003], rng.integers(0, 200000, size=n - 12)]).astype(np.uint32)
    out = np.zeros_like(states)
    L.rl_debug_rng_advance.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.rl_debug_rng_advance(0, n, states.ctypes.data, counts.ctypes.data, out.ctypes.data) == 0
    for i in range(n):
        r = orc.Rng.from_state([int(x) for x in states[i]])
        for _ in range(int(counts[i])):
            orc.lib().orc_rng_next_u64(r.state)
        assert list(r.state) == [int(x) for x in out[i]], (i, int(counts[i]))


def test_reference_order_speculative_chain(built, monkeypatch):
    """k_stream_spec (csrc/kernels/spec.hip.h), the first pass of reference-order streams with every lane busy: per-pixel windows of the block stream are
    walked speculatively and the true chain is threaded through the tracks.  Whatever the windows, group shapes, capacities and estimates are, the recorded
    sampler states — hence the image and every counter — must be those of the one-lane-per-block walk (RL_CHAIN_SERIAL: k_stream_chain) and of the oracle:
    on every feature that changes a draw count, on ragged frames, with tiny track capacities (the slow path does the work), no lead-in, no margins, no
    probes, without the trivial-pixel shortcut, in several chunks, on shards and through the streaming kernels.  These frames are far too small for the
    pass to be chosen by itself (RL_SPEC_FORCE)."""
    ref_mode = api.STREAM_REFERENCE_ORDER
    cases = [(scenes.cbox(70, 41), dict(spp=24)),
             (scenes.cbox(48, 48), dict(spp=9, strategy=api.STRATEGY_BSDF, max_depth=6)),
             (scenes.cbox(48, 48), dict(spp=7, rr_depth=None, max_depth=5)),
             (scenes.cbox(32, 32), dict(spp=5, max_depth=1)),
             (scenes.cbox_medium(40, 40, 0.8, 0.2, g=0.6), dict(spp=6)),
             (scenes.living_room(64, 48, n_spheres=27, tess=10), dict(spp=12, max_depth=10)),          # glass / mirror / phong / substrate
             (scenes.sky_scene(48, 48), dict(spp=8, min_depth=1)),
             (scenes.many_lights(48, 48, n=5, use_ats=True), dict(spp=6, max_depth=4))]
    shapes = [dict(spec_group=16, spec_sub=1), dict(spec_group=32, spec_sub=2), dict(spec_group=64, spec_sub=4), dict(spec_group=64, spec_sub=8),
              dict(spec_group=32, spec_sub=1, spec_cap=5), dict(spec_group=64, spec_sub=2, spec_lead=0, spec_ks=0, spec_ke=0, spec_probe=0),
              dict(spec_group=16, spec_sub=4, spec_no_trivial=1, spec_ks=5, spec_ke=5),
              dict(spec_group=256, spec_sub=16), dict(spec_group=256, spec_sub=1, spec_extra=1), dict(spec_group=64, spec_sub=16, spec_extra=1, spec_probe_every=1),
              dict(spec_group=32, spec_sub=2, spec_dense=32, spec_dense_frac=0), dict(spec_group=64, spec_sub=1, spec_dense=3, spec_dense_frac=0, spec_lead=0, spec_ks=0, spec_ke=0),
              dict(spec_group=16, spec_sub=2, spec_dense=0)]   # 256: a whole workgroup per block (votes and sums through the barrier)
    keys = ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws")
    for n_case, (sd, kw) in enumerate(cases):
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        seeds = api.IndependentSampler(7).block_seeds(sd.width, sd.height)
        with ctx.options(chain_serial=1):
            base = _render_pair(sd, ctx, osc, seed=7, stream_mode=ref_mode, **kw)
        _assert_parity(*base)
        assert base[1]["spec_group"] == 0
        ctx.set_option("spec_force", 1)
        for env in (shapes if n_case < 2 else shapes[n_case % 4::4]):
            with ctx.options(**env):
                img, st = ctx.render(seeds, api.path_params(stream_mode=ref_mode, **kw))
            assert st["spec_group"] == int(env["spec_group"]), env
            np.testing.assert_array_equal(img, base[0], err_msg=f"case {n_case} {env}")
            assert all(st[k] == base[1][k] for k in keys), (n_case, env)
        ctx.set_option("spec_force", None)
    # chunks (the chain is parked between them), shards, and the kernels that stream the BVH
    sd = scenes.cbox(160, 200)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    seeds = api.IndependentSampler(1).block_seeds(160, 200)
    with ctx.options(chain_serial=1):
        whole = _render_pair(sd, ctx, osc, seed=1, stream_mode=ref_mode, spp=40)
    _assert_parity(*whole)
    ctx.set_option("spec_force", 1)
    img, st = ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=40))
    assert st["spec_group"] > 0 and st["spec_samples"] > 0
    np.testing.assert_array_equal(img, whole[0])
    with ctx.options(state_budget_mb=2):
        img, st = ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=40))
    assert st["iterations"] > 5 and st["spec_group"] > 0 and st["chunks"] == st["iterations"]
    np.testing.assert_array_equal(img, whole[0])
    parts = [ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=40, shard_index=r, shard_count=3))[0] for r in range(3)]
    np.testing.assert_array_equal(parts[0] + parts[1] + parts[2], whole[0])
    monkeypatch.setenv("RL_FORCE_STREAMING", "1")
    sctx = api.Context(api.Scene(sd), 0)
    monkeypatch.delenv("RL_FORCE_STREAMING")
    assert not sctx.debug_sizes()["lds_scene"]
    sctx.set_option("spec_force", 1)
    for env in (dict(spec_group=64, spec_sub=4), dict(spec_group=64, spec_sub=2, spec_cap=5, spec_lead=0, spec_ks=0, spec_ke=0, spec_lds_levels=0), dict(spec_group=32, spec_sub=1)):
        with sctx.options(**env):
            img, st = sctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=40))
        assert st["spec_group"] == int(env["spec_group"])
        np.testing.assert_array_equal(img, whole[0], err_msg=str(env))
    # a device whose workgroups may not ask for the pass's LDS (53 KB on this scene): the serial chain renders the frame instead of a refused launch
    with ctx.options(spec_lds_limit_test=32768):
        img, st = ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=40))
    assert st["spec_group"] == 0 and st["ms_prepass"] > 0.0
    np.testing.assert_array_equal(img, whole[0])
    # the policy: k_stream_spec when a pixel has at least 4 x as many samples as a sample takes draws — 12 on a scene without a medium unless the host says
    # otherwise (option spec_draws_per_sample) — so 40 spp is left to the serial chain unless forced and 96 spp is enough.  The choice is a pure function
    # of scene, parameters and options: a context's first frame and its fifth take the same kernels (it used to follow the draws per
    # sample the context's PREVIOUS render had measured)
    ctx.set_option("spec_force", None)
    assert ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=8))[1]["spec_group"] == 0
    assert ctx.render(seeds, api.path_params(stream_mode=ref_mode, spp=40))[1]["spec_group"] == 0
    fresh = api.Context(api.Scene(sd), 0)
    choice = []
    for frame in range(5):
        img, st = fresh.render(seeds, api.path_params(stream_mode=ref_mode, spp=96))
        choice.append(st["spec_group"])
        if frame == 2: fresh.render(seeds, api.path_params(stream_mode=ref_mode, spp=8, max_depth=2))       # (a render with other parameters in between changes nothing)
    assert choice[0] > 0 and len(set(choice)) == 1, choice
    with fresh.options(spec_draws_per_sample=30):
        assert fresh.render(seeds, api.path_params(stream_mode=ref_mode, spp=96))[1]["spec_group"] == 0
    with pytest.raises(api.RustlightError): fresh.set_option("no_such_option", 1)
    with pytest.raises(api.RustlightError): fresh.set_option("force_streaming", 1)        # (creation-time: the environment when the context is made)
    assert fresh.get_option("spec_force") is None and ctx.get_option("spec_force") is None


def test_full_size_reference_order(built):
    """BASELINE cfg 2's frame in the drop-in default mode, RL_STREAM_REFERENCE_ORDER (rustlight's own stream assignment,
    src/integrators/mod.rs:420-435), 1920x1080 x 8 spp through the two-pass form: 8 blocks (the four busiest + fixed ones inside and outside
    the box) equal the oracle's walk of the same block streams bit for bit, the single-pass walk gives the same frame and counters."""
    sd = scenes.cbox(1920, 1080)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    a, st = ctx.render(seeds, api.path_params(spp=8, stream_mode=api.STREAM_REFERENCE_ORDER))
    assert st["camera_samples"] == 1920 * 1080 * 8 and st["ms_prepass"] > 0.0 and np.isfinite(a).all()
    nby, nb = 68, 120 * 68
    verts = 0
    for b in _busiest_blocks(a, 4) + [60 * 68 + 34, 45 * 68 + 20, 75 * 68 + 50, 2 * 68 + 3]:
        ref, ost = osc.render(seeds=seeds, spp=8, stream_mode=0, eval_order=1, shard_index=int(b), shard_count=nb)
        x0, y0 = (b // nby) * 16, (b % nby) * 16
        np.testing.assert_array_equal(a[y0:y0 + 16, x0:x0 + 16], ref[y0:y0 + 16, x0:x0 + 16], err_msg=f"block {b} at ({x0}, {y0})")
        verts += ost["vertices"]
    assert verts > 8000
    with ctx.options(ref_single_pass=1):
        b1, st1 = ctx.render(seeds, api.path_params(spp=8, stream_mode=api.STREAM_REFERENCE_ORDER))
    np.testing.assert_array_equal(a, b1)
    assert all(st[k] == st1[k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
    _full_size_default_mode(sd, ctx, osc, seeds, [60 * 68 + 34, 66 * 68 + 46, 66 * 68 + 6, 26 * 68 + 34], spp=8, min_verts=4000)


def test_bench_frames_equal_the_oracle(built):
    """tests/golden/bench_crcs.json (made by tests/golden/make_bench_golden.py in the build container: the parity build of the CPU oracle renders the very
    frames bench.py times) against the GPU: BASELINE configs[0] whole (256 x 256 x 16 spp, reference-order streams) and configs[1] at full size in
    rustlight's own reference-order streams (1920 x 1080 x 128 spp, master seed 2 = the last of bench.py's three timed steps) — the CRC-32 of the float32
    image.  bench.py prints the same comparison as `oracle_crc_match` / `reference_order_oracle_crc_match` for every frame in the table."""
    import json
    import zlib
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_crcs.json")))
    for w, h, spp, seed in ((256, 256, 16, 0), (1920, 1080, 128, 2)):
        e = table[f"cbox:{w}x{h}x{spp}:reference:seed{seed}"]
        img, st = api.Context(api.Scene(scenes.cbox(w, h)), 0).render(api.IndependentSampler(seed).block_seeds(w, h), api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER))
        assert f"{zlib.crc32(img.tobytes()):08x}" == e["crc32"], (w, h, spp)
        assert (st["camera_samples"], st["vertices"], st["rng_draws"]) == (e["camera_samples"], e["vertices"], e["rng_draws"])
        assert (st["spec_group"] > 0) == (spp == 128)          # the large frame goes through k_stream_spec, the small one through k_stream_chain


def test_cfg4_one_rank_of_eight(built):
    """BASELINE configs[3]'s per-rank workload exactly: cbox 1920 x 1080, 1024 spp, the blocks of rank 0 of 8 (b % 8 == 0: the round-robin deal of compute_mc's block
    list, src/integrators/mod.rs:351-450 / DESIGN.md 5), in both stream modes.  The whole shard against the oracle's render of the same shard (CRC and counters from
    tests/golden/bench_crcs.json, made by tests/golden/make_bench_golden.py with the parity build) and eight blocks of it — the busiest four + fixed ones inside and
    outside the box — against the oracle's walk of the same block streams run here, bit for bit.  In reference-order streams the chain pass must be the speculative one."""
    import json
    import zlib
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_crcs.json")))
    sd = scenes.cbox(1920, 1080)
    ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
    seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    nby, nb = 68, 120 * 68
    for mode, name, omode in ((api.STREAM_PER_SAMPLE, "per_sample", 1), (api.STREAM_REFERENCE_ORDER, "reference", 0)):
        img, st = ctx.render(seeds, api.path_params(spp=1024, shard_index=0, shard_count=8, stream_mode=mode))
        e = table[f"cbox:1920x1080x1024:{name}:seed0:shard0of8"]
        assert f"{zlib.crc32(img.tobytes()):08x}" == e["crc32"], name
        assert (st["camera_samples"], st["vertices"], st["rng_draws"]) == (e["camera_samples"], e["vertices"], e["rng_draws"]), name
        assert st["camera_samples"] == sum(min(16, 1920 - (b // nby) * 16) * min(16, 1080 - (b % nby) * 16) for b in range(0, nb, 8)) * 1024
        if mode == api.STREAM_REFERENCE_ORDER:
            assert st["ms_prepass"] > 0.0 and st["spec_group"] > 0
        owned = [b for b in _busiest_blocks(img, 64) if b % 8 == 0][:4] + [60 * 68 + 32, 45 * 68 + 20, 75 * 68 + 44, 2 * 68 + 0]
        assert len(owned) == 8 and all(b % 8 == 0 for b in owned)
        verts = 0
        for b in owned:
            ref, ost = osc.render(seeds=seeds, spp=1024, stream_mode=omode, eval_order=1, shard_index=int(b), shard_count=nb)
            x0, y0 = (b // nby) * 16, (b % nby) * 16
            np.testing.assert_array_equal(img[y0:y0 + 16, x0:x0 + 16], ref[y0:y0 + 16, x0:x0 + 16], err_msg=f"{name}: block {b} at ({x0}, {y0})")
            verts += ost["vertices"]
        assert verts > 500000, verts
        # a block of another rank stays black on this one
        assert not img[16:32, 0:16].any()


def test_cfg4_whole_frame_on_one_gpu(built):
    """BASELINE configs[3] WHOLE, as far as one GPU allows: cbox 1920 x 1080 x 1024 spp cut into the eight shards of `b % 8` (the round-robin deal of compute_mc's
    block list, src/integrators/mod.rs:351-374) and merged as accumulate_bitmap merges the blocks' bitmaps (mod.rs:445-448) — through rl_multi_create(scene, 8 shards
    on device 0) + rl_multi_render_path: eight device contexts, eight host threads, the co-resident shards added on the device, the RCCL clique of the one device, one
    download.  The summed frame and the summed counters must be the oracle's render of the whole 1024-spp frame (CRC and counters from tests/golden/bench_crcs.json,
    made by the parity build), in both stream modes; and `bench.py --gpus 8` (eight ranks sharing the device: plumbing mode) must report the same CRC match.  What is
    then still unrun of configs[3] is the xGMI hop of the reduce."""
    import json
    import subprocess
    import sys
    import zlib
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_crcs.json")))
    scene = api.Scene(scenes.cbox(1920, 1080))
    seeds = api.IndependentSampler(0).block_seeds(1920, 1080)
    mc = api.MultiContext(scene, 8)
    assert mc.info()["shards"] == 8
    for mode, name in ((api.STREAM_PER_SAMPLE, "per_sample"), (api.STREAM_REFERENCE_ORDER, "reference")):
        e = table[f"cbox:1920x1080x1024:{name}:seed0"]
        img, st = mc.render(seeds, api.path_params(spp=1024, stream_mode=mode))
        assert f"{zlib.crc32(img.tobytes()):08x}" == e["crc32"], name
        assert (st["camera_samples"], st["vertices"], st["rng_draws"]) == (e["camera_samples"], e["vertices"], e["rng_draws"]) and st["camera_samples"] == 1920 * 1080 * 1024, name
        per_shard = [mc.shard_stats(g)[1] for g in range(8)]
        assert sum(s["camera_samples"] for s in per_shard) == st["camera_samples"] and all(s["camera_samples"] > 0 for s in per_shard)
        # shard 0 of the eight is the shard test_cfg4_one_rank_of_eight holds against the oracle's own render of it
        e0 = table[f"cbox:1920x1080x1024:{name}:seed0:shard0of8"]
        assert (per_shard[0]["camera_samples"], per_shard[0]["vertices"], per_shard[0]["rng_draws"]) == (e0["camera_samples"], e0["vertices"], e0["rng_draws"]), name
        if mode == api.STREAM_REFERENCE_ORDER:
            assert all(s["ms_prepass"] > 0.0 and s["spec_group"] > 0 for s in per_shard)
    mc.close()
    # bench.py's own N-rank path on the same frame: eight ranks, one process each, sharing the one device (reduce through gloo — flagged `devices_shared`)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-also", "--no-verify"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["config"]["spp_total"] == 1024 and out["distributed"]["devices_shared"] is True
    assert out["oracle_crc32"] == table["cbox:1920x1080x1024:per_sample:seed0"]["crc32"] and out["oracle_crc_match"] is True


def test_multi_context_rccl_reduce(built, cbox64, ctx_cbox):
    """rl_multi_*: N shards from ONE process — one device context + host thread per shard, per-device framebuffers in HBM, shards that
    share a device added on the device, ONE ncclReduce over the distinct devices (a 1-rank clique on this box: the RCCL call path
    itself runs), one download.  The image is the single-context image bit for bit, counters add up."""
    seeds = api.IndependentSampler(5).block_seeds(64, 64)
    full, st = ctx_cbox.render(seeds, api.path_params(spp=4))
    scene = api.Scene(cbox64)
    for n in (1, 3):
        mc = api.MultiContext(scene, n)
        info = mc.info()
        assert info["shards"] == n and info["comm_ranks"] >= 1 and info["rccl_version"] > 0
        img, mst = mc.render(seeds, api.path_params(spp=4))
        np.testing.assert_array_equal(img, full)
        assert all(mst[k] == st[k] for k in ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws"))
        d = mc.describe()
        assert d["shards"] == n and d["merge"].startswith("ncclReduce") and len(d["devices"]) == d["comm_ranks"] and len(d["last_render"]["kernel_ms"]) == n
        assert all(dev["peer_access"][i] == 1 for i, dev in enumerate(d["devices"])) and d["devices"][0]["cus"] > 0
        assert sum(mc.shard_stats(g)[1]["camera_samples"] for g in range(n)) == st["camera_samples"]
        mc.close()
    # the merge step when RCCL is not available (communicator cannot be built / a reduce fails): per-device sums added on the host — same bits,
    # reported in describe(); RL_MULTI_NO_FALLBACK turns it into an error instead
    os.environ["RL_MULTI_FORCE_HOST_MERGE"] = "1"
    try:
        mc = api.MultiContext(scene, 3)
        img, mst = mc.render(seeds, api.path_params(spp=4, stream_mode=api.STREAM_REFERENCE_ORDER))
        ref, _ = ctx_cbox.render(seeds, api.path_params(spp=4, stream_mode=api.STREAM_REFERENCE_ORDER))
        np.testing.assert_array_equal(img, ref)
        assert mc.describe()["merge"].startswith("host sum") and mc.info()["comm_ranks"] == 0
        mc.close()
        os.environ["RL_MULTI_NO_FALLBACK"] = "1"
        with pytest.raises(api.RustlightError):
            api.MultiContext(scene, 2)
    finally:
        os.environ.pop("RL_MULTI_FORCE_HOST_MERGE", None); os.environ.pop("RL_MULTI_NO_FALLBACK", None)


def test_bench_two_ranks_on_one_gpu(built):
    """`python bench.py --gpus 2` starts its own two ranks; on a one-GPU box they share the device and reduce through gloo (plumbing
    mode, flagged).  Both ranks render their tiles with the KERNELS; the reduced image must carry the CRC of the 1-GPU render."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--width", "320", "--height", "200", "--spp", "4", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["distributed"]["world_size"] == 2 and len(out["distributed"]["ranks"]) == 2
    assert out["distributed"]["crc_match"] is True and out["config"]["spp_total"] == 8
    assert out["value"] > 0 and out["roofline"]["kernel"] == "k_path_fused"
    # the same shards once more in rustlight's own reference-order streams (the drop-in default), per rank: chain pass, evaluation pass, reduce
    ro = out["reference_order"]
    assert out["reference_order_value"] == ro["value"] > 0 and len(ro["ranks"]) == 2
    assert all(r["chain_ms"] > 0 and r["eval_span_ms"] > 0 and r["eval_tail_after_chain_ms"] >= 0 and r["reduce_ms"] >= 0 for r in ro["ranks"])
    assert all("reduce_ms_per_step" in r for r in out["distributed"]["ranks"])
    # ... and with three frames in flight per rank (three contexts each): the last frame's reduced image is the one-at-a-time image of that frame
    fl = ro["three_frames_in_flight"]
    assert fl["errors"] is None and fl["value"] > 0 and out["reference_order_in_flight_value"] == fl["value"]
    ctx = api.Context(api.Scene(scenes.cbox(320, 200)), 0)
    img, _ = ctx.render(api.IndependentSampler(fl["frames"] - 1).block_seeds(320, 200), api.path_params(spp=8, stream_mode=api.STREAM_REFERENCE_ORDER))
    import zlib
    assert fl["image_crc32_last_frame"] == f"{zlib.crc32(img.tobytes()):08x}"


def test_fast_numerics_tolerance_mode(built):
    """`numerics = fast` (opt-in; FMA contraction, v_rcp / v_rsq / v_sqrt, hardware sin / cos / exp2 / log2 — DESIGN.md §2 "Tolerance
    mode") is held to BASELINE.json's own bar, not to bit-exactness: RNG sequence bit-exact (same seeds, same draws on every path whose
    branch decisions do not flip), per-pixel squared L2 vs the oracle < 1e-3.  The exact build stays the default and the only one
    the other parity tests bless."""
    # (scene, render options, bound on the 99.9th percentile): SURVEY §8(d) makes the MEAN per-pixel squared L2 the target (< 1e-3) and asks for
    # the 99.9th percentile and the maximum to be reported.  On the diffuse scenes 99.9 % of the pixels are inside the tolerance even at these low
    # sample counts; with mirrors and glass a path whose decision flips (a ray grazing an edge now passes on the other side) can move a pixel by a
    # whole light-source sample / spp, so there the mean is bounded by the tolerance and single pixels only in number (1080p numbers: profiles/r03_fast_numerics_parity.json).
    cases = [(scenes.cbox(96, 96), dict(spp=16), L2_TOL), (scenes.living_room(96, 64, n_spheres=27, tess=10), dict(spp=8, max_depth=8), None),
             (scenes.cbox_medium(64, 64, 0.5), dict(spp=4), L2_TOL)]
    for sd, kw, p999_bound in cases:
        ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
        seeds = api.IndependentSampler(3).block_seeds(sd.width, sd.height)
        ref, ost = osc.render(seeds=seeds, stream_mode=1, eval_order=1, **kw)
        exact, st = ctx.render(seeds, api.path_params(**kw))
        fast, stf = ctx.render(seeds, api.path_params(numerics=api.NUMERICS_FAST, **kw))
        np.testing.assert_array_equal(exact, ref)
        assert not np.array_equal(fast, ref)                       # it really is another build
        e = per_pixel_l2(fast, ref)
        assert e.mean() < L2_TOL and np.isfinite(fast).all(), (e.mean(), np.quantile(e, 0.999), e.max())
        if p999_bound is not None:
            assert e.mean() < 1e-5 and np.quantile(e, 0.999) < p999_bound, (e.mean(), np.quantile(e, 0.999), e.max())
        else:
            # glossy scene (mirrors, glass; the tolerance build also traverses quantised BVH4 nodes here): no bound on single pixels, but on how MANY miss the
            # per-pixel tolerance and by how much at the 99.9th percentile (measured over three seeds at 8 spp: 0.18-0.42 % of the pixels over 1e-3, p99.9
            # 1.6e-3 ... 5.2e-3; at 64 spp 0.02-0.08 %) — BASELINE's bar holds in the mean, not per pixel, on such scenes, and the README says so
            assert (e > L2_TOL).mean() < 0.015 and np.quantile(e, 0.999) < 2e-2, ((e > L2_TOL).mean(), np.quantile(e, 0.999))
        assert abs(float(fast.mean()) / float(ref.mean()) - 1.0) < 0.01          # never systematic: the image mean agrees within 1 %
        assert stf["camera_samples"] == ost["camera_samples"]
        # paths whose decisions flipped change the draw count: a fraction of a percent
        assert abs(stf["rng_draws"] - ost["rng_draws"]) <= 5e-3 * ost["rng_draws"], (stf["rng_draws"], ost["rng_draws"])
        assert abs(stf["vertices"] - ost["vertices"]) <= 5e-3 * ost["vertices"], (stf["vertices"], ost["vertices"])
    # reference-order streams in the tolerance build.  Here ONE flipped decision (a path that takes another number of draws) shifts where every later sample of its
    # 16x16 block starts in the block's stream: from there on the block renders other — equally valid — samples than the exact build, so pixels differ by plain
    # Monte Carlo noise and only statistics can be held: finite, mean per-pixel L2 inside the tolerance at this sample count, image mean within 1 %, path census
    # within 0.5 %, and the same against the single-pass walk of the same build (the per-pixel bar of `numerics = fast` is a per-sample-stream statement)
    for sd, kw in ((scenes.cbox(64, 64), dict(spp=16)), (scenes.living_room(64, 48, n_spheres=27, tess=10), dict(spp=16, max_depth=8))):
        ctx = api.Context(api.Scene(sd), 0)
        seeds = api.IndependentSampler(5).block_seeds(sd.width, sd.height)
        exact, st = ctx.render(seeds, api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, **kw))
        fast, stf = ctx.render(seeds, api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, numerics=api.NUMERICS_FAST, **kw))
        with ctx.options(ref_single_pass=1):
            one, st1 = ctx.render(seeds, api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, numerics=api.NUMERICS_FAST, **kw))
        assert stf["ms_prepass"] > 0.0 and st1["ms_prepass"] == 0.0 and np.isfinite(fast).all() and np.isfinite(one).all()
        # what two independent renders of the scene differ by at this sample count (another master seed): the yardstick for "plain Monte Carlo noise"
        other, _ = ctx.render(api.IndependentSampler(6).block_seeds(sd.width, sd.height), api.path_params(stream_mode=api.STREAM_REFERENCE_ORDER, **kw))
        noise = per_pixel_l2(other, exact).mean()
        for img, stx in ((fast, stf), (one, st1)):
            assert per_pixel_l2(img, exact).mean() < 1.5 * noise + 1e-6, (per_pixel_l2(img, exact).mean(), noise)
            assert abs(stx["vertices"] - st["vertices"]) <= 5e-3 * st["vertices"] and abs(float(img.mean()) / float(exact.mean()) - 1.0) < 0.05
        assert stf["camera_samples"] == st["camera_samples"] == st1["camera_samples"]
    with pytest.raises(api.RustlightError, match="persistent kernel"):
        ctx.render(seeds, api.path_params(spp=1, numerics=api.NUMERICS_FAST, pipeline=api.PIPELINE_WAVEFRONT))
    with pytest.raises(api.RustlightError, match="numerics"):
        ctx.render(seeds, api.path_params(spp=1, numerics=7))


def test_scene_from_pod_description_renders_the_same_image(built):
    """The one-call POD entry (rl_scene_create_from_desc) and the builder calls describe the same scene to the kernels."""
    for sd in (scenes.cbox(48, 32), scenes.living_room(48, 32, n_spheres=8, tess=6), scenes.cbox_medium(32, 32, 0.5)):
        seeds = api.IndependentSampler(6).block_seeds(sd.width, sd.height)
        a, sa = api.Context(api.Scene(sd), 0).render(seeds, api.path_params(spp=3, max_depth=6))
        b, sb = api.Context(api.Scene.from_desc(sd), 0).render(seeds, api.path_params(spp=3, max_depth=6))
        np.testing.assert_array_equal(a, b)
        assert sa["vertices"] == sb["vertices"] and sa["rng_draws"] == sb["rng_draws"]


def _c_initializer(v):
    """A ctypes value as a C99 initializer (nested structs / arrays; pointers are not used here)."""
    import ctypes as C
    if isinstance(v, C.Structure):
        return "{" + ", ".join(f".{n} = {_c_initializer(getattr(v, n))}" for n, _ in v._fields_) + "}"
    if isinstance(v, C.Array):
        return "{" + ", ".join(_c_initializer(x) for x in v) + "}"
    if isinstance(v, float):
        return float(np.float32(v)).hex() + "f"
    return str(int(v))


def test_uv_dependent_emission_parity(built):
    """`-x hvs-light` / `-x texture-light` (examples/cli.rs:410-429): every light mesh's emission becomes EmissionType::HSV { scale } / Texture { scale, img }
    (src/geometry.rs:99-104), evaluated by Mesh::emit at the hit's uv (vertex.rs:69-82) and at the sampled point's normalised uv (emitter.rs:609-688), with
    Color::value(scale) as its flux and the triangle-centre emission in the light tree — image bits and counters of the oracle in both stream modes, both
    pipelines, with the light tree, through `direct`, and on a many-light scene whose lights have uv."""
    keys = ("camera_samples", "vertices", "extension_rays", "shadow_rays", "rng_draws")
    tex = (4, 3, np.random.default_rng(2).uniform(0.0, 2.0, (12, 3)).astype(np.float32))
    for kind in ("hsv", "texture"):
        for use_ats in (False, True):
            sd = scenes.cbox(40, 32)
            if kind == "texture": sd.bitmaps.append(tex)
            scenes.override_light_emission(sd, kind, bitmap_id=0)
            sd.use_ats = use_ats
            ctx, osc = api.Context(api.Scene(sd), 0), orc.Scene(sd)
            for mode in (api.STREAM_PER_SAMPLE, api.STREAM_REFERENCE_ORDER):
                for pipe in (api.PIPELINE_AUTO, api.PIPELINE_WAVEFRONT):
                    got = _render_pair(sd, ctx, osc, seed=5, spp=6, stream_mode=mode, pipeline=pipe)
                    _assert_parity(*got)
            seeds = api.IndependentSampler(5).block_seeds(sd.width, sd.height)
            img, st = ctx.render_direct(seeds, spp=4, nb_bsdf_samples=1, nb_light_samples=2)
            ref, ost = osc.render_direct(seeds=seeds, spp=4, nb_bsdf_samples=1, nb_light_samples=2)
            np.testing.assert_array_equal(img, ref)
            assert all(st[k] == ost[k] for k in ("camera_samples", "extension_rays", "shadow_rays", "rng_draws"))
            # the one-call POD carries the emission kind too
            img2, st2 = api.Context(api.Scene.from_desc(sd), 0).render(seeds, api.path_params(spp=3))
            np.testing.assert_array_equal(img2, ctx.render(seeds, api.path_params(spp=3))[0])
    # a light mesh without uv cannot take a uv-dependent emission (the reference's `uv.unwrap()` would panic): refused
    sd = scenes.cbox(16, 16)
    sd.meshes[-1].uv = None
    scenes.override_light_emission(sd, "hsv")
    with pytest.raises(api.RustlightError):
        api.Scene(sd)


def test_verbatim_reference_scene_renders_like_the_fixture(built):
    """examples/web/index.html:9-43 verbatim (tests/golden/web_cbox_verbatim.pbrt) through rl_scene_load_pbrt renders the image of the in-memory
    fixture — and of the oracle — bit for bit, in rustlight's own reference-order streams."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "web_cbox_verbatim.pbrt")
    sd = scenes.cbox(512, 512)
    seeds = api.IndependentSampler(4).block_seeds(512, 512)
    pp = api.path_params(spp=2, stream_mode=api.STREAM_REFERENCE_ORDER)
    img, st = api.Context(api.Scene.load_pbrt(path), 0).render(seeds, pp)
    ref, st2 = api.Context(api.Scene(sd), 0).render(seeds, pp)
    np.testing.assert_array_equal(img, ref)
    oimg, ost = orc.Scene(sd).render(master_seed=4, spp=2, stream_mode=0, eval_order=1)
    np.testing.assert_array_equal(img, oimg)
    assert all(st[k] == st2[k] == ost[k] for k in ("camera_samples", "vertices", "extension_rays", "rng_draws"))


def test_c99_consumer_renders_the_same_image(built, tmp_path):
    """The drop-in boundary from plain C (gcc -std=c99 -pedantic, nothing but include/rustlight_amd.h + the shared library) — what a Rust FFI
    caller does, minus Rust: rl_scene_create_from_desc -> rl_context_create -> rl_generate_block_seeds -> rl_render_path with the CLI's
    default parameters (reference-order streams).  Image and counters equal the ctypes binding's and the oracle's, bit for bit.  The consumer then renders
    the same frame twice more through rl_render_path_frames (two contexts, two frames in flight) and exits non-zero unless both equal the first."""
    import subprocess
    from rustlight_amd import abi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W, H, spp, seed = 48, 32, 3, 11
    sd = scenes.cbox(W, H)
    lines = ["#include <stdint.h>", f"#define SCENE_WIDTH {W}", f"#define SCENE_HEIGHT {H}", f"#define SCENE_SPP {spp}", f"#define SCENE_SEED {seed}ull",
             "#define SCENE_STREAM_MODE RL_STREAM_REFERENCE_ORDER", f"#define SCENE_N_MESHES {len(sd.meshes)}",
             f"static const float scene_fov = {float(np.float32(sd.fov)).hex()}f;", f"static const int scene_fov_axis = {int(sd.fov_axis)};", f"static const int scene_flip = {int(sd.flip)};",
             "static const float scene_to_world[16] = {" + ", ".join(float(x).hex() + "f" for x in np.asarray(sd.to_world, np.float32).ravel()) + "};",
             # the camera as rustlight's Camera holds it (camera.rs:5-15): what the consumer passes (rl_scene_desc.has_camera_matrices)
             "static const float scene_sample_to_camera[16] = {" + ", ".join(float(x).hex() + "f" for x in api.Scene(sd).camera_matrices()[0]) + "};"]
    for i, m in enumerate(sd.meshes):
        v, idx, n, uv, e = abi.mesh_arrays(m)
        lines.append(f"static const float mesh{i}_v[] = {{" + ", ".join(float(x).hex() + "f" for x in v.ravel()) + "};")
        lines.append(f"static const uint32_t mesh{i}_i[] = {{" + ", ".join(str(int(x)) for x in idx.ravel()) + "};")
        if n is not None: lines.append(f"static const float mesh{i}_n[] = {{" + ", ".join(float(x).hex() + "f" for x in n.ravel()) + "};")
        if uv is not None: lines.append(f"static const float mesh{i}_uv[] = {{" + ", ".join(float(x).hex() + "f" for x in uv.ravel()) + "};")
    lines.append("static const rl_mesh_desc scene_meshes[SCENE_N_MESHES] = {")
    for i, m in enumerate(sd.meshes):
        v, idx, n, uv, e = abi.mesh_arrays(m)
        em = "{0.0f, 0.0f, 0.0f}" if e is None else "{" + ", ".join(float(x).hex() + "f" for x in e) + "}"
        lines.append(f"    {{.vertices = mesh{i}_v, .n_vertices = {v.shape[0]}, .indices = mesh{i}_i, .n_triangles = {idx.shape[0]}, .normals = {'mesh%d_n' % i if n is not None else '0'}, "
                     f".uv = {'mesh%d_uv' % i if uv is not None else '0'}, .bsdf = {_c_initializer(abi.bsdf_desc(m.bsdf))}, .has_emission = {int(e is not None)}, .emission_rgb = {em}}},")
    lines.append("};")
    (tmp_path / "scene_data.h").write_text("\n".join(lines) + "\n")
    exe = tmp_path / "render_desc"
    libdir = os.path.join(root, "rustlight_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-I", str(tmp_path),
                           os.path.join(root, "tests", "c_consumer", "render_desc.c"), "-o", str(exe), "-L", libdir, "-lrustlight_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe), str(tmp_path / "img.raw")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    img = np.fromfile(tmp_path / "img.raw", dtype=np.float32).reshape(H, W, 3)
    # the same consumer with Camera::new's arguments instead of the matrices (-DSCENE_CAMERA_FROM_FOV): the same image
    exe2 = tmp_path / "render_desc_fov"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-DSCENE_CAMERA_FROM_FOV", "-I", os.path.join(root, "include"), "-I", str(tmp_path),
                           os.path.join(root, "tests", "c_consumer", "render_desc.c"), "-o", str(exe2), "-L", libdir, "-lrustlight_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out2 = subprocess.run([str(exe2), str(tmp_path / "img2.raw")], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr
    np.testing.assert_array_equal(np.fromfile(tmp_path / "img2.raw", dtype=np.float32).reshape(H, W, 3), img)
    counters = dict(zip(out.stdout.split()[0::2], (int(x) for x in out.stdout.split()[1::2])))
    ref, st = api.Context(api.Scene(sd), 0).render(api.IndependentSampler(seed).block_seeds(W, H), api.path_params(spp=spp, stream_mode=api.STREAM_REFERENCE_ORDER))
    np.testing.assert_array_equal(img, ref)
    oimg, ost = orc.Scene(sd).render(master_seed=seed, spp=spp, stream_mode=0, eval_order=1)
    np.testing.assert_array_equal(img, oimg)
    assert all(counters[k] == st[k] == ost[k] for k in ("camera_samples", "vertices", "extension_rays", "rng_draws")) and counters["shadow_rays"] == st["shadow_rays"]
