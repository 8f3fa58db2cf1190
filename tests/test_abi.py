"""C-ABI library: loads, exports every symbol include/rustlight_amd.h declares, host-side logic
(Mesh::new, build_emitters, BVH build, camera, PBRT loader) equals the oracle's.  CPU only —
no compute entry point is called (those need a GPU and fail loudly without one)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import orc
from rustlight_amd import api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "rustlight_amd.h")).read()
    declared = set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(api.PUBLIC_SYMBOLS), declared ^ set(api.PUBLIC_SYMBOLS)
    L = ctypes.CDLL(api.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in api.lib().rl_build_info()


def test_kernels_are_compiled_for_gfx950(built):
    data = open(api.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"k_extend" in data and b"k_shade" in data and b"k_shadow" in data and b"k_raygen" in data
    assert b"k_path_fused" in data and b"k_pixel_mc" in data


def test_library_links_rccl_for_the_framebuffer_reduce(built):
    """rl_multi_render_path reduces the per-GPU framebuffers with ncclReduce over xGMI inside the product library."""
    import subprocess
    out = subprocess.run(["readelf", "-d", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl.so" in out, out
    syms = subprocess.run(["nm", "-D", "--undefined-only", api.LIB_PATH], capture_output=True, text=True).stdout
    assert "ncclReduce" in syms and "ncclCommInitAll" in syms and "ncclGroupStart" in syms


def test_path_params_defaults_are_the_reference_cli_defaults(built):
    p = api.abi.PathParams()
    api.lib().rl_path_params_default(ctypes.byref(p))
    assert (p.spp, p.has_min_depth, p.min_depth, p.has_max_depth, p.has_rr_depth, p.rr_depth, p.strategy) == (1, 1, 0, 0, 1, 0, api.STRATEGY_ALL)
    assert p.stream_mode == api.STREAM_REFERENCE_ORDER and p.numerics == api.NUMERICS_EXACT     # seed-for-seed rustlight's streams, exact arithmetic


def test_no_gpu_means_loud_failure_not_fallback(built, cbox64):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.NoDeviceError):
        api.Context(api.Scene(cbox64))
    with pytest.raises(api.NoDeviceError):                     # the multi-GPU entry (RCCL reduce in the library) has no fallback either
        api.MultiContext(api.Scene(cbox64), 2)
    # the CLI (C++ mirror of examples/cli.rs) parses, loads and builds the scene, then refuses to render without a device
    import subprocess
    cli = os.path.join(os.path.dirname(api.LIB_PATH), "rustlight-amd")
    for args in (["-n", "1", "-o", "/tmp/never.pfm", "path"], ["-x", "ats", "-n", "1", "-o", "/tmp/never.pfm", "direct", "-b", "1"]):
        r = subprocess.run([cli, os.path.join(ROOT, "data", "cbox.pbrt"), *args], capture_output=True, text=True)
        assert r.returncode != 0 and "no CPU fallback" in r.stderr and not os.path.exists("/tmp/never.pfm")
    # `-x hvs-light` (EmissionType::HSV on every light mesh, cli.rs:410-429) is parsed and applied, then the same refusal; an unknown extra option is an error
    r = subprocess.run([cli, os.path.join(ROOT, "data", "cbox.pbrt"), "-x", "hvs-light", "-n", "1", "-o", "/tmp/never.pfm", "path"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr and not os.path.exists("/tmp/never.pfm")
    r = subprocess.run([cli, os.path.join(ROOT, "data", "cbox.pbrt"), "-x", "no-such-option", "-n", "1", "-o", "/tmp/never.pfm", "path"], capture_output=True, text=True)
    assert r.returncode == 2 and "not supported" in r.stderr


def test_error_codes(built):
    L = api.lib()
    assert L.rl_scene_create(None) == -1
    h = ctypes.c_void_p()
    assert L.rl_scene_create(ctypes.byref(h)) == 0
    assert L.rl_scene_build_emitters(h) == -1            # no camera yet
    v = np.zeros((3, 3), np.float32)
    idx = np.array([[0, 1, 7]], np.uint32)                # out-of-range index
    bd = api.abi.bsdf_desc(scenes.matte((0.5, 0.5, 0.5)))
    assert L.rl_scene_add_mesh(h, api.abi.fptr(v), 3, api.abi.u32ptr(idx), 1, None, None, ctypes.byref(bd), None) == -1
    assert L.rl_scene_add_mesh(h, api.abi.fptr(v), 3, api.abi.u32ptr(idx), 0, None, None, ctypes.byref(bd), None) == -1   # empty mesh
    L.rl_scene_destroy(h)
    with pytest.raises(api.RustlightError):
        api.Scene.load_pbrt("/nonexistent.pbrt")


@pytest.mark.parametrize("maker", [lambda: scenes.cbox(96, 64), lambda: scenes.living_room(64, 64, n_spheres=12, tess=8), scenes.single_triangle])
def test_host_bvh_and_camera_equal_oracle(built, maker):
    sd = maker()
    ps, os_ = api.Scene(sd), orc.Scene(sd)
    for a, b in zip(ps.debug_bvh(), os_.bvh()):
        np.testing.assert_array_equal(a, b)
    for px, py in [(0.0, 0.0), (10.25, 3.5), (sd.width - 0.001, sd.height - 0.5)]:
        o1, d1 = ps.camera_ray(px, py)
        o2, d2 = os_.camera_generate(px, py)
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(d1, d2)
        assert abs(float(np.dot(d1, d1)) - 1.0) < 1e-4     # Ray::new's assert_approx_eq (structure.rs:707)
    assert ps.counts()["emitters"] == os_.info()["emitters"]


def test_pixels_flagged_as_two_draw_pixels_cannot_reach_the_scene(built):
    """k_stream_spec's shortcut: a pixel whose camera rays cannot reach the scene's bounding box takes exactly two draws per sample (the jitter; without a medium a
    missed camera ray ends its path, path.rs:152-166), so its sampler states are skip-aheads.  The host decides it conservatively (rl_debug_trivial_pixels: the same
    function the render calls).  Here, host only: at 1920 x 1080 the flagged set is large (the reference's Fov::Y x aspect quirk) — and for every flagged pixel the
    ORACLE's camera rays through its four corners, its centre and random jitters miss everything, while the pixels next to the flagged region's border that do hit
    geometry are, of course, not flagged.  With a medium nothing is flagged; with max_depth <= 1 everything is."""
    import ctypes as C
    L = api.lib()
    L.rl_debug_trivial_pixels.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    for sd in (scenes.cbox(1920, 1080), scenes.living_room(320, 180, n_spheres=8, tess=6), scenes.cbox(333, 77)):
        ps, osc = api.Scene(sd), orc.Scene(sd)
        flags = np.zeros((sd.height, sd.width), np.uint8)
        assert L.rl_debug_trivial_pixels(ps.h, 0, 0, flags.ctypes.data) == 0
        frac = flags.mean()
        if sd.width == 1920: assert 0.35 < frac < 0.5, frac            # ~45 % of the 16:9 frame looks past the box
        ys, xs = np.nonzero(flags)
        rng = np.random.default_rng(3)
        pick = rng.choice(len(ys), size=min(len(ys), 20000), replace=False) if len(ys) else np.zeros(0, int)
        # the flagged pixels nearest to unflagged ones (the border of the region) are the critical ones: take all of them too
        edge = np.zeros_like(flags, bool)
        edge[:, 1:] |= (flags[:, 1:] == 1) & (flags[:, :-1] == 0); edge[:, :-1] |= (flags[:, :-1] == 1) & (flags[:, 1:] == 0)
        edge[1:, :] |= (flags[1:, :] == 1) & (flags[:-1, :] == 0); edge[:-1, :] |= (flags[:-1, :] == 1) & (flags[1:, :] == 0)
        ey, ex = np.nonzero(edge)
        px = np.concatenate([xs[pick], ex]).astype(np.float64); py = np.concatenate([ys[pick], ey]).astype(np.float64)
        o_all, d_all = [], []
        for du, dv in ((0.0, 0.0), (0.999999, 0.0), (0.0, 0.999999), (0.999999, 0.999999), (0.5, 0.5), (float(rng.random()), float(rng.random()))):
            for x, y in zip(px, py):
                o, d = osc.camera_generate(float(x + du), float(y + dv))
                o_all.append(o); d_all.append(d)
        t, u, v, m, tr = osc.trace(np.array(o_all), np.array(d_all))
        assert (m < 0).all(), f"{(m >= 0).sum()} camera rays of flagged pixels hit geometry"
        # and the shortcut is not vacuous at the border: unflagged neighbours of flagged pixels mostly do see geometry within a few pixels
        assert flags.sum() == 0 or len(ey) > 0
    with_medium = api.Scene(scenes.cbox_medium(64, 36, 0.5))
    f2 = np.zeros((36, 64), np.uint8)
    assert L.rl_debug_trivial_pixels(with_medium.h, 0, 0, f2.ctypes.data) == 0 and f2.sum() == 0
    f3 = np.zeros((36, 64), np.uint8)
    assert L.rl_debug_trivial_pixels(api.Scene(scenes.cbox(64, 36)).h, 1, 1, f3.ctypes.data) == 0 and f3.all()


def test_camera_from_matrices_is_the_same_camera(built):
    """rl_scene_set_camera_matrices / rl_scene_desc.has_camera_matrices (SURVEY 8(b) SceneDesc: `sample_to_camera[16], to_world[16]`): the camera handed over as
    the two matrices rustlight's Camera holds (camera.rs:5-15) generates the rays Camera::new's arguments give, bit for bit — through the builder call
    and through the one-call POD; non-finite matrices are refused."""
    for sd in (scenes.cbox(48, 32), scenes.living_room(40, 24, n_spheres=1, tess=4)):
        base = api.Scene(sd)
        stc, tw, pos = base.camera_matrices()
        assert np.isfinite(stc).all() and np.isfinite(tw).all()
        np.testing.assert_array_equal(tw, np.asarray(sd.to_world, np.float32).ravel())
        for other in (api.Scene(sd, camera_matrices=(stc, tw)), api.Scene.from_desc(sd, camera_matrices=(stc, tw))):
            for a, b in zip(other.camera_matrices(), (stc, tw, pos)):
                np.testing.assert_array_equal(a, b)
            for px, py in [(0.0, 0.0), (10.25, 3.5), (sd.width - 0.001, sd.height - 0.5)]:
                for a, b in zip(other.camera_ray(px, py), base.camera_ray(px, py)):
                    np.testing.assert_array_equal(a, b)
            for a, b in zip(other.debug_bvh(), base.debug_bvh()):
                np.testing.assert_array_equal(a, b)
    bad = stc.copy(); bad[5] = np.nan
    with pytest.raises(api.RustlightError):
        api.Scene(sd, camera_matrices=(bad, tw))


@pytest.mark.parametrize("maker", [lambda: scenes.many_lights(32, 32, 4, glowing_spheres=3), lambda: scenes.many_lights(16, 16, 1), lambda: scenes.cbox(16, 16)])
def test_light_tree_build_matches_the_oracle(built, maker):
    """`rl_scene_enable_ats`: the host builds LightSamplerATS (cone unions, bucketed split, itertools::partition order)
    bit for bit like the oracle — nodes, links and the reordered light list."""
    sd = maker()
    sd.use_ats = True
    got, want = api.Scene(sd).debug_ats(), orc.Scene(sd).ats_dump()
    np.testing.assert_array_equal(got[0].view(np.uint32), want[0].view(np.uint32))
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[2], want[2])
    assert got[0].shape[0] == 2 * len(got[1]) - 1
    with pytest.raises(api.RustlightError):                # assert!(e.is_surface()): point lights cannot enter the tree
        bad = scenes.cbox_other_lights(16, 16, environment=False, directional=False)
        bad.use_ats = True
        api.Scene(bad)


def test_pbrt_loader_round_trip(built, tmp_path):
    sd = scenes.cbox(128, 96)
    p = str(tmp_path / "cbox.pbrt")
    scenes.write_pbrt(sd, p)
    loaded = api.Scene.load_pbrt(p)
    direct = api.Scene(sd)
    assert loaded.size == (128, 96)
    assert loaded.counts() == direct.counts() == {"meshes": 8, "triangles": 36, "emitters": 1}
    for a, b in zip(loaded.debug_bvh(), direct.debug_bvh()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(loaded.camera_ray(5.5, 7.25)[1], direct.camera_ray(5.5, 7.25)[1])


def test_pbrt_loader_lights_and_environment_map(built, tmp_path):
    """LightSource point / distant / infinite (rgb L and mapname -> Bitmap::read_pfm + new_texture), scene_loader.rs:205-275."""
    sd = scenes.cbox_other_lights(64, 48)
    p = str(tmp_path / "lights.pbrt")
    scenes.write_pbrt(sd, p)
    assert api.Scene.load_pbrt(p).counts() == api.Scene(sd).counts() == {"meshes": 8, "triangles": 36, "emitters": 4}
    sky = scenes.sky_scene(64, 48, keep_area_light=True)
    p = str(tmp_path / "sky.pbrt")
    scenes.write_pbrt(sky, p)
    np.testing.assert_array_equal(api.load_pfm(str(tmp_path / "sky_env.pfm")), sky.environment_map)
    loaded, direct = api.Scene.load_pbrt(p), api.Scene(sky)
    assert loaded.counts() == direct.counts() == {"meshes": 4, "triangles": 28, "emitters": 2}
    np.testing.assert_array_equal(loaded.debug_emitters_cdf(), direct.debug_emitters_cdf())
    with pytest.raises(api.RustlightError):
        open(str(tmp_path / "bad.pbrt"), "w").write('WorldBegin\nLightSource "infinite" "string mapname" [ "missing.pfm" ]\nWorldEnd\n')
        api.Scene.load_pbrt(str(tmp_path / "bad.pbrt"))


def test_repo_cbox_scene_file_matches_fixture(built):
    loaded = api.Scene.load_pbrt(os.path.join(ROOT, "data", "cbox.pbrt"))
    direct = api.Scene(scenes.cbox(512, 512))
    assert loaded.size == (512, 512)
    for a, b in zip(loaded.debug_bvh(), direct.debug_bvh()):
        np.testing.assert_array_equal(a, b)


def test_save_pfm(built, tmp_path):
    img = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3) - 4.0
    p = str(tmp_path / "x.pfm")
    api.save_pfm(p, img)
    raw = open(p, "rb").read()
    assert raw.startswith(b"PF\n3 2\n-1.0\n")
    data = np.frombuffer(raw[len(b"PF\n3 2\n-1.0\n"):], dtype="<f4").reshape(2, 3, 3)
    np.testing.assert_array_equal(data, np.abs(img[::-1]))      # bottom-up rows, abs() (structure.rs:547-560)


def test_save_png_and_exr(built, tmp_path):
    import struct
    from PIL import Image

    img = np.random.default_rng(0).uniform(-0.2, 1.5, (6, 9, 3)).astype(np.float32)
    png, exr = str(tmp_path / "a.png"), str(tmp_path / "a.exr")
    api.save_image(png, img)
    api.save_image(exr, img)
    got = np.array(Image.open(png))
    want = (np.clip(np.minimum(img, 1.0), 0, None) ** (1 / 2.2) * 255).astype(np.uint8)       # Color::to_rgba
    assert got.shape == (6, 9, 3) and np.abs(got.astype(int) - want.astype(int)).max() <= 1
    raw = open(exr, "rb").read()
    assert struct.unpack("<II", raw[:8]) == (20000630, 2)
    pos, attrs = 8, {}
    while raw[pos] != 0:                                                                         # header attributes
        name = raw[pos:raw.index(b"\0", pos)].decode(); pos += len(name) + 1
        typ = raw[pos:raw.index(b"\0", pos)].decode(); pos += len(typ) + 1
        size = struct.unpack("<I", raw[pos:pos + 4])[0]; pos += 4
        attrs[name] = (typ, raw[pos:pos + size]); pos += size
    pos += 1
    assert attrs["compression"][1] == b"\0" and struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, 8, 5)
    offsets = struct.unpack("<6Q", raw[pos:pos + 48])
    for y, off in enumerate(offsets):
        yy, nbytes = struct.unpack("<iI", raw[off:off + 8])
        assert (yy, nbytes) == (y, 9 * 12)
        planes = np.frombuffer(raw[off + 8:off + 8 + nbytes], "<f4").reshape(3, 9)              # B, G, R
        np.testing.assert_array_equal(planes[::-1].T, img[y])
    with pytest.raises(api.RustlightError):
        api.save_image(str(tmp_path / "a.jpg"), img)


def test_emitters_with_non_finite_area_or_flux_are_refused(built):
    """An emissive mesh whose area is NaN / infinite (or emitters whose total flux is) would leave NaNs in the sampling cdfs — the
    reference panics on those in `sample_discrete`; the drop-in refuses the scene when the emitters are built.  A zero area is fine
    (Distribution1DConstruct::normalize skips the division)."""
    for spoil in ("nan", "inf", "huge"):
        sd = scenes.cbox(16, 16)
        light = [m for m in sd.meshes if m.emission][0]
        v = light.vertices.copy()
        if spoil == "nan": v[1, 0] = np.nan
        elif spoil == "inf": v[2, 1] = np.inf
        else: v *= np.float32(1e30)
        light.vertices = v
        with pytest.raises(api.RustlightError, match="no finite area"):
            api.Scene(sd)
    sd = scenes.cbox(16, 16)
    light = [m for m in sd.meshes if m.emission][0]
    light.vertices = np.repeat(light.vertices[:1], len(light.vertices), 0)        # zero-area light: accepted
    api.Scene(sd)
    sd = scenes.cbox(16, 16)                        # NaN in a non-emissive mesh is tolerated (those triangles are never hit) ...
    sd.meshes[0].vertices = sd.meshes[0].vertices.copy(); sd.meshes[0].vertices[0, 0] = np.nan
    api.Scene(sd)
    sd.environment = (0.5, 0.5, 0.5)                 # ... also next to an environment light: box unions skip NaNs, the bounding sphere stays finite
    api.Scene(sd)
    sd = scenes.cbox(16, 16)
    [m for m in sd.meshes if m.emission][0].emission = (float("inf"), 1.0, 1.0)
    with pytest.raises(api.RustlightError, match="total flux is not finite"):
        api.Scene(sd)


def test_scene_from_one_pod_description_equals_the_builder_calls(built):
    """rl_scene_create_from_desc ("SceneDesc POD: counts + pointers") builds the scene the builder calls build: same BVH, camera,
    emitter table, light tree; and refuses what they refuse."""
    S = scenes
    tex = S.cbox(32, 24)
    tex.bitmaps = [(3, 2, np.arange(18, dtype=np.float32).reshape(2, 3, 3) / 18.0)]
    tex.meshes[0].uv = np.zeros((len(tex.meshes[0].vertices), 2), np.float32)
    tex.meshes[0].bsdf = S.Bsdf(type=S.DIFFUSE, diffuse={"type": S.TEX_BITMAP, "bitmap_id": 0, "color0": (1, 1, 1)})
    cases = [S.cbox(40, 24), S.cbox_other_lights(24, 24), S.sky_scene(24, 20, keep_area_light=True), S.many_lights(24, 20, 3, glowing_spheres=1),
             S.cbox_medium(24, 24, 0.5, 0.1, g=0.3), S.living_room(24, 16, n_spheres=8, tess=6), tex]
    for sd in cases:
        a, b = api.Scene(sd), api.Scene.from_desc(sd)
        assert a.size == b.size and a.counts() == b.counts()
        for x, y in zip(a.debug_bvh(), b.debug_bvh()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a.debug_emitters_cdf(), b.debug_emitters_cdf())
        for x, y in zip(a.debug_ats(), b.debug_ats()):
            np.testing.assert_array_equal(x, y)
        for px, py in ((0.5, 0.5), (13.25, 7.75)):
            for x, y in zip(a.camera_ray(px, py), b.camera_ray(px, py)):
                np.testing.assert_array_equal(x, y)
    bad = S.cbox(16, 16)
    bad.meshes[1].bsdf = S.Bsdf(type=S.DIFFUSE, diffuse={"type": S.TEX_BITMAP, "bitmap_id": 3, "color0": (1, 1, 1)})
    with pytest.raises(api.RustlightError, match="bitmap"):
        api.Scene.from_desc(bad)
    with pytest.raises(api.RustlightError, match="bitmap"):
        api.Scene(bad)


def test_scale_image_refuses_what_cannot_be_a_pixel_count(built, cbox64):
    """Camera::scale_image (src/camera.rs:73-78, the CLI's -s): the scale is caller input — negative, zero, NaN, infinite or absurdly large values
    are refused before the float -> unsigned conversion (undefined behaviour outside the target's range), valid ones rescale the image only."""
    import ctypes as C
    import math
    L = api.lib()
    sc = api.Scene(cbox64)
    for bad in (0.0, -1.0, -0.5, math.nan, math.inf, -math.inf, 1e-9, 1e30):
        assert L.rl_scene_scale_image(sc.h, C.c_float(bad)) == -1, bad      # RL_ERR_INVALID_ARGUMENT
    assert sc.size == (64, 64)
    assert L.rl_scene_scale_image(sc.h, C.c_float(0.5)) == api.RL_OK and sc.size == (32, 32)
    assert L.rl_scene_scale_image(sc.h, C.c_float(3.0)) == api.RL_OK and sc.size == (96, 96)


@pytest.mark.parametrize("maker", [lambda: scenes.cbox(64, 64), lambda: scenes.living_room(64, 64, n_spheres=27, tess=12), lambda: scenes.living_room(64, 64, n_spheres=64, tess=24),
                                   lambda: scenes.single_triangle(), lambda: scenes.many_lights(64, 48, n=5, use_ats=False)])
def test_structures_derived_from_the_bvh2_are_the_same_tree(built, maker):
    """Host-only: the treelet-blocked copy of the node array (k_stream_chain on streaming scenes) is the BVH2 itself — same boxes, same leaves, every inner node in
    exactly one slot — and every BVH4 node of the tolerance build covers the BVH2 subtrees build_bvh4's collapse rule gives it, with quantised boxes that contain
    the originals (so a BVH4 traversal visits a superset of the BVH2's leaves)."""
    import ctypes as C
    sc = api.Scene(maker())
    out = (C.c_uint64 * 6)()
    fn = api.lib().rl_debug_check_derived_bvhs
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert fn(sc.h, out) == 0
    n2, slots, n4, bad, leaves4, leaves2 = (int(x) for x in out)
    assert bad == 0, list(out)
    if n2:
        assert slots % 16 == 0 and n2 <= slots <= 16 * n2 and leaves4 == leaves2 == n2 + 1 and 0 < n4 <= n2


def _nan_scene():
    """A Cornell box with a few non-finite vertices (the geometry of test_non_finite_and_degenerate_geometry_parity's kind)."""
    sd = scenes.cbox(32, 32)
    m = sd.meshes[2]
    m.vertices = m.vertices.copy()
    m.vertices[0, 1] = np.nan
    m.vertices[1, 0] = np.inf
    return sd


@pytest.mark.parametrize("maker", [lambda: scenes.cbox(64, 64), lambda: scenes.living_room(64, 64, n_spheres=27, tess=12), lambda: scenes.living_room(64, 64, n_spheres=64, tess=24),
                                   lambda: scenes.single_triangle(), lambda: scenes.many_lights(64, 48, n=5, use_ats=False), _nan_scene])
def test_two_level_records_are_the_bvh2(built, maker):
    """Host-only: the two-level node records the exact build traverses on scenes that stream their BVH (device_types.h: BvhNode2, trace.hip.h: traverse2) hold the
    BVH2 itself — every record's children and slots are the BVH2's children and grandchildren with bit-identical boxes, the union of a child's two slots IS the
    child's box (what lets the kernel derive a child's slab distances from its slots'), and walking the records reaches every inner node and every leaf once.
    src/accel.rs:183-188 (node boxes), src/structure.rs:779-784 (union_aabb)."""
    import ctypes as C
    sc = api.Scene(maker())
    out = (C.c_uint64 * 6)()
    fn = api.lib().rl_debug_check_two_level
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    assert fn(sc.h, out) == 0
    n2, expanded, leaf_children, plain, bad, leaves = (int(x) for x in out)
    assert bad == 0, list(out)
    if n2:
        assert expanded + leaf_children + plain == 2 * n2 and leaves == n2 + 1 == leaf_children
        assert expanded + plain == n2 - 1                 # every inner node but the root is some node's child
        assert plain == 0                                  # boxes made by range_box always pass the union check
