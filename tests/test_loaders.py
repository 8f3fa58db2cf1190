"""Scene / mesh / image file readers: fixtures are written to disk in each format and must come
back as the same flattened scene — same BVH (boxes, topology, primitive order), camera rays, emitter cdf."""
import os

import numpy as np
import pytest

from rustlight_amd import api, export, scenes

PROGRESSIVE_JPEG = True


def _same_scene(loaded, direct):
    assert loaded.size == direct.size
    assert loaded.counts() == direct.counts()
    for a, b in zip(loaded.debug_bvh(), direct.debug_bvh()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(loaded.debug_emitters_cdf(), direct.debug_emitters_cdf())
    for px in ((0.5, 0.5), (17.25, 3.5)):
        np.testing.assert_array_equal(loaded.camera_ray(*px)[1], direct.camera_ray(*px)[1])


def test_verbatim_reference_scene_text_loads_as_the_fixture(built):
    """The one scene text in the reference tree written by someone else — the textarea of examples/web/index.html:9-43, copied verbatim by
    tests/golden/extract_web_scene.py (pbrt-v3 exporter formatting: `Integrator`, `Sampler "sobol"`, `PixelFilter`, `"string filename"`, trailing blanks,
    `1.74846e-007` exponents, `-0`; materials declared in another order than the shapes use them) — through rl_scene_load_pbrt must be the in-memory
    fixture rustlight_amd/scenes.py builds from SURVEY App. C: same BVH (boxes, topology, primitive order), camera matrices and rays, emitter table.
    (A `-m gpu` test renders both: tests/test_gpu_parity.py::test_verbatim_reference_scene_renders_like_the_fixture.)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "web_cbox_verbatim.pbrt")
    text = open(path, encoding="utf-8").read()
    assert text.startswith('Integrator "path" "integer maxdepth" [ 65 ] \n') and 'Sampler "sobol"' in text and text.rstrip().endswith("WorldEnd")     # the verbatim text, not a regeneration
    loaded = api.Scene.load_pbrt(path)
    direct = api.Scene(scenes.cbox(512, 512))
    _same_scene(loaded, direct)
    for a, b in zip(loaded.camera_matrices(), direct.camera_matrices()):
        np.testing.assert_array_equal(a, b)


def _mts_cbox(w=96, h=64):
    sd = scenes.cbox(w, h)
    sd.flip = True                     # MTSSceneLoader: Camera::new(.., flip = true) (scene_loader.rs:337)
    sd.fov_axis = 0
    return sd


@pytest.mark.parametrize("fmt", ["obj", "ply", "serialized"])
def test_mitsuba_xml_round_trip(built, tmp_path, fmt):
    sd = _mts_cbox()
    sd.lights.append({"type": "point", "a": (0.2, 1.2, 0.1), "intensity": (1.0, 2.0, 3.0)})
    sd.medium = scenes.Medium((0.01, 0.02, 0.03), (0.5, 0.4, 0.3), scenes.PHASE_HG, 0.3)
    p = str(tmp_path / "cbox.xml")
    export.write_mitsuba(sd, p, fmt)
    loaded, direct = api.Scene.load(p), api.Scene(sd)
    _same_scene(loaded, direct)
    assert loaded.counts()["emitters"] == 2
    # faceNormals / use_shading_normals = False drop the normals; geometry and BVH stay
    flat = api.Scene.load_mitsuba(p, use_shading_normals=False)
    for a, b in zip(flat.debug_bvh(), direct.debug_bvh()):
        np.testing.assert_array_equal(a, b)


def test_mitsuba_builtin_shapes_transforms_and_materials(built, tmp_path):
    xml = """<scene version="0.6.0">
      <default name="res" value="48"/>
      <sensor type="perspective"><float name="fov" value="40"/><string name="fovAxis" value="y"/>
        <transform name="toWorld"><lookat origin="0, 1, 5" target="0, 1, 0" up="0, 1, 0"/></transform>
        <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="32"/></film></sensor>
      <bsdf type="twosided" id="wall"><bsdf type="diffuse"><rgb name="reflectance" value="0.5, 0.25, 0.125"/></bsdf></bsdf>
      <bsdf type="roughconductor" id="gold"><string name="distribution" value="ggx"/><float name="alpha" value="0.2"/>
        <rgb name="eta" value="0.14, 0.37, 1.44"/><rgb name="k" value="3.98, 2.38, 1.6"/><float name="extEta" value="1"/></bsdf>
      <shape type="rectangle"><transform name="toWorld"><scale x="2" y="3"/><rotate x="1" angle="-90"/><translate y="0.5"/></transform><ref id="wall"/></shape>
      <shape type="sphere"><point name="center" x="0.25" y="1" z="0"/><float name="radius" value="0.5"/><ref id="gold"/></shape>
      <shape type="rectangle"><transform name="toWorld"><matrix value="0.5 0 0 0  0 0 0.5 3  0 -0.5 0 0  0 0 0 1"/></transform>
        <bsdf type="phong"><float name="exponent" value="50"/><spectrum name="specularReflectance" value="0.3"/><rgb name="diffuseReflectance" value="0.2, 0.3, 0.4"/></bsdf>
        <emitter type="area"><rgb name="radiance" value="5, 4, 3"/></emitter></shape>
      <shape type="cube"/>
      <emitter type="spot"/>
    </scene>"""
    p = str(tmp_path / "builtin.xml")
    open(p, "w").write(xml)
    sc = api.Scene.load(p)
    assert sc.size == (48, 32)
    assert sc.counts() == {"meshes": 3, "triangles": 2 + 31 * 31 * 2 + 2, "emitters": 1}      # cube / spot: ignored, as the reference does
    o, d = sc.camera_ray(24.0, 16.0)
    np.testing.assert_allclose(o, [0, 1, 5], atol=1e-6)
    np.testing.assert_allclose(d, [0, 0, -1], atol=1e-5)
    boxes = sc.debug_bvh()[0]
    np.testing.assert_allclose(boxes[0][:3], [-2, 0.5 - 1e-4, -3], atol=2e-4)            # the floor rectangle: scale, rotate about x, lift
    np.testing.assert_allclose(boxes[0][3:], [2, 3 + 1e-4, 3], atol=2e-4)                 # ... up to the emitter quad the <matrix> puts at y = 3


def test_pbrt_plymesh_include_instances_and_textures(built, tmp_path):
    sd = scenes.cbox(64, 48)
    tall, short = sd.meshes[6], sd.meshes[5]
    export.write_ply(tall, str(tmp_path / "tall.ply"), "ascii")
    rng = np.random.default_rng(1)
    tex = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    export.write_png(tex, str(tmp_path / "tex.png"))
    np.testing.assert_array_equal(api.load_image(str(tmp_path / "tex.png")), tex.astype(np.float32) / np.float32(255.0))
    gray = rng.integers(0, 256, (4, 3), dtype=np.uint8)
    export.write_png(gray, str(tmp_path / "gray.png"))
    np.testing.assert_array_equal(api.load_image(str(tmp_path / "gray.png"))[..., 1], gray.astype(np.float32) / np.float32(255.0))
    open(str(tmp_path / "mats.pbrt"), "w").write(
        'Texture "wood" "spectrum" "imagemap" "string filename" [ "tex.png" ]\n'
        'MakeNamedMaterial "woody" "string type" [ "matte" ] "texture Kd" [ "wood" ]\n')
    body = ['Film "image" "integer xresolution" [ 64 ] "integer yresolution" [ 48 ]',
            "Transform [ " + " ".join(repr(float(x)) for x in scenes.CBOX_WORLD_TO_CAMERA) + " ]",
            f'Camera "perspective" "float fov" [ {scenes.CBOX_FOV} ]', "WorldBegin", 'Include "mats.pbrt"',
            'ObjectBegin "box"', 'NamedMaterial "woody"',
            'Shape "trianglemesh" "integer indices" [ ' + " ".join(str(int(i)) for i in short.indices.reshape(-1)) + ' ] "point P" [ '
            + " ".join(repr(float(x)) for x in short.vertices.reshape(-1)) + ' ] "float uv" [ ' + " ".join(repr(float(x)) for x in short.uv.reshape(-1)) + " ]",
            "ObjectEnd",
            "AttributeBegin", 'AreaLightSource "diffuse" "rgb L" [ 3 2 1 ]', "Translate 0 0.5 0", 'Shape "plymesh" "string filename" "tall.ply"', "AttributeEnd",
            "AttributeBegin", "Translate 1 0 0", 'ObjectInstance "box"', "AttributeEnd",
            "Translate -1 0 0.5", 'ObjectInstance "box"', "WorldEnd"]
    p = str(tmp_path / "inst.pbrt")
    open(p, "w").write("\n".join(body) + "\n")
    loaded = api.Scene.load(p)
    # the same scene assembled by hand: the plain shape first, then the two instances (scene_loader.rs:170-204)
    def moved(m, t, **kw):
        return scenes.MeshData(m.name, (m.vertices + np.asarray(t, np.float32)).astype(np.float32), m.indices, m.normals, m.uv, kw.get("bsdf", m.bsdf), kw.get("emission"))
    woody = scenes.Bsdf(type=scenes.DIFFUSE, diffuse={"type": scenes.TEX_BITMAP, "color0": (0, 0, 0), "bitmap_id": 0})
    direct = scenes.SceneData(64, 48, scenes.CBOX_FOV, 1, np.asarray(scenes.CBOX_TO_WORLD, np.float32), False,
                              [moved(tall, (0, 0.5, 0), bsdf=scenes.matte((0.5, 0.5, 0.5)), emission=(3.0, 2.0, 1.0)), moved(short, (1, 0, 0), bsdf=woody), moved(short, (-1, 0, 0.5), bsdf=woody)],
                              bitmaps=[(7, 5, (tex.astype(np.float32) / np.float32(255.0)).reshape(-1))])
    _same_scene(loaded, api.Scene(direct))
    with pytest.raises(api.RustlightError):
        open(str(tmp_path / "bad.pbrt"), "w").write('WorldBegin\nShape "plymesh" "string filename" "missing.ply"\nWorldEnd\n')
        api.Scene.load(str(tmp_path / "bad.pbrt"))
    with pytest.raises(api.RustlightError):
        api.Scene.load(str(tmp_path / "scene.json"))


def test_pbrt_transform_directives(built, tmp_path):
    """Rotate / Scale / Translate / CoordinateSystem / CoordSysTransform compose like pbrt's CTM (each one is post-multiplied)."""
    body = ['Film "image" "integer xresolution" [ 16 ] "integer yresolution" [ 16 ]', "LookAt 0 0 5  0 0 0  0 1 0", 'Camera "perspective" "float fov" [ 40 ]',
            "WorldBegin", 'CoordinateSystem "world0"', "Translate 1 0 0", "Rotate 90 0 0 1", "Scale 2 1 1",
            'Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  1 0 0  0 1 0 ]',
            'CoordSysTransform "world0"', 'MediumInterface "" ""',
            'Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  1 0 0  0 1 0 ]', "WorldEnd"]
    p = str(tmp_path / "xf.pbrt")
    open(p, "w").write("\n".join(body) + "\n")
    sc = api.Scene.load(p)
    assert sc.counts()["triangles"] == 2
    boxes = sc.debug_bvh()[0]
    # first triangle: scale x by 2, rotate 90 deg about z (x -> y), then shift by +1 in x: spans x in [0, 1], y in [0, 2]
    np.testing.assert_allclose(boxes[0][:3], [0, 0, -1e-4], atol=2e-4)
    np.testing.assert_allclose(boxes[0][3:], [1, 2, 1e-4], atol=2e-4)
    o, d = sc.camera_ray(8.0, 8.0)
    np.testing.assert_allclose(o, [0, 0, 5], atol=1e-5)
    np.testing.assert_allclose(d, [0, 0, -1], atol=1e-5)


def test_serialized_versions_and_obj_polygons(built, tmp_path):
    sd = _mts_cbox(32, 32)
    for version, dbl in ((3, False), (4, True)):
        export.write_serialized(sd.meshes, str(tmp_path / "m.serialized"), version=version, double=dbl)
        xml = ['<scene version="0.5.0"><sensor type="perspective"><float name="fov" value="19.5"/><transform name="toWorld"><matrix value="'
               + " ".join(repr(float(x)) for x in np.asarray(sd.to_world, np.float32).reshape(4, 4).T.reshape(-1)) + '"/></transform>'
               '<film type="hdrfilm"><integer name="width" value="32"/><integer name="height" value="32"/></film></sensor>']
        for i, m in enumerate(sd.meshes):
            kd = m.bsdf.diffuse["color0"]
            em = f'<emitter type="area"><rgb name="radiance" value="{m.emission[0]}, {m.emission[1]}, {m.emission[2]}"/></emitter>' if m.emission else ""
            xml.append(f'<shape type="serialized"><string name="filename" value="m.serialized"/><integer name="shapeIndex" value="{i}"/>'
                       f'<bsdf type="diffuse"><rgb name="reflectance" value="{kd[0]}, {kd[1]}, {kd[2]}"/></bsdf>{em}</shape>')
        p = str(tmp_path / f"s{version}.xml")
        open(p, "w").write("\n".join(xml) + "</scene>\n")
        _same_scene(api.Scene.load(p), api.Scene(sd))
    # OBJ: a quad face is fan-triangulated, negative indices count from the end, `g` starts a model
    open(str(tmp_path / "quad.obj"), "w").write("v -1 0 -1\nv -1 0 1\nv 1 0 1\nv 1 0 -1\ng floor\nf -4 -3 -2 -1\n")
    xml = ('<scene version="0.5.0"><sensor type="perspective"><film type="hdrfilm"/></sensor>'
           '<shape type="obj"><string name="filename" value="quad.obj"/></shape></scene>')
    open(str(tmp_path / "q.xml"), "w").write(xml)
    sc = api.Scene.load(str(tmp_path / "q.xml"))
    assert sc.counts() == {"meshes": 1, "triangles": 2, "emitters": 0} and sc.size == (768, 576)


def test_malformed_files_fail_with_an_error_not_a_crash(built, tmp_path):
    """Truncated / corrupted inputs come back as an error code + message through the C-ABI — no C++ exception, abort or
    undecodable message (found by byte-flipping the fixtures; the reference panics on the same inputs)."""
    sd = _mts_cbox(32, 32)
    p = str(tmp_path / "c.xml")
    export.write_mitsuba(sd, p, "ply")
    ply = sorted(f for f in os.listdir(tmp_path) if f.endswith(".ply"))[0]
    raw = bytearray(open(tmp_path / ply, "rb").read())
    hdr = raw.index(b"end_header\n") + len(b"end_header\n")
    good = bytes(raw)
    # a face-list count of 0xff.. : used to reach std::vector(n) with a huge n
    for off in range(hdr, len(raw)):
        raw[off] = 0xFF
    open(tmp_path / ply, "wb").write(bytes(raw))
    with pytest.raises(api.RustlightError):
        api.Scene.load(p)
    open(tmp_path / ply, "wb").write(good[: hdr + 5])                 # truncated body
    with pytest.raises(api.RustlightError):
        api.Scene.load(p)
    open(tmp_path / ply, "wb").write(good)
    api.Scene.load(p)                                                 # and the intact file still loads
    # non-UTF-8 bytes inside a PBRT file end up in the error text
    bad = str(tmp_path / "bad.pbrt")
    open(bad, "wb").write(b'Camera "perspective"\nWorldBegin\nShape "\xad\xfe\xff" "integer indices" [0 1 2]\nWorldEnd\n')
    with pytest.raises(api.RustlightError):
        api.Scene.load(bad)
    for name, blob in (("t.pfm", b"PF\n4 4\n-1.0\n\x00\x00"), ("t.png", b"\x89PNG\r\n\x1a\n" + b"\x00" * 40),
                       ("huge.pfm", b"PF\n99999999 99999999\n-1.0\n")):
        open(tmp_path / name, "wb").write(blob)
        with pytest.raises(api.RustlightError):
            api.load_image(str(tmp_path / name))


def test_openexr_reader(built, tmp_path):
    """Bitmap::read_exr (structure.rs:607-640): the R, G, B channels as f32 — scanline files, NONE / RLE / ZIPS / ZIP, HALF / FLOAT,
    a dataWindow that does not start at the origin, an extra channel to skip; our own writer's files; PIZ is refused with an error."""
    rng = np.random.default_rng(0)
    img = (rng.uniform(0, 4, (37, 23, 3)) ** 3).astype(np.float32)
    img[5:9] = 0.25                                     # long runs for the RLE / predictor paths
    img[0, 0] = (1e-6, 65000.0, 0.0)
    for comp in ("none", "rle", "zips", "zip"):
        for half in (False, True):
            for kw in (dict(), dict(data_origin=(-3, 7), extra_channel=True)):
                p = str(tmp_path / f"t_{comp}_{int(half)}_{len(kw)}.exr")
                export.write_exr(img, p, comp, half, **kw)
                want = img.astype(np.float16).astype(np.float32) if half else img
                np.testing.assert_array_equal(api.load_image(p), want, err_msg=p)
    p = str(tmp_path / "own.exr")
    api.save_image(p, img)
    np.testing.assert_array_equal(api.load_image(p), img)
    raw = bytearray(open(p, "rb").read())
    k = raw.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    raw[k] = 4                                          # PIZ
    open(p, "wb").write(bytes(raw))
    with pytest.raises(api.RustlightError, match="PIZ"):
        api.load_image(p)
    # an .exr environment map behaves like the .pfm one
    sd = scenes.sky_scene(32, 24, keep_area_light=True)
    pb = str(tmp_path / "sky.pbrt")
    scenes.write_pbrt(sd, pb)
    txt = open(pb).read()
    env = [f for f in os.listdir(tmp_path) if f.endswith(".pfm")][0]
    export.write_exr(api.load_image(str(tmp_path / env)), str(tmp_path / "env.exr"), "zip", False)
    open(pb, "w").write(txt.replace(env, "env.exr"))
    _same_scene(api.Scene.load(pb), api.Scene(sd))


def test_baseline_jpeg_reader(built, tmp_path):
    """read_ldr_image for .jpg textures: files written by Pillow / libjpeg-turbo in the authoring container (tests/golden/jpeg_fixture.npz,
    made by make_jpeg_fixture.py) decode to exactly the pixels Pillow gets — 4:4:4, 4:2:2 with optimised Huffman tables, 4:2:0 with
    restart markers, greyscale, progressive (spectral selection + successive approximation); value / 255 as read_ldr_image does."""
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_fixture.npz"))
    names = sorted(k[:-5] for k in fx.files if k.endswith("_file"))
    assert len(names) == 5
    for name in names:
        p = str(tmp_path / (name + ".jpg"))
        open(p, "wb").write(fx[name + "_file"].tobytes())
        if name.startswith("progressive") and not PROGRESSIVE_JPEG:
            with pytest.raises(api.RustlightError, match="progressive"):
                api.load_image(p)
            continue
        got = api.load_image(p)
        np.testing.assert_array_equal(got, fx[name + "_rgb"].astype(np.float32) / np.float32(255.0), err_msg=name)
    for cut in (2, 200, 700):                       # truncated files: an error or a partial image, never a crash
        p = str(tmp_path / f"cut{cut}.jpg")
        open(p, "wb").write(fx["yuv420_q60_rst_file"].tobytes()[:cut])
        try:
            api.load_image(p)
        except api.RustlightError:
            pass


def test_tga_reader(built, tmp_path):
    """read_ldr_image for .tga: raw / run-length encoded, 24 / 32-bit true colour and 8-bit grey, bottom-up (default) and top-down."""
    import struct
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)
    img[3:6] = img[3, 0]                                  # runs

    def tga(pixels_bgr, typ, bpp, top_down, rle):
        h, w = pixels_bgr.shape[:2]
        rows = pixels_bgr if top_down else pixels_bgr[::-1]
        flat = rows.reshape(h * w, -1)
        if rle:
            body = bytearray(); i = 0
            while i < len(flat):
                j = i
                while j + 1 < len(flat) and j - i < 127 and (flat[j + 1] == flat[i]).all(): j += 1
                if j > i: body += bytes([0x80 | (j - i)]) + flat[i].tobytes(); i = j + 1
                else: body += bytes([0]) + flat[i].tobytes(); i += 1
        else:
            body = flat.tobytes()
        return struct.pack("<BBBHHBHHHHBB", 0, 0, typ, 0, 0, 0, 0, 0, w, h, bpp, 0x20 if top_down else 0) + bytes(body)
    want = img.astype(np.float32) / np.float32(255.0)
    bgr = img[:, :, ::-1]
    bgra = np.concatenate([bgr, np.full((9, 7, 1), 200, np.uint8)], -1)
    grey = img[:, :, :1]
    for k, (px, typ, bpp, exp) in enumerate(((bgr, 2, 24, want), (bgra, 2, 32, want), (grey, 3, 8, np.repeat(want[:, :, :1], 3, -1)))):
        for top_down in (False, True):
            for rle in (False, True):
                p = str(tmp_path / f"t{k}_{int(top_down)}_{int(rle)}.tga")
                open(p, "wb").write(tga(px, typ + (8 if rle else 0), bpp, top_down, rle))
                np.testing.assert_array_equal(api.load_image(p), exp, err_msg=p)
    open(str(tmp_path / "bad.tga"), "wb").write(tga(bgr, 2, 24, False, False)[:40])
    with pytest.raises(api.RustlightError):
        api.load_image(str(tmp_path / "bad.tga"))
