"""Writers for the file formats the scene loaders read (OBJ + MTL, PLY, Mitsuba ``.serialized``, PNG, Mitsuba XML).

They exist so a ``SceneData`` can leave the process as files another renderer — rustlight itself — can open, and so
the loader tests can round-trip: fixture -> files -> ``rl_scene_load`` -> the same flattened scene.  Data only.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

from . import scenes as S


def _r(x) -> str:
    return repr(float(np.float32(x)))


# ------------------------------------------------------------------------------------------ meshes
def write_obj(meshes, path: str, with_mtl: bool = True) -> None:
    """One ``o`` block per mesh (single v/vt/vn index per corner), an MTL with ``Kd`` next to it."""
    base = os.path.splitext(os.path.basename(path))[0]
    lines, mtl = [f"mtllib {base}.mtl"] if with_mtl else [], []
    off = 1
    for m in meshes:
        lines.append(f"o {m.name}")
        if with_mtl:
            kd = m.bsdf.diffuse["color0"]
            mtl += [f"newmtl {m.name}", f"Kd {_r(kd[0])} {_r(kd[1])} {_r(kd[2])}", ""]
            lines.append(f"usemtl {m.name}")
        for v in m.vertices:
            lines.append(f"v {_r(v[0])} {_r(v[1])} {_r(v[2])}")
        if m.uv is not None:
            for t in m.uv:
                lines.append(f"vt {_r(t[0])} {_r(t[1])}")
        if m.normals is not None:
            for n in m.normals:
                lines.append(f"vn {_r(n[0])} {_r(n[1])} {_r(n[2])}")
        for tri in m.indices:
            def corner(i):
                i = int(i) + off
                if m.uv is not None and m.normals is not None:
                    return f"{i}/{i}/{i}"
                if m.uv is not None:
                    return f"{i}/{i}"
                if m.normals is not None:
                    return f"{i}//{i}"
                return f"{i}"
            lines.append("f " + " ".join(corner(i) for i in tri))
        off += m.vertices.shape[0]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    if with_mtl:
        with open(os.path.join(os.path.dirname(path) or ".", base + ".mtl"), "w") as f:
            f.write("\n".join(mtl) + "\n")


def write_ply(mesh, path: str, fmt: str = "binary_little_endian") -> None:
    """vertex x y z [nx ny nz] [u v], face ``list uchar int vertex_indices``; ascii / binary_little_endian / binary_big_endian."""
    n = mesh.vertices.shape[0]
    cols = [mesh.vertices]
    props = ["x", "y", "z"]
    if mesh.normals is not None:
        cols.append(mesh.normals); props += ["nx", "ny", "nz"]
    if mesh.uv is not None:
        cols.append(mesh.uv); props += ["u", "v"]
    table = np.concatenate(cols, axis=1).astype(np.float32)
    head = ["ply", f"format {fmt} 1.0", "comment rustlight_amd export", f"element vertex {n}"]
    head += [f"property float {p}" for p in props]
    head += [f"element face {mesh.indices.shape[0]}", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if fmt == "ascii":
            for row in table:
                f.write((" ".join(_r(x) for x in row) + "\n").encode())
            for tri in mesh.indices:
                f.write(("3 " + " ".join(str(int(i)) for i in tri) + "\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            f.write(table.astype(e + "f4").tobytes())
            for tri in mesh.indices:
                f.write(struct.pack(e + "B3i", 3, *[int(i) for i in tri]))


def write_serialized(meshes, path: str, version: int = 4, double: bool = False) -> None:
    """Mitsuba 0.5 ``.serialized``: per mesh [0x041C, version] + zlib(flags, name, counts, positions, normals, uvs, indices), offset table."""
    blobs, offsets = [], []
    pos = 0
    for m in meshes:
        flags = (0x2000 if double else 0x1000) | (0x0001 if m.normals is not None else 0) | (0x0002 if m.uv is not None else 0)
        ft = "<f8" if double else "<f4"
        body = struct.pack("<I", flags)
        if version == 4:
            body += m.name.encode() + b"\0"
        body += struct.pack("<QQ", m.vertices.shape[0], m.indices.shape[0])
        body += m.vertices.astype(ft).tobytes()
        if m.normals is not None:
            body += m.normals.astype(ft).tobytes()
        if m.uv is not None:
            body += m.uv.astype(ft).tobytes()
        body += m.indices.astype("<u4").tobytes()
        blob = struct.pack("<HH", 0x041C, version) + zlib.compress(body)
        offsets.append(pos)
        blobs.append(blob)
        pos += len(blob)
    with open(path, "wb") as f:
        for b in blobs:
            f.write(b)
        for o in offsets:
            f.write(struct.pack("<Q" if version == 4 else "<I", o))
        f.write(struct.pack("<I", len(blobs)))


def write_png(img_u8: np.ndarray, path: str) -> None:
    """8-bit RGB / RGBA / gray, non-interlaced, filter 0 on every row."""
    a = np.ascontiguousarray(img_u8, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = b"".join(b"\0" + a[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def write_exr(img: np.ndarray, path: str, compression: str = "zip", half: bool = False, data_origin=(0, 0), extra_channel: bool = False) -> None:
    """Single-part scanline OpenEXR (test fixture writer): R, G, B as FLOAT or HALF (plus an optional A channel the reader must
    skip), compression none | rle | zips | zip, dataWindow starting at `data_origin`."""
    a = np.ascontiguousarray(img, np.float32)
    h, w, _ = a.shape
    comp = {"none": 0, "rle": 1, "zips": 2, "zip": 3}[compression]
    names = ["A", "B", "G", "R"] if extra_channel else ["B", "G", "R"]
    src = {"R": 0, "G": 1, "B": 2}
    ptype = 1 if half else 2

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    chl = b"".join(n.encode() + b"\0" + struct.pack("<IBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = data_origin
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = (struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp]))
           + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
           + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
           + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")

    def rle(b):
        out = bytearray(); i = 0
        while i < len(b):
            j = i
            while j + 1 < len(b) and b[j + 1] == b[i] and j - i < 126: j += 1
            if j - i >= 2:
                out += bytes([j - i, b[i]]); i = j + 1
            else:
                k = i
                while k < len(b) and k - i < 127 and not (k + 2 < len(b) and b[k] == b[k + 1] == b[k + 2]): k += 1
                out += bytes([(-(k - i)) & 0xFF]) + b[i:k]; i = k
        return bytes(out)
    lines = 16 if comp == 3 else 1
    chunks = []
    for r0 in range(0, h, lines):
        raw = bytearray()
        for y in range(r0, min(h, r0 + lines)):
            for nme in names:
                plane = a[y, :, src[nme]] if nme in src else np.ones(w, np.float32)
                raw += (plane.astype(np.float16) if half else plane).tobytes()
        raw = bytes(raw)
        data = raw
        if comp:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]])                                   # interleave: even bytes first, then odd
            p = t.astype(np.int32)
            p[1:] = (p[1:] - p[:-1] + 128 + 256) & 0xFF                              # predictor
            enc = rle(bytes(p.astype(np.uint8))) if comp == 1 else zlib.compress(p.astype(np.uint8).tobytes())
            if len(enc) < len(raw): data = enc
        chunks.append(struct.pack("<iI", y0 + r0, len(data)) + data)
    off = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off); off += len(c)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(chunks))


# ------------------------------------------------------------------------------------------ Mitsuba XML
def _rgb(name, c):
    return f'<rgb name="{name}" value="{_r(c[0])}, {_r(c[1])}, {_r(c[2])}"/>'


def _color_xml(name, tex, tex_dir, counter):
    t = tex.get("type", S.TEX_CONSTANT)
    if t == S.TEX_CONSTANT:
        return _rgb(name, tex["color0"])
    if t == S.TEX_BITMAP:
        raise ValueError("bitmap textures are exported by the caller (needs the image file)")
    kind = "checkerboard" if t == S.TEX_CHECKERBOARD else "gridtexture"
    off, sc = tex.get("offset", (0.0, 0.0)), tex.get("scale", (1.0, 1.0))
    extra = f'<float name="lineWidth" value="{_r(tex.get("line_width", 0.01))}"/>' if t == S.TEX_GRID else ""
    return (f'<texture type="{kind}" name="{name}">{_rgb("color0", tex["color0"])}{_rgb("color1", tex["color1"])}'
            f'<float name="uoffset" value="{_r(off[0])}"/><float name="voffset" value="{_r(off[1])}"/>'
            f'<float name="uscale" value="{_r(sc[0])}"/><float name="vscale" value="{_r(sc[1])}"/>{extra}</texture>')


def bsdf_xml(b: S.Bsdf, ident: str) -> str:
    """The Mitsuba plugin + parameters that bsdf_mts maps back to this Bsdf (src/bsdfs/mod.rs:499-612)."""
    def dist():
        if b.distribution == S.MF_NONE:
            return ""
        return (f'<string name="distribution" value="{"ggx" if b.distribution == S.MF_GGX else "beckmann"}"/>'
                f'<float name="alpha" value="{_r(b.alpha_u)}"/>')
    rough = "rough" if b.distribution != S.MF_NONE else ""
    if b.type == S.DIFFUSE:
        body = ("diffuse", _color_xml("reflectance", b.diffuse, None, None))
    elif b.type == S.PHONG:
        body = ("phong", _color_xml("diffuseReflectance", b.diffuse, None, None) + _color_xml("specularReflectance", b.specular, None, None)
                + f'<float name="exponent" value="{_r(b.exponent)}"/>')
    elif b.type == S.GLASS:
        body = ("dielectric", _color_xml("specularReflectance", b.specular, None, None) + _color_xml("specularTransmittance", b.transmittance, None, None)
                + f'<float name="intIOR" value="{_r(b.glass_eta)}"/><float name="extIOR" value="1.0"/>')
    elif b.type == S.METAL:
        body = (rough + "conductor", _color_xml("specularReflectance", b.specular, None, None) + _rgb("eta", b.eta["color0"]) + _rgb("k", b.k["color0"])
                + '<float name="extEta" value="1.0"/>' + dist())
    else:
        body = (rough + "plastic", _color_xml("specularReflectance", b.specular, None, None) + _color_xml("diffuseReflectance", b.diffuse, None, None) + dist())
    return f'<bsdf type="{body[0]}" id="{ident}">{body[1]}</bsdf>'


def write_mitsuba(scene: S.SceneData, path: str, shape_format: str = "obj") -> None:
    """Mitsuba 0.5 XML + one mesh file per shape (obj | ply | serialized).  The sensor transform is written as a row-major
    <matrix>; the loader builds Camera::new(.., flip = true), so a round trip needs ``scene.flip == True``."""
    d = os.path.dirname(path) or "."
    base = os.path.splitext(os.path.basename(path))[0]
    tw = np.asarray(scene.to_world, np.float32).reshape(4, 4).T   # row-major
    out = ['<?xml version="1.0" encoding="utf-8"?>', "<!-- exported by rustlight_amd -->", '<scene version="0.5.0">',
           '<default name="spp" value="16"/>', '<integrator type="path"/>',
           '<sensor type="perspective">', f'<float name="fov" value="{_r(scene.fov)}"/>',
           f'<string name="fovAxis" value="{"y" if scene.fov_axis == 1 else "x"}"/>',
           '<transform name="toWorld"><matrix value="' + " ".join(_r(x) for x in tw.reshape(-1)) + '"/></transform>',
           '<sampler type="independent"><integer name="sampleCount" value="$spp"/></sampler>',
           f'<film type="hdrfilm"><integer name="width" value="{scene.width}"/><integer name="height" value="{scene.height}"/></film>', "</sensor>"]
    for i, m in enumerate(scene.meshes):
        out.append(bsdf_xml(m.bsdf, f"mat{i}"))
    if shape_format == "serialized":
        write_serialized(scene.meshes, os.path.join(d, base + ".serialized"))
    for i, m in enumerate(scene.meshes):
        if shape_format == "obj":
            fn = f"{base}_{i}.obj"
            write_obj([m], os.path.join(d, fn), with_mtl=False)
            shape = f'<shape type="obj"><string name="filename" value="{fn}"/>'
        elif shape_format == "ply":
            fn = f"{base}_{i}.ply"
            write_ply(m, os.path.join(d, fn), ["binary_little_endian", "ascii", "binary_big_endian"][i % 3])
            shape = f'<shape type="ply"><string name="filename" value="{fn}"/>'
        else:
            shape = f'<shape type="serialized"><string name="filename" value="{base}.serialized"/><integer name="shapeIndex" value="{i}"/>'
        shape += f'<ref id="mat{i}"/>'
        if m.emission is not None:
            shape += f'<emitter type="area">{_rgb("radiance", m.emission)}</emitter>'
        out.append(shape + "</shape>")
    for lt in scene.lights:
        if lt["type"] == "point":
            a = lt["a"]
            out.append(f'<emitter type="point"><point name="position" x="{_r(a[0])}" y="{_r(a[1])}" z="{_r(a[2])}"/>{_rgb("intensity", lt["intensity"])}</emitter>')
    if scene.medium is not None:
        md = scene.medium
        ph = '<phase type="isotropic"/>' if md.phase == S.PHASE_ISOTROPIC else f'<phase type="hg"><float name="g" value="{_r(md.g)}"/></phase>'
        out.append(f'<medium type="homogeneous" id="fog">{_rgb("sigmaS", md.sigma_s)}{_rgb("sigmaA", md.sigma_a)}<float name="scale" value="1"/>{ph}</medium>')
    out.append("</scene>")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
