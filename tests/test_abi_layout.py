"""One ABI, checked three ways.  The POD structs of the C-ABI exist in three hand-maintained copies: `include/rustlight_amd.h` (the
boundary), the ctypes mirror `rustlight_amd/abi.py` (tests, bench.py) and the `#[repr(C)]` block of INTEGRATION.md (what a rustlight
maintainer pastes into `src/integrators/amd_ffi.rs`).  This test
  1. parses every `typedef struct` of the header and generates a C99 program that prints sizeof / offsetof of every field
     (compiled with `gcc -std=c99 -pedantic`: the header itself must be valid C, not only C++),
  2. compares that ground truth with the ctypes layouts field by field (name, offset, size),
  3. parses the Rust structs out of INTEGRATION.md, lays them out by the `repr(C)` rules and compares names, order, offsets and sizes.
A field added, renamed, reordered or retyped in one copy only fails here, on CPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from rustlight_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rustlight_amd.h")

CTYPES_OF = {"rl_color_desc": abi.ColorDesc, "rl_bsdf_desc": abi.BsdfDesc, "rl_mesh_desc": abi.MeshDesc, "rl_bitmap_desc": abi.BitmapDesc,
             "rl_light_desc": abi.LightDesc, "rl_scene_desc": abi.SceneDesc, "rl_sampler": abi.Sampler, "rl_path_params": abi.PathParams,
             "rl_mc_params": abi.McParams, "rl_render_stats": abi.RenderStats}
RUST_NAME = {"type": "ty"}      # `type` is a Rust keyword


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def header_structs():
    """{struct name: [field names in declaration order]} for every `typedef struct NAME { ... } NAME;` of the header."""
    text = _strip_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        assert m.group(3) == name
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # "const float* vertices", "uint32_t width, height", "float sigma_a[3], sigma_s[3]", "rl_bsdf_desc bsdf"
            first, *rest = [d.strip() for d in decl.split(",")]
            fields.append(re.search(r"(\w+)\s*(\[\d+\])?$", first).group(1))
            for r in rest:
                fields.append(re.search(r"(\w+)\s*(\[\d+\])?$", r).group(1))
        out[name] = fields
    return out


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    """Ground truth: {struct: (sizeof, [(field, offset, size), ...])} printed by a C99 program generated from the header."""
    structs = header_structs()
    assert set(structs) == set(CTYPES_OF), (sorted(structs), sorted(CTYPES_OF))
    d = tmp_path_factory.mktemp("abi")
    src = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{HEADER}"', "int main(void) {"]
    for name, fields in structs.items():
        src.append(f'    printf("S {name} %zu\\n", sizeof({name}));')
        for f in fields:
            src.append(f'    printf("F {name} {f} %zu %zu\\n", offsetof({name}, {f}), sizeof((({name}*)0)->{f}));')
    src += ["    return 0;", "}"]
    c_file, exe = d / "abi_layout.c", d / "abi_layout"
    c_file.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", str(c_file), "-o", str(exe)])
    out = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        p = line.split()
        if p[0] == "S":
            out[p[1]] = (int(p[2]), [])
        else:
            out[p[1]][1].append((p[2], int(p[3]), int(p[4])))
    return out


def test_header_is_c99_and_every_struct_is_mirrored(c_layout):
    assert len(c_layout) == 10 and all(size > 0 and fields for size, fields in c_layout.values())


def test_ctypes_mirror_matches_the_header(c_layout):
    for name, (size, fields) in c_layout.items():
        cls = CTYPES_OF[name]
        assert C.sizeof(cls) == size, (name, C.sizeof(cls), size)
        mine = [(f[0], getattr(cls, f[0]).offset, getattr(cls, f[0]).size) for f in cls._fields_]
        assert mine == fields, (name, [a for a, b in zip(mine, fields) if a != b], len(mine), len(fields))


# ---- the Rust side: #[repr(C)] layout rules over the structs written in INTEGRATION.md
_PRIM = {"u8": 1, "i8": 1, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8, "c_int": 4}


def rust_structs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = "\n".join(re.findall(r"```rust\n(.*?)```", text, flags=re.S))
    code = re.sub(r"/\*.*?\*/", " ", code, flags=re.S)
    code = re.sub(r"//[^\n]*", " ", code)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{;]*?pub\s+struct\s+(\w+)\s*\{(.*?)\}", code, flags=re.S):
        fields = re.findall(r"pub\s+(\w+)\s*:\s*([^,}]+?)\s*(?:,|$)", m.group(2).strip() + ",", flags=re.S)
        out[m.group(1)] = [(n, t.strip()) for n, t in fields]
    return out


def _rust_layout(ty, structs):
    """(size, align) of a Rust type under repr(C) on x86-64."""
    ty = ty.strip()
    if ty in _PRIM:
        return _PRIM[ty], _PRIM[ty]
    if ty.startswith("*"):
        return 8, 8
    m = re.fullmatch(r"\[\s*(.+?)\s*;\s*(\d+)\s*\]", ty)
    if m:
        s, a = _rust_layout(m.group(1), structs)
        return s * int(m.group(2)), a
    off, align = 0, 1
    for _, t in structs[ty]:
        s, a = _rust_layout(t, structs)
        off = (off + a - 1) // a * a + s
        align = max(align, a)
    return (off + align - 1) // align * align, align


def test_rust_repr_c_block_matches_the_header(c_layout):
    structs = rust_structs()
    missing = [n for n in c_layout if n not in structs]
    assert not missing, f"INTEGRATION.md has no #[repr(C)] struct for {missing}"
    for name, (size, fields) in c_layout.items():
        assert [RUST_NAME.get(f[0], f[0]) for f in fields] == [n for n, _ in structs[name]], name
        off, align, got = 0, 1, []
        for n, t in structs[name]:
            s, a = _rust_layout(t, structs)
            off = (off + a - 1) // a * a
            got.append((n, off, s))
            off += s
            align = max(align, a)
        assert got == [(RUST_NAME.get(f, f), o, s) for f, o, s in fields], (name, got, fields)
        assert (off + align - 1) // align * align == size, name
    # the opaque handles are zero-sized on the Rust side
    for opaque in ("rl_scene", "rl_context", "rl_multi"):
        assert structs[opaque] == [] or structs[opaque][0][1].startswith("[u8; 0]"), opaque


def test_option_table_is_consistent():
    """kernels/knobs.h: the enum of execution options and the table of their names (rl_context_set_option looks a name up by its position) must have the same length and order
    of magnitude of entries, every name must be lower-case `[a-z_]+`, unique, and the ones the header documents must exist."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rustlight_amd", "csrc", "kernels", "knobs.h")).read()
    body = src[src.index("enum KnobId {"):src.index("K_COUNT")]
    body = re.sub(r"//[^\n]*", "", body)
    ids = re.findall(r"\bK_[A-Z0-9_]+\b", body)
    table = src[src.index("names[K_COUNT] = {"):src.index("};", src.index("names[K_COUNT] = {"))]
    names = re.findall(r'"([^"]+)"', table)
    assert len(ids) == len(names) == len(set(names)) == len(set(ids)), (len(ids), len(names))
    assert all(re.fullmatch(r"[a-z][a-z0-9_]*", n) for n in names)
    assert [i[2:].lower() for i in ids] == names          # K_SPEC_FORCE <-> "spec_force": same order, same spelling
    header = open(os.path.join(root, "include", "rustlight_amd.h")).read()
    doc = header[header.index("Execution options."):header.index("#ifndef RUSTLIGHT_AMD_H")]
    documented = set(re.findall(r"\*\s+([a-z][a-z0-9_]*) = ", doc)) | set(re.findall(r", ([a-z][a-z0-9_]*) = 1", doc))
    assert documented and documented <= set(names), documented - set(names)
