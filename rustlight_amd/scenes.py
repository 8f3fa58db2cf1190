"""Synthetic scene fixtures (data only) shared by the product bindings, tests and bench.py.

A scene is a plain ``SceneData`` of numpy arrays mirroring what rustlight's loaders hand to
``Scene`` / ``Mesh::new`` (src/scene.rs:16-30, src/geometry.rs:122-182).  Nothing here computes
radiance; ``to_product`` / the oracle adapter push the arrays through the respective C-ABIs.

* ``cbox``           Bitterli's Cornell box exactly as embedded in the reference's web demo
                     (examples/web/index.html:9-43): 8 meshes, 36 triangles,
                     all ``matte``, one area light L=(17,12,4).
* ``living_room``    the "living-room-class" stand-in of SURVEY.md §8(d): cbox shell x4 + K
                     tessellated spheres with cycling materials + 2 emissive quads.
* ``furnace``        closed emissive box for the white-furnace invariant.
* ``single_triangle``known-answer ray/triangle fixture.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import os

import numpy as np

# rl_bsdf_type / rl_tex_type / rl_microfacet_type (include/rustlight_amd.h)
DIFFUSE, PHONG, METAL, GLASS, SUBSTRATE = 0, 1, 2, 3, 4
TEX_CONSTANT, TEX_CHECKERBOARD, TEX_GRID, TEX_BITMAP = 0, 1, 2, 3
MF_NONE, MF_BECKMANN, MF_GGX = 0, 1, 2
PHASE_ISOTROPIC, PHASE_HG = 0, 1


def const_color(rgb):
    return {"type": TEX_CONSTANT, "color0": tuple(float(c) for c in rgb)}


@dataclass
class Bsdf:
    type: int = DIFFUSE
    diffuse: dict = field(default_factory=lambda: const_color((0.5, 0.5, 0.5)))
    specular: dict = field(default_factory=lambda: const_color((1.0, 1.0, 1.0)))
    transmittance: dict = field(default_factory=lambda: const_color((1.0, 1.0, 1.0)))
    eta: dict = field(default_factory=lambda: const_color((0.2, 0.92, 1.1)))
    k: dict = field(default_factory=lambda: const_color((3.9, 2.45, 2.14)))
    exponent: float = 30.0
    weight_specular: float = 0.5
    distribution: int = MF_NONE
    alpha_u: float = 0.1
    alpha_v: float = 0.1
    glass_eta: float = 1.5046 / 1.000277  # bk7 / air, BSDFGlass::default (src/bsdfs/glass.rs:61-73)


def matte(rgb) -> Bsdf:
    """PBRT ``matte`` -> BSDFDiffuse (src/bsdfs/mod.rs:299-305)."""
    return Bsdf(type=DIFFUSE, diffuse=const_color(rgb))


@dataclass
class MeshData:
    name: str
    vertices: np.ndarray            # (n, 3) f32
    indices: np.ndarray             # (m, 3) u32
    normals: Optional[np.ndarray]   # (n, 3) f32 or None
    uv: Optional[np.ndarray]        # (n, 2) f32 or None
    bsdf: Bsdf
    emission: Optional[tuple] = None
    emission_kind: Optional[tuple] = None      # None: EmissionType::Color; ('hsv', scale) | ('texture', scale, bitmap_id) (geometry.rs:99-104)


@dataclass
class Medium:
    sigma_a: tuple
    sigma_s: tuple
    phase: int = PHASE_ISOTROPIC
    g: float = 0.0


@dataclass
class SceneData:
    width: int
    height: int
    fov: float
    fov_axis: int            # 0 = Fov::X, 1 = Fov::Y
    to_world: np.ndarray     # 16 f32, column-major camera-to-world
    flip: bool
    meshes: List[MeshData]
    medium: Optional[Medium] = None
    bitmaps: list = field(default_factory=list)
    lights: list = field(default_factory=list)           # [{"type": "point"|"directional", "a": position|direction, "intensity": rgb}]
    environment: Optional[tuple] = None                  # EnvironmentLightColor::Constant(rgb)
    environment_map: Optional[np.ndarray] = None         # EnvironmentLightColor::Texture: H x W x 3 lat-long image (z up)
    use_ats: bool = False                                # Scene::build_emitters(build_ats): the `-x ats` light tree

    @property
    def n_triangles(self) -> int:
        return int(sum(m.indices.shape[0] for m in self.meshes))


def _f32(a, cols):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, cols))


def _quad_mesh(name, P, N, bsdf, emission=None):
    uv = [0, 0, 1, 0, 1, 1, 0, 1]
    return MeshData(name, _f32(P, 3), np.asarray([[0, 1, 2], [0, 2, 3]], dtype=np.uint32),
                    _f32(N * 4 if len(N) == 3 else N, 3), _f32(uv, 2), bsdf, emission)


# ------------------------------------------------------------------------------------------
# Cornell box — the literal numbers of examples/web/index.html:24-41 (scene data, not code)
_CBOX_SHORT_P = [
    -0.0460751, 0.6, 0.573007, -0.0460751, -2.98023e-008, 0.573007, 0.124253, 0, 0.00310463, 0.124253, 0.6, 0.00310463,
    0.533009, 0, 0.746079, 0.533009, 0.6, 0.746079, 0.703337, 0.6, 0.176177, 0.703337, 2.98023e-008, 0.176177,
    0.533009, 0.6, 0.746079, -0.0460751, 0.6, 0.573007, 0.124253, 0.6, 0.00310463, 0.703337, 0.6, 0.176177,
    0.703337, 2.98023e-008, 0.176177, 0.124253, 0, 0.00310463, -0.0460751, -2.98023e-008, 0.573007, 0.533009, 0, 0.746079,
    0.533009, 0, 0.746079, -0.0460751, -2.98023e-008, 0.573007, -0.0460751, 0.6, 0.573007, 0.533009, 0.6, 0.746079,
    0.703337, 0.6, 0.176177, 0.124253, 0.6, 0.00310463, 0.124253, 0, 0.00310463, 0.703337, 2.98023e-008, 0.176177]
_CBOX_SHORT_N = (
    [-0.958123, -4.18809e-008, -0.286357] * 4 + [0.958123, 4.18809e-008, 0.286357] * 4 +
    [-4.37114e-008, 1, -1.91069e-015] * 4 + [4.37114e-008, -1, 1.91069e-015] * 4 +
    [-0.286357, -1.25171e-008, 0.958123] * 4 + [0.286357, 1.25171e-008, -0.958123] * 4)
_CBOX_TALL_P = [
    -0.720444, 1.2, -0.473882, -0.720444, 0, -0.473882, -0.146892, 0, -0.673479, -0.146892, 1.2, -0.673479,
    -0.523986, 0, 0.0906493, -0.523986, 1.2, 0.0906492, 0.0495656, 1.2, -0.108948, 0.0495656, 0, -0.108948,
    -0.523986, 1.2, 0.0906492, -0.720444, 1.2, -0.473882, -0.146892, 1.2, -0.673479, 0.0495656, 1.2, -0.108948,
    0.0495656, 0, -0.108948, -0.146892, 0, -0.673479, -0.720444, 0, -0.473882, -0.523986, 0, 0.0906493,
    -0.523986, 0, 0.0906493, -0.720444, 0, -0.473882, -0.720444, 1.2, -0.473882, -0.523986, 1.2, 0.0906492,
    0.0495656, 1.2, -0.108948, -0.146892, 1.2, -0.673479, -0.146892, 0, -0.673479, 0.0495656, 0, -0.108948]
_CBOX_TALL_N = (
    [-0.328669, -4.1283e-008, -0.944445] * 4 + [0.328669, 4.1283e-008, 0.944445] * 4 +
    [3.82137e-015, 1, -4.37114e-008] * 4 + [-3.82137e-015, -1, 4.37114e-008] * 4 +
    [-0.944445, 1.43666e-008, 0.328669] * 4 + [0.944445, -1.43666e-008, -0.328669] * 4)
_BOX_IDX = [0, 2, 1, 0, 3, 2, 4, 6, 5, 4, 7, 6, 8, 10, 9, 8, 11, 10, 12, 14, 13, 12, 15, 14,
            16, 18, 17, 16, 19, 18, 20, 22, 21, 20, 23, 22]
_BOX_UV = [0, 0, 1, 0, 1, 1, 0, 1] * 6

# camera: `Transform [1 0 0 0  0 1 0 0  0 0 -1 0  0 -1 6.8 1]` is world->camera (index.html:10);
# the PBRT loader inverts it (src/scene_loader.rs:288).  The inverse, column-major:
CBOX_WORLD_TO_CAMERA = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, -1, 6.8, 1]
CBOX_TO_WORLD = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 1, 6.8, 1]
CBOX_FOV = 19.5


def cbox_meshes() -> List[MeshData]:
    white = (0.725, 0.71, 0.68)
    meshes = [
        _quad_mesh("Floor", [-1, 1.74846e-007, -1, -1, 1.74846e-007, 1, 1, -1.74846e-007, 1, 1, -1.74846e-007, -1],
                   [4.37114e-008, 1, 1.91069e-015], matte(white)),
        _quad_mesh("Ceiling", [1, 2, 1, -1, 2, 1, -1, 2, -1, 1, 2, -1],
                   [-8.74228e-008, -1, -4.37114e-008], matte(white)),
        _quad_mesh("BackWall", [-1, 0, -1, -1, 2, -1, 1, 2, -1, 1, 0, -1],
                   [8.74228e-008, -4.37114e-008, -1], matte(white)),
        _quad_mesh("RightWall", [1, 0, -1, 1, 2, -1, 1, 2, 1, 1, 0, 1],
                   [1, -4.37114e-008, 1.31134e-007], matte((0.14, 0.45, 0.091))),
        _quad_mesh("LeftWall", [-1, 0, 1, -1, 2, 1, -1, 2, -1, -1, 0, -1],
                   [-1, -4.37114e-008, -4.37114e-008], matte((0.63, 0.065, 0.05))),
        MeshData("ShortBox", _f32(_CBOX_SHORT_P, 3), np.asarray(_BOX_IDX, dtype=np.uint32).reshape(-1, 3),
                 _f32(_CBOX_SHORT_N, 3), _f32(_BOX_UV, 2), matte(white)),
        MeshData("TallBox", _f32(_CBOX_TALL_P, 3), np.asarray(_BOX_IDX, dtype=np.uint32).reshape(-1, 3),
                 _f32(_CBOX_TALL_N, 3), _f32(_BOX_UV, 2), matte(white)),
        _quad_mesh("Light", [-0.24, 1.98, -0.22, 0.23, 1.98, -0.22, 0.23, 1.98, 0.16, -0.24, 1.98, 0.16],
                   [-8.74228e-008, -1, 1.86006e-007], matte((0.0, 0.0, 0.0)), emission=(17.0, 12.0, 4.0)),
    ]
    return meshes


def cbox(width: int = 256, height: int = 256, medium: Optional[Medium] = None) -> SceneData:
    """Cornell box; image size overridden as BASELINE.json's configs do (256x256 / 1920x1080)."""
    return SceneData(width, height, CBOX_FOV, 1, np.asarray(CBOX_TO_WORLD, dtype=np.float32), False,
                     cbox_meshes(), medium)


def cbox_medium(width: int = 256, height: int = 256, sigma_s: float = 0.5, sigma_a: float = 0.0,
                g: Optional[float] = None) -> SceneData:
    """cfg 5: ``-m 0.5`` => sigma_s = 0.5, sigma_a = 0, isotropic (examples/cli.rs:355-399)."""
    med = Medium((sigma_a,) * 3, (sigma_s,) * 3, PHASE_ISOTROPIC if g is None else PHASE_HG, 0.0 if g is None else g)
    return cbox(width, height, med)


# ------------------------------------------------------------------------------------------
def single_triangle() -> SceneData:
    v = _f32([0, 0, 0, 1, 0, 0, 0, 1, 0], 3)
    m = MeshData("tri", v, np.asarray([[0, 1, 2]], dtype=np.uint32), None, None, matte((0.5, 0.5, 0.5)))
    light = _quad_mesh("light", [-1, -1, 5, 1, -1, 5, 1, 1, 5, -1, 1, 5], [0, 0, -1], matte((0, 0, 0)), emission=(1, 1, 1))
    to_world = np.asarray([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.25, 0.25, -3, 1], dtype=np.float32)
    return SceneData(32, 32, 40.0, 0, to_world, False, [m, light])


def cbox_other_lights(width: int = 64, height: int = 64, point=True, directional=True, environment=True, keep_area_light=True) -> SceneData:
    """Cornell box lit by the non-mesh emitters of SURVEY.md a24 (point, directional, constant environment)."""
    sd = cbox(width, height)
    if not keep_area_light:
        sd.meshes[-1].emission = None
    if point:
        sd.lights.append({"type": "point", "a": (0.3, 1.5, 0.4), "intensity": (2.0, 1.5, 1.0)})
    if directional:
        d = np.asarray((0.3, -0.8, -0.52), dtype=np.float64)
        d = d / np.linalg.norm(d)
        sd.lights.append({"type": "directional", "a": tuple(float(np.float32(x)) for x in d), "intensity": (1.0, 1.0, 1.2)})
    if environment:
        sd.environment = (0.3, 0.4, 0.6)
    return sd


def sky_map(w: int = 32, h: int = 16) -> np.ndarray:
    """A small procedural lat-long environment (H x W x 3, row 0 = +z): blue-ish gradient, a bright 2 x 2 'sun',
    one exactly black row and a black column so the zero-probability bins of the Distribution2D are exercised."""
    img = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        t = np.float32(y) / np.float32(h - 1)
        img[y, :, 0] = np.float32(0.15) + np.float32(0.25) * t
        img[y, :, 1] = np.float32(0.25) + np.float32(0.2) * t
        img[y, :, 2] = np.float32(0.6) - np.float32(0.3) * t
    img[h // 4: h // 4 + 2, w // 3: w // 3 + 2] = (40.0, 36.0, 30.0)
    img[h - 3, :] = 0.0
    img[:, w - 5] = 0.0
    return img


def sky_scene(width: int = 64, height: int = 64, keep_area_light: bool = False) -> SceneData:
    """The Cornell floor and boxes under a textured environment (EnvironmentLightColor::Texture).
    The lat-long parameterisation is z-up while the fixture is y-up; the reference applies no transform, neither do we."""
    sd = cbox(width, height)
    keep = {"Floor", "ShortBox", "TallBox"} | ({"Light"} if keep_area_light else set())
    sd.meshes = [m for m in sd.meshes if m.name in keep]
    sd.environment_map = sky_map()
    return sd


def many_lights(width: int = 64, height: int = 64, n: int = 5, use_ats: bool = True, glowing_spheres: int = 0) -> SceneData:
    """Cornell box whose single light is replaced by an n x n grid of small emissive quads under the ceiling, with emission
    varying over two decades and alternating tilt — the many-light case the `-x ats` light tree is for."""
    sd = cbox(width, height)
    sd.meshes = [m for m in sd.meshes if m.name != "Light"]
    k = 0
    for i in range(n):
        for j in range(n):
            cx = -0.8 + 1.6 * (i + 0.5) / n
            cz = -0.8 + 1.6 * (j + 0.5) / n
            h = 0.06
            tilt = 0.05 * ((i + 2 * j) % 3 - 1)
            e = 0.5 * (1.0 + ((7 * i + 3 * j) % 11)) ** 2 / 4.0
            P = [cx - h, 1.95 + tilt, cz - h, cx + h, 1.95 - tilt, cz - h, cx + h, 1.95 - tilt, cz + h, cx - h, 1.95 + tilt, cz + h]
            sd.meshes.append(_quad_mesh(f"L{k}", P, [0, -1, 0], matte((0.0, 0.0, 0.0)), emission=(e, 0.8 * e, 0.5 * e)))
            k += 1
    for q in range(glowing_spheres):        # emissive tessellated spheres: light-triangle normals in every direction (cone unions, rotations)
        v, idx, nrm, uv = uv_sphere((-0.55 + 0.55 * q, 0.35 + 0.3 * q, 0.45 - 0.25 * q), 0.12 + 0.03 * q, 6, 8)
        e = 3.0 + 2.0 * q
        sd.meshes.append(MeshData(f"Glow{q}", v, idx, nrm, uv, matte((0.0, 0.0, 0.0)), emission=(e, e, 0.7 * e)))
    sd.use_ats = use_ats
    return sd


def furnace(albedo: float = 0.5, le: float = 1.0, width: int = 32, height: int = 32) -> SceneData:
    """Closed cube whose six walls all emit ``le`` and reflect ``albedo`` (inward normals).
    Every pixel converges to le / (1 - albedo)."""
    meshes = []
    faces = [
        ("floor", [-1, -1, -1, -1, -1, 1, 1, -1, 1, 1, -1, -1], [0, 1, 0]),
        ("ceil", [1, 1, 1, -1, 1, 1, -1, 1, -1, 1, 1, -1], [0, -1, 0]),
        ("back", [-1, -1, -1, -1, 1, -1, 1, 1, -1, 1, -1, -1], [0, 0, 1]),
        ("front", [-1, -1, 1, 1, -1, 1, 1, 1, 1, -1, 1, 1], [0, 0, -1]),
        ("right", [1, -1, -1, 1, 1, -1, 1, 1, 1, 1, -1, 1], [-1, 0, 0]),
        ("left", [-1, -1, 1, -1, 1, 1, -1, 1, -1, -1, -1, -1], [1, 0, 0]),
    ]
    for name, P, N in faces:
        meshes.append(_quad_mesh(name, P, N, matte((albedo,) * 3), emission=(le,) * 3))
    to_world = np.asarray([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0.5, 1], dtype=np.float32)
    return SceneData(width, height, 60.0, 0, to_world, False, meshes)


# ------------------------------------------------------------------------------------------
class _Xoshiro:
    """Xoshiro256++ (same generator family as the renderer; used here only to place objects)."""

    def __init__(self, seed: int):
        self.s = []
        x = seed & 0xFFFFFFFFFFFFFFFF
        for _ in range(4):  # SplitMix64
            x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            self.s.append(z ^ (z >> 31))

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & 0xFFFFFFFFFFFFFFFF

    def next_u64(self):
        s = self.s
        r = (self._rotl((s[0] + s[3]) & 0xFFFFFFFFFFFFFFFF, 23) + s[0]) & 0xFFFFFFFFFFFFFFFF
        t = (s[1] << 17) & 0xFFFFFFFFFFFFFFFF
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return r

    def uniform(self):
        return (self.next_u64() >> 40) * (1.0 / 16777216.0)


def uv_sphere(center, radius, n_theta: int = 32, n_phi: int = 32):
    """Tessellated sphere in the spirit of the Mitsuba loader's sphere (src/scene_loader.rs:598-629)."""
    verts, normals, uvs = [], [], []
    for i in range(n_theta + 1):
        th = math.pi * i / n_theta
        for j in range(n_phi + 1):
            ph = 2.0 * math.pi * j / n_phi
            n = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
            normals.append(n)
            verts.append((center[0] + radius * n[0], center[1] + radius * n[1], center[2] + radius * n[2]))
            uvs.append((j / n_phi, i / n_theta))
    idx = []
    for i in range(n_theta):
        for j in range(n_phi):
            a = i * (n_phi + 1) + j
            b = a + n_phi + 1
            if i != 0:
                idx.append((a, a + 1, b))
            if i != n_theta - 1:
                idx.append((a + 1, b + 1, b))
    return _f32(verts, 3), np.asarray(idx, dtype=np.uint32), _f32(normals, 3), _f32(uvs, 2)


def living_room_materials() -> List[Bsdf]:
    """Material cycle of SURVEY.md §8(d): Diffuse / Phong / Metal specular / Metal GGX / Glass / Substrate GGX."""
    return [
        matte((0.6, 0.55, 0.5)),
        Bsdf(type=PHONG, diffuse=const_color((0.4, 0.4, 0.45)), specular=const_color((0.3, 0.3, 0.3)),
             exponent=50.0, weight_specular=0.3),
        Bsdf(type=METAL, specular=const_color((1, 1, 1)), distribution=MF_NONE),
        Bsdf(type=METAL, specular=const_color((1, 1, 1)), distribution=MF_GGX, alpha_u=0.1, alpha_v=0.1),
        Bsdf(type=GLASS),
        Bsdf(type=SUBSTRATE, diffuse=const_color((0.5, 0.3, 0.2)), specular=const_color((0.05, 0.05, 0.05)),
             distribution=MF_GGX, alpha_u=0.05, alpha_v=0.05),
    ]


def living_room(width: int = 1920, height: int = 1080, n_spheres: int = 256, tess: int = 32,
                seed: int = 1234) -> SceneData:
    """Living-room-class synthetic scene (the real PBRT living-room is not available offline):
    the Cornell shell scaled x4, ``n_spheres`` tessellated spheres on a jittered grid with cycling
    materials, and two emissive quads.  ``tess`` scales the triangle count (2*tess*(tess-1) per sphere)."""
    s = 4.0
    meshes = []
    for m in cbox_meshes()[:5]:
        meshes.append(MeshData(m.name, (m.vertices * s).astype(np.float32), m.indices, m.normals, m.uv, m.bsdf))
    rng = _Xoshiro(seed)
    mats = living_room_materials()
    g = max(1, int(math.ceil(n_spheres ** (1.0 / 3.0))))
    cell = (2.0 * s * 0.8) / g
    k = 0
    for iz in range(g):
        for iy in range(g):
            for ix in range(g):
                if k >= n_spheres:
                    break
                jx, jy, jz = rng.uniform(), rng.uniform(), rng.uniform()
                r = cell * (0.18 + 0.17 * rng.uniform())
                c = (-s * 0.8 + (ix + 0.25 + 0.5 * jx) * cell,
                     0.3 + (iy + 0.25 + 0.5 * jy) * (2.0 * s * 0.8 - 0.6) / g,
                     -s * 0.8 + (iz + 0.25 + 0.5 * jz) * cell)
                v, i, n, uv = uv_sphere(c, r, tess, tess)
                meshes.append(MeshData(f"sphere{k}", v, i, n, uv, mats[k % len(mats)]))
                k += 1
    y = 2.0 * s - 0.02
    for name, x0 in (("LightA", -2.4), ("LightB", 1.2)):
        P = [x0, y, -0.6, x0 + 1.2, y, -0.6, x0 + 1.2, y, 0.6, x0, y, 0.6]
        meshes.append(_quad_mesh(name, P, [0, -1, 0], matte((0, 0, 0)), emission=(17.0, 12.0, 4.0)))
    to_world = np.asarray([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, s, 6.8 * s, 1], dtype=np.float32)
    return SceneData(width, height, CBOX_FOV, 1, to_world, False, meshes)


def write_pbrt(scene: SceneData, path: str) -> None:
    """Emit the scene as the PBRT subset rustlight's loader consumes (matte materials only)."""
    tw = np.asarray(scene.to_world, dtype=np.float64).reshape(4, 4).T  # row-major math matrix
    w2c = np.linalg.inv(tw).T.reshape(-1)  # back to column-major
    lines = ["Transform [ " + " ".join(repr(float(np.float32(x))) for x in w2c) + " ]",
             f'Film "image" "integer xresolution" [ {scene.width} ] "integer yresolution" [ {scene.height} ]',
             f'Camera "perspective" "float fov" [ {scene.fov} ]', "WorldBegin"]
    for m in scene.meshes:
        if m.bsdf.type != DIFFUSE:
            raise ValueError("write_pbrt only emits matte materials")
        kd = m.bsdf.diffuse["color0"]
        lines.append(f'MakeNamedMaterial "{m.name}" "string type" [ "matte" ] "rgb Kd" [ {kd[0]!r} {kd[1]!r} {kd[2]!r} ]')
    for m in scene.meshes:
        def fmt(a):
            return " ".join(repr(float(x)) for x in np.asarray(a).reshape(-1))
        shape = (f'Shape "trianglemesh" "integer indices" [ {" ".join(str(int(i)) for i in m.indices.reshape(-1))} ] '
                 f'"point P" [ {fmt(m.vertices)} ]')
        if m.normals is not None:
            shape += f' "normal N" [ {fmt(m.normals)} ]'
        if m.uv is not None:
            shape += f' "float uv" [ {fmt(m.uv)} ]'
        if m.emission is not None:
            lines += ["AttributeBegin",
                      f'AreaLightSource "diffuse" "rgb L" [ {m.emission[0]!r} {m.emission[1]!r} {m.emission[2]!r} ]',
                      f'NamedMaterial "{m.name}"', shape, "AttributeEnd"]
        else:
            lines += [f'NamedMaterial "{m.name}"', shape]
    for lt in scene.lights:
        i = lt["intensity"]
        if lt["type"] == "point":
            lines.append(f'LightSource "point" "rgb I" [ {i[0]!r} {i[1]!r} {i[2]!r} ] "point from" [ {lt["a"][0]!r} {lt["a"][1]!r} {lt["a"][2]!r} ]')
        else:   # DirectionalLight.direction = (to - from).normalize(): write the direction itself as `to`
            lines.append(f'LightSource "distant" "rgb L" [ {i[0]!r} {i[1]!r} {i[2]!r} ] "point from" [ 0 0 0 ] "point to" [ {lt["a"][0]!r} {lt["a"][1]!r} {lt["a"][2]!r} ]')
    if scene.environment is not None:
        e = scene.environment
        lines.append(f'LightSource "infinite" "rgb L" [ {e[0]!r} {e[1]!r} {e[2]!r} ]')
    if scene.environment_map is not None:   # Spectrum::Mapname: a PFM next to the scene file (rows bottom-up, little endian)
        em = np.ascontiguousarray(scene.environment_map, np.float32)
        name = os.path.splitext(os.path.basename(path))[0] + "_env.pfm"
        with open(os.path.join(os.path.dirname(path) or ".", name), "wb") as f:
            f.write(f"PF\n{em.shape[1]} {em.shape[0]}\n-1.0\n".encode())
            f.write(em[::-1].astype("<f4").tobytes())
        lines.append(f'LightSource "infinite" "string mapname" [ "{name}" ]')
    lines.append("WorldEnd")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def luminance(rgb) -> np.float32:
    """Color::luminance (src/structure.rs:173-176), in f32 like the reference."""
    r, g, b = (np.float32(x) for x in rgb)
    return np.float32(np.float32(r * np.float32(0.212671) + g * np.float32(0.715160)) + b * np.float32(0.072169))


def override_light_emission(sd: "SceneData", kind: str, bitmap_id: int = -1) -> "SceneData":
    """examples/cli.rs:410-429 (`-x hvs-light` / `-x texture-light`): every light mesh becomes EmissionType::HSV / Texture with scale = its colour's luminance."""
    for m in sd.meshes:
        if m.emission is not None:
            scale = float(luminance(m.emission)) if m.emission_kind is None else 1.0
            m.emission_kind = ("hsv", scale) if kind == "hsv" else ("texture", scale, bitmap_id)
    return sd
