"""ctypes binding of the product C-ABI (include/rustlight_amd.h) + the Python mirror of rustlight's
`Integrator` surface for the `path` hot path.

The library must already be built in-tree (``python -m rustlight_amd.build`` or
``__graft_entry__.build()``); there is no CPU fallback — on a machine without a GPU
``Context`` raises ``NoDeviceError`` (RL_ERR_NO_DEVICE).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Optional

import numpy as np

from . import abi
from . import scenes as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librustlight_amd.so")

RL_OK = 0
RL_ERR_NO_DEVICE = -2
STRATEGY_ALL, STRATEGY_BSDF, STRATEGY_EMITTER = 0, 1, 2
STREAM_REFERENCE_ORDER, STREAM_PER_SAMPLE = 0, 1
PIPELINE_AUTO, PIPELINE_WAVEFRONT, PIPELINE_FUSED = 0, 1, 2
NUMERICS_EXACT, NUMERICS_FAST = 0, 1

# every symbol include/rustlight_amd.h declares (tests check the .so exports all of them)
PUBLIC_SYMBOLS = [
    "rl_scene_create", "rl_scene_create_from_desc", "rl_scene_destroy", "rl_scene_set_camera", "rl_scene_set_camera_matrices", "rl_scene_get_camera_matrices", "rl_scene_set_mesh_emission", "rl_scene_override_light_emission", "rl_scene_scale_image", "rl_scene_add_mesh",
    "rl_scene_add_bitmap", "rl_scene_set_medium", "rl_scene_add_point_light", "rl_scene_add_directional_light",
    "rl_scene_set_environment", "rl_scene_set_environment_map", "rl_scene_build_emitters", "rl_scene_enable_ats", "rl_scene_load_pbrt", "rl_scene_load_mitsuba", "rl_scene_load",
    "rl_scene_image_size", "rl_scene_counts", "rl_sampler_seed", "rl_sampler_next_u64", "rl_sampler_next_f32",
    "rl_path_params_default", "rl_device_count", "rl_context_create", "rl_context_destroy", "rl_context_set_option", "rl_context_get_option", "rl_last_error", "rl_block_count",
    "rl_generate_block_seeds", "rl_render_path", "rl_render_path_frames", "rl_multi_create", "rl_multi_destroy", "rl_multi_info", "rl_multi_describe", "rl_multi_shard_stats", "rl_multi_render_path", "rl_render_ao", "rl_render_direct", "rl_trace_batch", "rl_visible_batch", "rl_load_pfm", "rl_load_image", "rl_save_pfm", "rl_save_png", "rl_save_exr", "rl_save_image", "rl_build_info",
]


class RustlightError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"rustlight_amd error {code}: {msg}")
        self.code = code


class NoDeviceError(RustlightError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m rustlight_amd.build` "
                           "(the HIP extension is mandatory, there is no fallback path)")
    L = C.CDLL(LIB_PATH)
    vp, f32p, u32p, u64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.rl_context_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.rl_context_get_option.argtypes = [vp, C.c_char_p]
    L.rl_context_get_option.restype = C.c_char_p
    L.rl_scene_create.argtypes = [C.POINTER(vp)]
    L.rl_scene_create_from_desc.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(vp)]
    L.rl_scene_destroy.argtypes = [vp]
    L.rl_scene_destroy.restype = None
    L.rl_scene_set_camera.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_float, C.c_int, f32p, C.c_int]
    L.rl_scene_set_camera_matrices.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, f32p]
    L.rl_scene_get_camera_matrices.argtypes = [vp, f32p, f32p, f32p]
    L.rl_scene_set_mesh_emission.argtypes = [vp, C.c_uint32, C.c_int, C.c_float, C.c_int]
    L.rl_scene_override_light_emission.argtypes = [vp, C.c_int, C.c_int]
    L.rl_scene_scale_image.argtypes = [vp, C.c_float]
    L.rl_scene_add_mesh.argtypes = [vp, f32p, C.c_size_t, u32p, C.c_size_t, f32p, f32p, C.POINTER(abi.BsdfDesc), f32p]
    L.rl_scene_add_bitmap.argtypes = [vp, C.c_uint32, C.c_uint32, f32p]
    L.rl_scene_set_medium.argtypes = [vp, f32p, f32p, C.c_int, C.c_float]
    L.rl_scene_build_emitters.argtypes = [vp]
    L.rl_scene_add_point_light.argtypes = [vp, f32p, f32p]
    L.rl_scene_add_directional_light.argtypes = [vp, f32p, f32p]
    L.rl_scene_set_environment.argtypes = [vp, f32p]
    L.rl_scene_set_environment_map.argtypes = [vp, C.c_uint32, C.c_uint32, f32p]
    L.rl_scene_load_pbrt.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rl_scene_load_mitsuba.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rl_scene_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rl_scene_image_size.argtypes = [vp, u32p, u32p]
    L.rl_scene_counts.argtypes = [vp, u64p, u64p, u64p]
    L.rl_sampler_seed.argtypes = [C.POINTER(abi.Sampler), C.c_uint64, C.c_int]
    L.rl_sampler_seed.restype = None
    L.rl_sampler_next_u64.argtypes = [C.POINTER(abi.Sampler)]
    L.rl_sampler_next_u64.restype = C.c_uint64
    L.rl_sampler_next_f32.argtypes = [C.POINTER(abi.Sampler)]
    L.rl_sampler_next_f32.restype = C.c_float
    L.rl_path_params_default.argtypes = [C.POINTER(abi.PathParams)]
    L.rl_path_params_default.restype = None
    L.rl_context_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rl_context_destroy.argtypes = [vp]
    L.rl_context_destroy.restype = None
    L.rl_last_error.restype = C.c_char_p
    L.rl_block_count.argtypes = [C.c_uint32, C.c_uint32]
    L.rl_block_count.restype = C.c_size_t
    L.rl_generate_block_seeds.argtypes = [C.POINTER(abi.Sampler), C.c_uint32, C.c_uint32, u64p, C.c_size_t]
    L.rl_render_path.argtypes = [vp, C.POINTER(abi.PathParams), u64p, C.c_size_t, vp, C.c_int, vp, C.POINTER(abi.RenderStats)]
    L.rl_render_path_frames.argtypes = [C.POINTER(vp), C.c_size_t, C.POINTER(abi.PathParams), C.POINTER(u64p), C.c_size_t, C.c_size_t, C.POINTER(C.POINTER(C.c_float)), C.POINTER(abi.RenderStats)]
    for fn in (L.rl_render_ao, L.rl_render_direct):
        fn.argtypes = [vp, C.POINTER(abi.McParams), u64p, C.c_size_t, vp, C.c_int, vp, C.POINTER(abi.RenderStats)]
    L.rl_multi_create.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.rl_multi_destroy.argtypes = [vp]
    L.rl_multi_destroy.restype = None
    L.rl_multi_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rl_multi_render_path.argtypes = [vp, C.POINTER(abi.PathParams), u64p, C.c_size_t, vp, C.POINTER(abi.RenderStats)]
    L.rl_multi_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.rl_multi_shard_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(abi.RenderStats)]
    L.rl_trace_batch.argtypes = [vp, C.c_size_t, f32p, f32p, f32p, f32p, f32p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rl_visible_batch.argtypes = [vp, C.c_size_t, f32p, f32p, C.POINTER(C.c_uint8)]
    for fn in (L.rl_save_pfm, L.rl_save_png, L.rl_save_exr, L.rl_save_image):
        fn.argtypes = [C.c_char_p, f32p, C.c_uint32, C.c_uint32]
    L.rl_build_info.restype = C.c_char_p
    # test hooks
    L.rl_debug_numerics.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p]
    L.rl_debug_bvh.argtypes = [vp, u64p, u64p, f32p, u64p, u64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rl_debug_bvh_sizes.argtypes = [vp, u64p, u64p, u32p, C.POINTER(C.c_int)]
    L.rl_debug_camera_ray.argtypes = [vp, C.c_float, C.c_float, f32p, f32p]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        msg = (lib().rl_last_error() or b"").decode("utf-8", "replace")
        raise (NoDeviceError if rc == RL_ERR_NO_DEVICE else RustlightError)(rc, msg)
    return rc


class IndependentSampler:
    """IndependentSampler { rnd: SmallRng::seed_from_u64(seed) } (examples/cli.rs:886-890)."""

    def __init__(self, seed: int, variant: int = 0):
        self.s = abi.Sampler()
        self.variant = variant
        lib().rl_sampler_seed(C.byref(self.s), seed, variant)

    def next(self) -> float:
        return float(lib().rl_sampler_next_f32(C.byref(self.s)))

    def next_u64(self) -> int:
        return int(lib().rl_sampler_next_u64(C.byref(self.s)))

    def block_seeds(self, width: int, height: int) -> np.ndarray:
        """generate_img_blocks: one clone_box seed per 16x16 block, x-major (integrators/mod.rs:357-371)."""
        n = lib().rl_block_count(width, height)
        seeds = np.zeros(n, dtype=np.uint64)
        _check(lib().rl_generate_block_seeds(C.byref(self.s), width, height, abi.u64ptr(seeds), n))
        return seeds


class Scene:
    """Host-side flattened `Scene` (src/scene.rs:16-30)."""

    def __init__(self, sd: Optional[S.SceneData] = None, handle=None, camera_matrices=None):
        """camera_matrices = (sample_to_camera, to_world), column-major 16-vectors: the camera as rustlight's `Camera` holds it
        (rl_scene_set_camera_matrices) instead of Camera::new's arguments."""
        L = lib()
        self.sd = sd
        if handle is not None:
            self.h = handle
        else:
            h = C.c_void_p()
            _check(L.rl_scene_create(C.byref(h)))
            self.h = h
            tw = np.ascontiguousarray(sd.to_world, dtype=np.float32)
            if camera_matrices is not None:
                stc, twm = (np.ascontiguousarray(m, dtype=np.float32).ravel() for m in camera_matrices)
                _check(L.rl_scene_set_camera_matrices(self.h, sd.width, sd.height, abi.fptr(stc), abi.fptr(twm)))
            else:
                _check(L.rl_scene_set_camera(self.h, sd.width, sd.height, sd.fov, sd.fov_axis, abi.fptr(tw), int(sd.flip)))
            for (w, hgt, rgb) in sd.bitmaps:
                a = np.ascontiguousarray(rgb, dtype=np.float32)
                _check(L.rl_scene_add_bitmap(self.h, w, hgt, abi.fptr(a)))
            for k, m in enumerate(sd.meshes):
                v, i, n, uv, e = abi.mesh_arrays(m)
                bd = abi.bsdf_desc(m.bsdf)
                _check(L.rl_scene_add_mesh(self.h, abi.fptr(v), v.shape[0], abi.u32ptr(i), i.shape[0], abi.fptr(n),
                                           abi.fptr(uv), C.byref(bd), abi.fptr(e)))
                if getattr(m, "emission_kind", None):          # EmissionType::HSV / Texture
                    ek = m.emission_kind
                    _check(L.rl_scene_set_mesh_emission(self.h, k, 1 if ek[0] == "hsv" else 2, float(ek[1]), int(ek[2]) if len(ek) > 2 else -1))
            if sd.medium is not None:
                sa = np.asarray(sd.medium.sigma_a, dtype=np.float32)
                ss = np.asarray(sd.medium.sigma_s, dtype=np.float32)
                _check(L.rl_scene_set_medium(self.h, abi.fptr(sa), abi.fptr(ss), sd.medium.phase, sd.medium.g))
            for lt in sd.lights:
                a = np.asarray(lt["a"], np.float32)
                b = np.asarray(lt["intensity"], np.float32)
                fn = L.rl_scene_add_point_light if lt["type"] == "point" else L.rl_scene_add_directional_light
                _check(fn(self.h, abi.fptr(a), abi.fptr(b)))
            if sd.environment is not None:
                e = np.asarray(sd.environment, np.float32)
                _check(L.rl_scene_set_environment(self.h, abi.fptr(e)))
            if sd.environment_map is not None:
                em = np.ascontiguousarray(sd.environment_map, np.float32)
                _check(L.rl_scene_set_environment_map(self.h, em.shape[1], em.shape[0], abi.fptr(em)))
        if sd is not None and getattr(sd, "use_ats", False):
            _check(L.rl_scene_enable_ats(self.h, 1))
        _check(L.rl_scene_build_emitters(self.h))

    def camera_matrices(self):
        """(sample_to_camera[16], to_world[16], position[3]) the scene's camera uses, column-major."""
        a, b, c = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(3, np.float32)
        _check(lib().rl_scene_get_camera_matrices(self.h, abi.fptr(a), abi.fptr(b), abi.fptr(c)))
        return a, b, c

    @classmethod
    def from_desc(cls, sd: S.SceneData, camera_matrices=None) -> "Scene":
        """The same scene through the one-call POD entry (rl_scene_create_from_desc, SURVEY §8(b) "SceneDesc"); camera_matrices as in __init__."""
        keep = []                                  # numpy arrays must outlive the call
        def fp(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32); keep.append(a)
            return abi.fptr(a)
        d = abi.SceneDesc()
        d.width, d.height, d.fov_degrees, d.fov_axis, d.flip = sd.width, sd.height, sd.fov, sd.fov_axis, int(sd.flip)
        d.to_world = (C.c_float * 16)(*np.asarray(sd.to_world, np.float32).ravel())
        if camera_matrices is not None:
            d.has_camera_matrices = 1
            d.sample_to_camera = (C.c_float * 16)(*np.asarray(camera_matrices[0], np.float32).ravel())
            d.to_world = (C.c_float * 16)(*np.asarray(camera_matrices[1], np.float32).ravel())
            d.fov_degrees, d.fov_axis, d.flip = 0.0, 0, 0
        meshes = (abi.MeshDesc * max(1, len(sd.meshes)))()
        for k, m in enumerate(sd.meshes):
            v, i, n, uv, e = abi.mesh_arrays(m)
            keep.extend([v, i])
            meshes[k].vertices, meshes[k].n_vertices = abi.fptr(v), v.shape[0]
            meshes[k].indices, meshes[k].n_triangles = abi.u32ptr(i), i.shape[0]
            meshes[k].normals, meshes[k].uv = fp(n), fp(uv)
            meshes[k].bsdf = abi.bsdf_desc(m.bsdf)
            if e is not None:
                meshes[k].has_emission = 1
                meshes[k].emission_rgb = (C.c_float * 3)(*e)
            if getattr(m, "emission_kind", None):
                ek = m.emission_kind
                meshes[k].emission_type, meshes[k].emission_scale, meshes[k].emission_bitmap_id = (1 if ek[0] == "hsv" else 2), float(ek[1]), (int(ek[2]) if len(ek) > 2 else -1)
        d.meshes, d.n_meshes = meshes, len(sd.meshes)
        bitmaps = (abi.BitmapDesc * max(1, len(sd.bitmaps)))()
        for k, (w, h, rgb) in enumerate(sd.bitmaps):
            bitmaps[k].width, bitmaps[k].height, bitmaps[k].rgb = w, h, fp(rgb)
        d.bitmaps, d.n_bitmaps = bitmaps, len(sd.bitmaps)
        lights = (abi.LightDesc * max(1, len(sd.lights)))()
        for k, lt in enumerate(sd.lights):
            lights[k].kind = 0 if lt["type"] == "point" else 1
            lights[k].a = (C.c_float * 3)(*lt["a"])
            lights[k].intensity = (C.c_float * 3)(*lt["intensity"])
        d.lights, d.n_lights = lights, len(sd.lights)
        if sd.environment is not None:
            d.has_environment = 1
            d.environment_rgb = (C.c_float * 3)(*sd.environment)
        if sd.environment_map is not None:
            em = np.ascontiguousarray(sd.environment_map, np.float32); keep.append(em)
            d.env_map_width, d.env_map_height, d.env_map_rgb = em.shape[1], em.shape[0], abi.fptr(em)
        if sd.medium is not None:
            d.has_medium = 1
            d.sigma_a = (C.c_float * 3)(*sd.medium.sigma_a)
            d.sigma_s = (C.c_float * 3)(*sd.medium.sigma_s)
            d.phase_type, d.g = sd.medium.phase, sd.medium.g
        d.build_ats = int(bool(getattr(sd, "use_ats", False)))
        h = C.c_void_p()
        _check(lib().rl_scene_create_from_desc(C.byref(d), C.byref(h)))
        obj = cls.__new__(cls)
        obj.sd, obj.h = sd, h
        return obj

    @classmethod
    def load_pbrt(cls, path: str, use_shading_normals: bool = True) -> "Scene":
        h = C.c_void_p()
        _check(lib().rl_scene_load_pbrt(path.encode(), int(use_shading_normals), C.byref(h)))
        return cls(None, handle=h)

    @classmethod
    def load_mitsuba(cls, path: str, use_shading_normals: bool = True) -> "Scene":
        h = C.c_void_p()
        _check(lib().rl_scene_load_mitsuba(path.encode(), int(use_shading_normals), C.byref(h)))
        return cls(None, handle=h)

    @classmethod
    def load(cls, path: str, use_shading_normals: bool = True) -> "Scene":
        """SceneLoaderManager::load: .pbrt or .xml by extension (src/scene_loader.rs:27-58)."""
        h = C.c_void_p()
        _check(lib().rl_scene_load(path.encode(), int(use_shading_normals), C.byref(h)))
        return cls(None, handle=h)

    def set_medium(self, sigma_a, sigma_s, phase=S.PHASE_ISOTROPIC, g=0.0):
        sa = np.asarray(sigma_a, dtype=np.float32)
        ss = np.asarray(sigma_s, dtype=np.float32)
        _check(lib().rl_scene_set_medium(self.h, abi.fptr(sa), abi.fptr(ss), phase, g))

    def __del__(self):
        try:
            lib().rl_scene_destroy(self.h)
        except Exception:
            pass

    @property
    def size(self):
        w, h = C.c_uint32(), C.c_uint32()
        _check(lib().rl_scene_image_size(self.h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().rl_scene_counts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"meshes": a.value, "triangles": b.value, "emitters": c.value}

    def debug_bvh(self):
        nn, npr = C.c_uint64(), C.c_uint64()
        _check(lib().rl_debug_bvh(self.h, C.byref(nn), C.byref(npr), None, None, None, None, None))
        boxes = np.zeros((nn.value, 6), np.float32)
        info = np.zeros(nn.value, np.uint64)
        count = np.zeros(nn.value, np.uint64)
        pm = np.zeros(npr.value, np.int32)
        pt = np.zeros(npr.value, np.int32)
        _check(lib().rl_debug_bvh(self.h, C.byref(nn), C.byref(npr), abi.fptr(boxes), abi.u64ptr(info), abi.u64ptr(count),
                                  pm.ctypes.data_as(C.POINTER(C.c_int32)), pt.ctypes.data_as(C.POINTER(C.c_int32))))
        return boxes, info, count, pm, pt

    def debug_emitters_cdf(self):
        n = C.c_uint64(0)
        L = lib()
        L.rl_debug_emitters_cdf.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        _check(L.rl_debug_emitters_cdf(self.h, C.byref(n), None))
        out = np.zeros(n.value, np.float32)
        _check(L.rl_debug_emitters_cdf(self.h, C.byref(n), abi.fptr(out)))
        return out

    def debug_ats(self):
        """(nodes [n, 16] f32 — struct LightNode, link words are int32 bit patterns —, light_emitter, light_prim)."""
        L = lib()
        i32p = C.POINTER(C.c_int32)
        L.rl_debug_ats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint64), i32p, i32p]
        nn, nl = C.c_uint64(), C.c_uint64()
        _check(L.rl_debug_ats(self.h, C.byref(nn), None, C.byref(nl), None, None))
        nodes = np.zeros((nn.value, 16), np.float32); le = np.zeros(nl.value, np.int32); lp = np.zeros(nl.value, np.int32)
        _check(L.rl_debug_ats(self.h, C.byref(nn), abi.fptr(nodes), C.byref(nl), le.ctypes.data_as(i32p), lp.ctypes.data_as(i32p)))
        return nodes, le, lp

    def camera_ray(self, px, py):
        o = (C.c_float * 3)()
        d = (C.c_float * 3)()
        _check(lib().rl_debug_camera_ray(self.h, px, py, o, d))
        return np.array(o[:], np.float32), np.array(d[:], np.float32)


def path_params(spp=1, min_depth=0, max_depth=None, rr_depth=0, strategy=STRATEGY_ALL, single_scattering=False,
                stream_mode=STREAM_PER_SAMPLE, seed_variant=0, shard_index=0, shard_count=1, pool_slots=0, pipeline=0, sample_split=0, numerics=0) -> abi.PathParams:
    """rl_path_params for the tests and bench.py.  NOTE the default stream mode: this helper defaults to the throughput decomposition
    (STREAM_PER_SAMPLE), which most parity tests exercise; the plugin-level surfaces — rl_path_params_default, the C++ / Python
    IntegratorPathTracing, the CLI — default to rustlight's own RL_STREAM_REFERENCE_ORDER."""
    p = abi.PathParams()
    lib().rl_path_params_default(C.byref(p))
    p.spp = spp
    p.has_min_depth, p.min_depth = (0, 0) if min_depth is None else (1, min_depth)
    p.has_max_depth, p.max_depth = (0, 0) if max_depth is None else (1, max_depth)
    p.has_rr_depth, p.rr_depth = (0, 0) if rr_depth is None else (1, rr_depth)
    p.strategy = strategy
    p.single_scattering = int(single_scattering)
    p.stream_mode = stream_mode
    p.seed_variant = seed_variant
    p.shard_index, p.shard_count = shard_index, shard_count
    p.pool_slots = pool_slots
    p.pipeline = pipeline
    p.sample_split = sample_split
    p.numerics = numerics
    return p


class Context:
    """Device context = BVHAccel::new(scene) + the scene uploaded to one MI355X."""

    def __init__(self, scene: Scene, device: int = 0):
        self.scene = scene
        self.device = device
        h = C.c_void_p()
        _check(lib().rl_context_create(scene.h, device, C.byref(h)))
        self.h = h
        self.width, self.height = scene.size

    def close(self):
        if getattr(self, "h", None):
            lib().rl_context_destroy(self.h)
            self.h = None

    def set_option(self, name: str, value=None):
        """rl_context_set_option: an execution option of this context (csrc/kernels/knobs.h: "spec_force", "no_overlap", "state_budget_mb", ...; none changes a
        result), `None` = back to the default.  The context took the environment's RL_<NAME> values when it was created; renders never read the environment."""
        _check(lib().rl_context_set_option(self.h, name.lower().encode(), None if value is None else str(value).encode()))

    def get_option(self, name: str):
        v = lib().rl_context_get_option(self.h, name.lower().encode())
        return None if v is None else v.decode()

    @contextlib.contextmanager
    def options(self, **kw):
        """`with ctx.options(spec_force=1, no_overlap=1): ...` — the options set inside the block, back to what they were after it."""
        before = {k: self.get_option(k) for k in kw}
        try:
            for k, v in kw.items():
                self.set_option(k, v)
            yield self
        finally:
            for k, v in before.items():
                self.set_option(k, v)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, seeds: np.ndarray, params: abi.PathParams, out_device_ptr: Optional[int] = None, stream: Optional[int] = None, out_host_ptr: Optional[int] = None):
        """Integrator::compute.  Returns (image HxWx3 f32 | None when rendering into a device pointer or a caller's (e.g. pinned) host buffer, stats dict)."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        st = abi.RenderStats()
        if out_host_ptr is not None:
            _check(lib().rl_render_path(self.h, C.byref(params), abi.u64ptr(seeds), seeds.shape[0], C.c_void_p(out_host_ptr), 0,
                                        C.c_void_p(stream) if stream else None, C.byref(st)))
            return None, st.as_dict()
        if out_device_ptr is None:
            img = np.zeros((self.height, self.width, 3), dtype=np.float32)
            _check(lib().rl_render_path(self.h, C.byref(params), abi.u64ptr(seeds), seeds.shape[0], img.ctypes.data_as(C.c_void_p), 0,
                                        C.c_void_p(stream) if stream else None, C.byref(st)))
            return img, st.as_dict()
        _check(lib().rl_render_path(self.h, C.byref(params), abi.u64ptr(seeds), seeds.shape[0], C.c_void_p(out_device_ptr), 1,
                                    C.c_void_p(stream) if stream else None, C.byref(st)))
        return None, st.as_dict()

    def _render_mc(self, fn, seeds, spp=1, stream_mode=STREAM_PER_SAMPLE, seed_variant=0, shard_index=0, shard_count=1,
                   max_distance=1.0, normal_correction=False, nb_bsdf_samples=1, nb_light_samples=1):
        p = abi.McParams()
        p.spp, p.stream_mode, p.seed_variant, p.shard_index, p.shard_count = spp, stream_mode, seed_variant, shard_index, shard_count
        p.has_max_distance, p.max_distance = (0, 0.0) if max_distance is None else (1, max_distance)
        p.normal_correction = int(normal_correction)
        p.nb_bsdf_samples, p.nb_light_samples = nb_bsdf_samples, nb_light_samples
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        st = abi.RenderStats()
        img = np.zeros((self.height, self.width, 3), dtype=np.float32)
        _check(fn(self.h, C.byref(p), abi.u64ptr(seeds), seeds.shape[0], img.ctypes.data_as(C.c_void_p), 0, None, C.byref(st)))
        return img, st.as_dict()

    def render_ao(self, seeds, **kw):
        """IntegratorAO { max_distance, normal_correction } (src/integrators/ao.rs)."""
        return self._render_mc(lib().rl_render_ao, seeds, **kw)

    def render_direct(self, seeds, **kw):
        """IntegratorDirect { nb_bsdf_samples, nb_light_samples } (src/integrators/direct.rs)."""
        return self._render_mc(lib().rl_render_direct, seeds, **kw)

    def trace(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        m = np.zeros(n, np.int32); tr = np.zeros(n, np.int32)
        _check(lib().rl_trace_batch(self.h, n, abi.fptr(o), abi.fptr(d), abi.fptr(t), abi.fptr(u), abi.fptr(v),
                                    m.ctypes.data_as(C.POINTER(C.c_int32)), tr.ctypes.data_as(C.POINTER(C.c_int32))))
        return t, u, v, m, tr

    def trace_fast(self, origins, directions):
        """Test hook: closest hits through the tolerance build's traversal of a streaming scene (quantised BVH4): (t, mesh, tri, node trips per ray)."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32); m = np.zeros(n, np.int32); tr = np.zeros(n, np.int32); st = np.zeros(n, np.int32)
        i32p = C.POINTER(C.c_int32)
        fn = lib().rl_debug_trace_batch_fast
        fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), i32p, i32p, i32p]
        _check(fn(self.h, n, abi.fptr(o), abi.fptr(d), abi.fptr(t), m.ctypes.data_as(i32p), tr.ctypes.data_as(i32p), st.ctypes.data_as(i32p)))
        return t, m, tr, st

    def trace_two_level(self, origins, directions, segment_lengths=None):
        """Test hook: rl_trace_batch through the two-level node records (trace.hip.h: traverse2): (t, u, v, mesh, tri, node trips per ray); with `segment_lengths`
        the any-hit form: (found, node trips per ray)."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        any_hit = segment_lengths is not None
        t = np.ascontiguousarray(segment_lengths, dtype=np.float32).copy() if any_hit else np.zeros(n, np.float32)
        u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        m = np.zeros(n, np.int32); tr = np.zeros(n, np.int32); st = np.zeros(n, np.int32)
        i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        fn = lib().rl_debug_trace_batch_two_level
        fn.argtypes = [C.c_void_p, C.c_size_t, f32p, f32p, f32p, f32p, f32p, i32p, i32p, i32p, C.c_int]
        _check(fn(self.h, n, abi.fptr(o), abi.fptr(d), abi.fptr(t), abi.fptr(u), abi.fptr(v), m.ctypes.data_as(i32p), tr.ctypes.data_as(i32p), st.ctypes.data_as(i32p), int(any_hit)))
        return (t != 0.0, st) if any_hit else (t, u, v, m, tr, st)

    def visible(self, p0, p1):
        a = np.ascontiguousarray(p0, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(a.shape[0], np.uint8)
        _check(lib().rl_visible_batch(self.h, a.shape[0], abi.fptr(a), abi.fptr(b), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def debug_sizes(self):
        a, b = C.c_uint64(), C.c_uint64()
        d = C.c_uint32()
        l = C.c_int()
        _check(lib().rl_debug_bvh_sizes(self.h, C.byref(a), C.byref(b), C.byref(d), C.byref(l)))
        return {"ref_nodes": a.value, "prims": b.value, "stack_depth": d.value, "lds_scene": bool(l.value)}


class MultiContext:
    """N device contexts of one node + an RCCL communicator clique in ONE process (rl_multi_*): shard renders on one host thread
    per GPU, a single ncclReduce over xGMI, one download from the root."""

    def __init__(self, scene: Scene, n_shards: int, devices=None):
        self.scene = scene
        h = C.c_void_p()
        dv = None if devices is None else (C.c_int * n_shards)(*devices)
        _check(lib().rl_multi_create(scene.h, dv, n_shards, C.byref(h)))
        self.h = h
        self.width, self.height = scene.size

    def info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(lib().rl_multi_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"shards": a.value, "comm_ranks": b.value, "rccl_version": c.value}

    def describe(self) -> dict:
        """rl_multi_describe: devices, peer access matrix, how the framebuffers are merged, per-shard kernel ms of the last render."""
        import json

        buf = C.create_string_buffer(1 << 16)
        _check(lib().rl_multi_describe(self.h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def shard_stats(self, shard: int):
        dev, st = C.c_int(), abi.RenderStats()
        _check(lib().rl_multi_shard_stats(self.h, shard, C.byref(dev), C.byref(st)))
        return dev.value, st.as_dict()

    def render(self, seeds: np.ndarray, params: abi.PathParams):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        st = abi.RenderStats()
        img = np.zeros((self.height, self.width, 3), dtype=np.float32)
        _check(lib().rl_multi_render_path(self.h, C.byref(params), abi.u64ptr(seeds), seeds.shape[0], img.ctypes.data_as(C.c_void_p), C.byref(st)))
        return img, st.as_dict()

    def close(self):
        if getattr(self, "h", None):
            lib().rl_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IntegratorPathTracing:
    """struct IntegratorPathTracing (src/integrators/explicit/path.rs:14-20) + Integrator::compute."""

    def __init__(self, min_depth=0, max_depth=None, rr_depth=0, strategy=STRATEGY_ALL, single_scattering=False,
                 stream_mode=STREAM_REFERENCE_ORDER, device=0, numerics=NUMERICS_EXACT, frames_in_flight=1, options=None):
        """stream_mode: the plugin's default is rustlight's own per-block stream order (seed-for-seed the reference's image);
        STREAM_PER_SAMPLE is the opt-in throughput decomposition.  frames_in_flight (MI355X-specific, not in the reference): how many
        independent frames `compute_frames` — and the progressive wrappers through it — keep on the GPU at once (one device context and
        one host thread each; the images are those of one frame after the other)."""
        self.min_depth, self.max_depth, self.rr_depth = min_depth, max_depth, rr_depth
        self.strategy, self.single_scattering = strategy, single_scattering
        self.stream_mode, self.device, self.numerics = stream_mode, device, numerics
        self.frames_in_flight = max(1, int(frames_in_flight))
        self.options = dict(options or {})          # execution options of every context this integrator creates (Context.set_option; none changes an image)
        self.last_stats = None
        self._ctx = None
        self._extra = []

    def _new_context(self, scene: Scene) -> "Context":
        c = Context(scene, self.device)
        for k, v in self.options.items():
            c.set_option(k, v)
        return c

    def compute(self, sampler: IndependentSampler, scene: Scene, nb_samples: int = 1):
        """IntegratorType::compute (integrators/mod.rs:274-338): BVH build (untimed) then the render."""
        if self._ctx is None or self._ctx.scene is not scene:
            self._ctx = self._new_context(scene)
        w, h = scene.size
        seeds = sampler.block_seeds(w, h)
        p = path_params(nb_samples, self.min_depth, self.max_depth, self.rr_depth, self.strategy, self.single_scattering,
                        self.stream_mode, sampler.variant, numerics=self.numerics)
        img, self.last_stats = self._ctx.render(seeds, p)
        return img

    def compute_frames(self, sampler: IndependentSampler, scene: Scene, nb_samples: int, n_frames: int):
        """`n_frames` consecutive `compute` calls — the block seeds of every frame are drawn from the master sampler in call order, exactly
        as that many sequential calls would draw them — with up to `frames_in_flight` of them on the GPU at once.  A frame's render is a
        chain of dependent launches that leaves much of the chip idle at its tail (reference-order streams: the chain pass ends with its
        slowest wave); another context's frame fills it.  Returns the images in frame order."""
        if self._ctx is None or self._ctx.scene is not scene:
            self._ctx = self._new_context(scene)
            self._extra = []
        k = min(self.frames_in_flight, max(1, n_frames))
        while len(self._extra) < k - 1:
            self._extra.append(self._new_context(scene))
        w, h = scene.size
        jobs = [(sampler.block_seeds(w, h),
                 path_params(nb_samples, self.min_depth, self.max_depth, self.rr_depth, self.strategy, self.single_scattering,
                             self.stream_mode, sampler.variant, numerics=self.numerics)) for _ in range(n_frames)]
        out = render_in_flight([self._ctx] + self._extra[:k - 1], jobs)
        if out:
            self.last_stats = out[-1][1]
        return [img for img, _ in out]


def render_frames(contexts, seeds_list, params: abi.PathParams):
    """rl_render_path_frames: the frames of `seeds_list` (block seeds per frame) with the same parameters, frame f on contexts[f % len(contexts)] — the library's own
    threads.  Returns ([image, ...], [stats dict, ...]) in frame order."""
    n = len(seeds_list)
    w, h = contexts[0].width, contexts[0].height
    seeds = [np.ascontiguousarray(s, dtype=np.uint64) for s in seeds_list]
    imgs = [np.zeros((h, w, 3), dtype=np.float32) for _ in range(n)]
    vp, u64p, fp = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    c_ctx = (vp * len(contexts))(*[c.h.value for c in contexts])
    c_seeds = (u64p * max(n, 1))(*[abi.u64ptr(s) for s in seeds])
    c_out = (fp * max(n, 1))(*[abi.fptr(i) for i in imgs])
    st = (abi.RenderStats * max(n, 1))()
    _check(lib().rl_render_path_frames(c_ctx, len(contexts), C.byref(params), c_seeds, seeds[0].shape[0] if n else 0, n, c_out, st))
    return imgs, [st[f].as_dict() for f in range(n)]


def render_in_flight(contexts, jobs):
    """Render `jobs` = [(block seeds, path params), ...] on `contexts` (same scene, any devices), job j on context j % len(contexts), one host
    thread per context (rl_render_path is synchronous; contexts own their streams and share nothing mutable, and ctypes releases the GIL
    for the call).  Returns [(image, stats), ...] in job order; the first error is re-raised after every thread has stopped."""
    import threading
    k = max(1, min(len(contexts), len(jobs)))
    out = [None] * len(jobs)
    errors = []

    def work(c):
        try:
            for j in range(c, len(jobs), k):
                if errors:
                    return
                out[j] = contexts[c].render(*jobs[j])
        except Exception as e:      # noqa: BLE001 — handed to the caller below
            errors.append(e)

    if k == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(c,)) for c in range(k)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return out


def _load_with(fn_name: str, path: str) -> np.ndarray:
    w, h = C.c_uint32(), C.c_uint32()
    fn = getattr(lib(), fn_name)
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_size_t]
    _check(fn(path.encode(), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 3), np.float32)
    _check(fn(path.encode(), C.byref(w), C.byref(h), abi.fptr(out), out.size))
    return out


def load_pfm(path: str) -> np.ndarray:
    """Bitmap::read_pfm (src/structure.rs:563-607) -> H x W x 3 float32, top row first."""
    return _load_with("rl_load_pfm", path)


def load_image(path: str) -> np.ndarray:
    """Bitmap::read (src/structure.rs:670-683): .pfm / .png -> H x W x 3 float32."""
    return _load_with("rl_load_image", path)


def save_pfm(path: str, img: np.ndarray):
    a = np.ascontiguousarray(img, dtype=np.float32)
    _check(lib().rl_save_pfm(path.encode(), abi.fptr(a), a.shape[1], a.shape[0]))


def save_image(path: str, img: np.ndarray):
    """Bitmap::save: .pfm | .png | .exr by extension (src/structure.rs:528-545)."""
    a = np.ascontiguousarray(img, dtype=np.float32)
    _check(lib().rl_save_image(path.encode(), abi.fptr(a), a.shape[1], a.shape[0]))


class IntegratorAverage:
    """IntegratorAverage { time_out, integrator, dump_all } (src/integrators/avg.rs:5-131): re-runs the inner integrator with
    the same, advancing master sampler, keeps the running average, dumps `<base>_<iter>.<ext>` and `<base>_time.csv`."""

    def __init__(self, integrator, time_out=None, dump_all=True, max_iterations=None):
        self.integrator, self.time_out, self.dump_all, self.max_iterations = integrator, time_out, dump_all, max_iterations
        self.iterations = 0

    def compute(self, sampler, scene, nb_samples=1, output_img_path="out.pfm"):
        import time
        if not self.dump_all and self.time_out is None and self.max_iterations is None:
            raise ValueError("Impossible to have infinite approach and not dumping all images")
        base, ext = os.path.splitext(output_img_path)
        csv = open(base + "_time.csv", "w") if self.dump_all else None
        bitmap, iteration, elapsed = None, 1, 0.0
        # frames in flight (an inner integrator that has `frames_in_flight` > 1): the passes are rendered a batch at a time and folded in pass order, each
        # charged its share of the batch's time; with a time-out the passes of the last batch beyond it are dropped (their seeds are drawn: the next
        # call of the same sampler goes on after them)
        batch = max(1, getattr(self.integrator, "frames_in_flight", 1)) if hasattr(self.integrator, "compute_frames") else 1
        pending = []
        while True:
            t0 = time.perf_counter()
            if batch > 1:
                if not pending:
                    n = batch if self.max_iterations is None else min(batch, self.max_iterations - iteration + 1)
                    pending = self.integrator.compute_frames(sampler, scene, nb_samples, n)
                    share = (time.perf_counter() - t0) / len(pending)
                new = pending.pop(0)
                t0 = time.perf_counter() - share
            else:
                new = self.integrator.compute(sampler, scene, nb_samples)
            if iteration == 1:
                bitmap = new
            else:   # bitmap.scale(iteration); accumulate_bitmap; scale(1 / (iteration + 1))   (avg.rs:59-61, sic)
                bitmap = ((bitmap * np.float32(iteration)) + new) * (np.float32(1.0) / np.float32(iteration + 1))
            elapsed += time.perf_counter() - t0
            if self.dump_all:
                save_image(f"{base}_{iteration}{ext}", bitmap)
                csv.write(f"{int(elapsed)}.{int((elapsed % 1) * 1000)},\n")
            self.iterations = iteration
            if (self.time_out is not None and int(elapsed) >= self.time_out) or (self.max_iterations is not None and iteration >= self.max_iterations):
                break
            iteration += 1
        if csv:
            csv.close()
        return bitmap


class IntegratorEqualTime:
    """IntegratorEqualTime { target_time_ms, integrator } (src/integrators/equal_time.rs:4-66)."""

    def __init__(self, integrator, target_time_ms):
        self.integrator, self.target_time_ms = integrator, target_time_ms
        self.iterations = 0

    def compute(self, sampler, scene, nb_samples=1):
        import time
        bitmap, iteration, elapsed = None, 1, 0.0
        batch = max(1, getattr(self.integrator, "frames_in_flight", 1)) if hasattr(self.integrator, "compute_frames") else 1      # frames in flight: as in IntegratorAverage
        pending = []
        while True:
            t0 = time.perf_counter()
            if batch > 1:
                if not pending:
                    pending = self.integrator.compute_frames(sampler, scene, nb_samples, batch)
                    share = (time.perf_counter() - t0) / len(pending)
                new = pending.pop(0)
                t0 = time.perf_counter() - share
            else:
                new = self.integrator.compute(sampler, scene, nb_samples)
            bitmap = new if iteration == 1 else bitmap + new
            elapsed += time.perf_counter() - t0
            if elapsed * 1000.0 >= self.target_time_ms:
                break
            iteration += 1
        self.iterations = iteration
        return bitmap * (np.float32(1.0) / np.float32(iteration))


def numerics_probe(a: np.ndarray, b: np.ndarray, device: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((10, a.shape[0]), np.float32)
    _check(lib().rl_debug_numerics(device, a.shape[0], abi.fptr(a), abi.fptr(b), abi.fptr(out)))
    return out
