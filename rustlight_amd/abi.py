"""ctypes mirror of include/rustlight_amd.h (POD structs only — no logic)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import scenes as S


class ColorDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("color0", C.c_float * 3), ("color1", C.c_float * 3),
                ("offset", C.c_float * 2), ("scale", C.c_float * 2), ("line_width", C.c_float),
                ("bitmap_id", C.c_int32)]


class BsdfDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("diffuse", ColorDesc), ("specular", ColorDesc),
                ("transmittance", ColorDesc), ("eta", ColorDesc), ("k", ColorDesc),
                ("exponent", C.c_float), ("weight_specular", C.c_float), ("distribution", C.c_int32),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("glass_eta", C.c_float)]


class MeshDesc(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("n_vertices", C.c_size_t), ("indices", C.POINTER(C.c_uint32)), ("n_triangles", C.c_size_t),
                ("normals", C.POINTER(C.c_float)), ("uv", C.POINTER(C.c_float)), ("bsdf", BsdfDesc), ("has_emission", C.c_int32),
                ("emission_rgb", C.c_float * 3), ("emission_type", C.c_int32), ("emission_scale", C.c_float), ("emission_bitmap_id", C.c_int32)]


class BitmapDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_float))]


class LightDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float * 3), ("intensity", C.c_float * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("fov_degrees", C.c_float), ("fov_axis", C.c_int32), ("to_world", C.c_float * 16),
                ("flip", C.c_int32), ("meshes", C.POINTER(MeshDesc)), ("n_meshes", C.c_size_t), ("bitmaps", C.POINTER(BitmapDesc)),
                ("n_bitmaps", C.c_size_t), ("lights", C.POINTER(LightDesc)), ("n_lights", C.c_size_t), ("has_environment", C.c_int32),
                ("environment_rgb", C.c_float * 3), ("env_map_width", C.c_uint32), ("env_map_height", C.c_uint32),
                ("env_map_rgb", C.POINTER(C.c_float)), ("has_medium", C.c_int32), ("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3),
                ("phase_type", C.c_int32), ("g", C.c_float), ("build_ats", C.c_int32), ("has_camera_matrices", C.c_int32), ("sample_to_camera", C.c_float * 16)]


class Sampler(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


class PathParams(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("has_min_depth", C.c_int32), ("min_depth", C.c_uint32),
                ("has_max_depth", C.c_int32), ("max_depth", C.c_uint32), ("has_rr_depth", C.c_int32),
                ("rr_depth", C.c_uint32), ("strategy", C.c_int32), ("single_scattering", C.c_int32),
                ("stream_mode", C.c_int32), ("seed_variant", C.c_int32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("pool_slots", C.c_uint32), ("pipeline", C.c_uint32), ("sample_split", C.c_uint32), ("numerics", C.c_uint32)]


class McParams(C.Structure):
    _fields_ = [("spp", C.c_uint32), ("stream_mode", C.c_int32), ("seed_variant", C.c_int32), ("shard_index", C.c_uint32),
                ("shard_count", C.c_uint32), ("has_max_distance", C.c_int32), ("max_distance", C.c_float),
                ("normal_correction", C.c_int32), ("nb_bsdf_samples", C.c_uint32), ("nb_light_samples", C.c_uint32),
                ("reserved", C.c_uint32 * 4)]


class RenderStats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("vertices", C.c_uint64), ("extension_rays", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("rng_draws", C.c_uint64), ("iterations", C.c_uint64),
                ("kernel_launches", C.c_uint64), ("render_ms", C.c_double), ("ms_raygen", C.c_double),
                ("ms_extend", C.c_double), ("ms_shade", C.c_double), ("ms_shadow", C.c_double),
                ("ms_prepass", C.c_double), ("ms_other", C.c_double), ("n_extend_launches", C.c_uint64),
                ("reserved", C.c_uint64 * 4), ("chunks", C.c_uint32), ("overlapped", C.c_uint32), ("ms_eval_span", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        # k_stream_spec's counters (reference-order streams, first pass): samples walked speculatively / serially / by the estimate probes, lanes per block (0: the serial chain ran)
        d["spec_samples"], d["spec_serial_samples"], d["spec_probe_samples"], d["spec_group"] = (int(v) for v in self.reserved)
        return d


def color_desc(d: dict) -> ColorDesc:
    c = ColorDesc()
    c.type = int(d.get("type", S.TEX_CONSTANT))
    c.color0 = (C.c_float * 3)(*d.get("color0", (1.0, 1.0, 1.0)))
    c.color1 = (C.c_float * 3)(*d.get("color1", (0.0, 0.0, 0.0)))
    c.offset = (C.c_float * 2)(*d.get("offset", (0.0, 0.0)))
    c.scale = (C.c_float * 2)(*d.get("scale", (1.0, 1.0)))
    c.line_width = float(d.get("line_width", 0.0))
    c.bitmap_id = int(d.get("bitmap_id", -1))
    return c


def bsdf_desc(b: S.Bsdf) -> BsdfDesc:
    d = BsdfDesc()
    d.type = b.type
    d.diffuse = color_desc(b.diffuse)
    d.specular = color_desc(b.specular)
    d.transmittance = color_desc(b.transmittance)
    d.eta = color_desc(b.eta)
    d.k = color_desc(b.k)
    d.exponent = b.exponent
    d.weight_specular = b.weight_specular
    d.distribution = b.distribution
    d.alpha_u = b.alpha_u
    d.alpha_v = b.alpha_v
    d.glass_eta = b.glass_eta
    return d


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def u32ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def u64ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def mesh_arrays(m: S.MeshData):
    v = np.ascontiguousarray(m.vertices, dtype=np.float32)
    i = np.ascontiguousarray(m.indices, dtype=np.uint32)
    n = None if m.normals is None else np.ascontiguousarray(m.normals, dtype=np.float32)
    uv = None if m.uv is None else np.ascontiguousarray(m.uv, dtype=np.float32)
    e = None if m.emission is None else np.asarray(m.emission, dtype=np.float32)
    return v, i, n, uv, e
