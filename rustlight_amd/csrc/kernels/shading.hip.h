// shading.hip.h — device-side surface interaction, BSDFs, emitter sampling and the homogeneous
// medium: the per-vertex work of rustlight's path tracer, restated for the shade kernels.
//
//   Intersection::fill_intersection      src/structure.rs:965-1059
//   BSDFColor::color                     src/bsdfs/mod.rs:31-101
//   BSDFDiffuse / Phong / Metal / Glass / Substrate     src/bsdfs/{diffuse,phong,metal,glass,substrate}.rs
//   MicrofacetDistribution, Fresnel      src/bsdfs/{distribution,utils}.rs
//   Mesh area light + EmitterSampler     src/emitter.rs:570-688, 1566-1647; src/geometry.rs:261-348
//   HomogenousVolume, PhaseFunction      src/volume.rs
//
// `MAT` template arguments select one BSDF so each shade kernel only carries that material's
// code (and registers); MAT = -1 is the generic kernel that switches at run time.
#pragma once
#include "../device_types.h"
#include "devmath.hip.h"

namespace rl {

enum { BSDF_DIFFUSE = 0, BSDF_PHONG = 1, BSDF_METAL = 2, BSDF_GLASS = 3, BSDF_SUBSTRATE = 4 };
enum { TEXTURE_CONSTANT = 0, TEXTURE_CHECKER = 1, TEXTURE_GRID = 2, TEXTURE_BITMAP = 3 };
enum { MICRO_NONE = 0, MICRO_BECKMANN = 1, MICRO_GGX = 2 };
enum { PDF_SOLID_ANGLE = 0, PDF_AREA = 1, PDF_DISCRETE = 2 };

struct SurfacePoint {   // struct Intersection (src/structure.rs:926-949)
    V3 p, n_g, n_s, wi;
    Frame frame;
    V2 uv;
    bool has_uv;
    int mesh;
};

// ---- Rust `as` casts (saturating, NaN -> 0)
RL_DEV unsigned long long f32_as_usize(float f) { if (!(f > 0.0f)) return 0ull; if (f >= 1.8446744e19f) return ~0ull; return (unsigned long long)f; }
RL_DEV int f32_as_i32(float f) { if (f != f) return 0; if (f >= 2147483648.0f) return 2147483647; if (f <= -2147483648.0f) return (-2147483647 - 1); return (int)f; }
RL_DEV float modulo1(float a) { return fmodf(fmodf(a, 1.0f) + 1.0f, 1.0f); }   // ModuloSignedExt (src/tools.rs:32-45)

RL_DEV Col tex_color(const DeviceScene& sc, const ColorTex& t, bool has_uv, V2 uv) {
    if (t.type == TEXTURE_CONSTANT) return mkc(t.c0[0], t.c0[1], t.c0[2]);
    if (!has_uv) return czero();
    if (t.type == TEXTURE_BITMAP) {   // Bitmap::pixel_uv (src/structure.rs:434-453)
        if (t.bitmap < 0) return czero();
        BitmapDesc bd = sc.bitmaps[t.bitmap];
        float ux = modulo1(uv.x), uy = modulo1(uv.y);
        unsigned long long x = f32_as_usize(ux * (float)bd.w), y = f32_as_usize(uy * (float)bd.h);
        unsigned long long i = (unsigned long long)bd.w * y + x;
        if (i >= (unsigned long long)bd.w * bd.h) return czero();
        const float* px = sc.bitmap_texels + 3ull * (bd.offset + i);
        return mkc(px[0], px[1], px[2]);
    }
    if (t.type == TEXTURE_CHECKER) {
        float px = uv.x * t.scale[0] + t.offset[0], py = uv.y * t.scale[1] + t.offset[1];
        int x = 2 * (f32_as_i32(px * 2.0f) % 2) - 1;
        int y = 2 * (f32_as_i32(py * 2.0f) % 2) - 1;
        return (x * y == 1) ? mkc(t.c0[0], t.c0[1], t.c0[2]) : mkc(t.c1[0], t.c1[1], t.c1[2]);
    }
    // grid — `uv.y + scale.y` is the reference's own expression (bsdfs/mod.rs:82)
    float px = uv.x * t.scale[0] + t.offset[0], py = uv.y + t.scale[1] + t.offset[1];
    float x = px - floorf(px), y = py - floorf(py);
    if (x > 0.5f) x -= 1.0f;
    if (y > 0.5f) y -= 1.0f;
    return (fabsf(x) < t.line_width || fabsf(y) < t.line_width) ? mkc(t.c0[0], t.c0[1], t.c0[2]) : mkc(t.c1[0], t.c1[1], t.c1[2]);
}

// LIGHTS: what the scene's emitter set may contain, known when the kernel is picked (see sample_light): LIGHTS_AREA_ONLY = mesh area lights with a CONSTANT
// emission only, no light tree — the code of everything else is compiled out of the caller.
enum { LIGHTS_ANY = 0, LIGHTS_AREA_ONLY = 1 };
// Mesh::emit(uv) (src/geometry.rs:184-206) of a mesh known to be a light: the constant colour, or one of the two uv-dependent kinds the CLI's
// `-x hvs-light` / `-x texture-light` switch every light to (a light mesh of those kinds always has uv: the host refuses it otherwise, where the
// reference's `uv.unwrap()` would panic).  The uv-dependent kinds live in a function of their own that is NOT inlined: inlined, their code raised the
// register pressure of every kernel that shades (the headline kernel went from 6 to 16 spilled VGPRs for a branch it never takes).
static __device__ __noinline__ Col mesh_emit_uv(const BitmapDesc* bitmaps, const float* texels, int type, float scale, int bitmap, bool has_uv, float ux_in, float uy_in) {
    if (type == 1) {
        const float x = fmodf(fabsf(ux_in), 1.0f);                              // uv.x.abs() % 1.0
        const Col c = x * mkc(1.0f, 0.0f, 0.0f) + (1.0f - x) * mkc(0.0f, 1.0f, 0.0f);
        return c * scale;                                                       // Color * f32 (guarded)
    }
    if (!has_uv || bitmap < 0) return czero();
    BitmapDesc bd = bitmaps[bitmap];                                            // Bitmap::pixel_uv (src/structure.rs:434-453)
    float ux = modulo1(ux_in), uy = modulo1(uy_in);
    unsigned long long x = f32_as_usize(ux * (float)bd.w), y = f32_as_usize(uy * (float)bd.h);
    unsigned long long i = (unsigned long long)bd.w * y + x;
    Col c = czero();
    if (i < (unsigned long long)bd.w * bd.h) { const float* px = texels + 3ull * (bd.offset + i); c = mkc(px[0], px[1], px[2]); }
    return c * scale;
}
template <int LIGHTS = LIGHTS_ANY>
RL_DEV Col mesh_emit(const DeviceScene& sc, const MeshRecord& mr, bool has_uv, V2 uv) {
    if (LIGHTS == LIGHTS_AREA_ONLY || __builtin_expect(mr.emission_type == 0, 1)) return mkc(mr.emission[0], mr.emission[1], mr.emission[2]);
    return mesh_emit_uv(sc.bitmaps, sc.bitmap_texels, mr.emission_type, mr.emission_scale, mr.emission_bitmap, has_uv, uv.x, uv.y);
}
// the emission at a SAMPLED point of triangle (i0, i1, i2): its uv is interpolated and then `.normalize()`d as a 2-vector (sic, geometry.rs:316-325)
static __device__ __noinline__ Col mesh_emit_sampled(const BitmapDesc* bitmaps, const float* texels, const float* uvs, int type, float scale, int bitmap, bool has_uv,
                                                     unsigned i0, unsigned i1, unsigned i2, float bx, float by, float w2) {
    float sx = 0.0f, sy = 0.0f;
    if (has_uv) {
        const float tx = uvs[2 * i0] * bx + uvs[2 * i1] * by + uvs[2 * i2] * w2, ty = uvs[2 * i0 + 1] * bx + uvs[2 * i1 + 1] * by + uvs[2 * i2 + 1] * w2;
        const float inv = div_rn(1.0f, sqrt_rn(tx * tx + ty * ty));
        sx = tx * inv; sy = ty * inv;
    }
    return mesh_emit_uv(bitmaps, texels, type, scale, bitmap, has_uv, sx, sy);
}

// Intersection::fill_intersection
RL_DEV SurfacePoint fill_intersection(const DeviceScene& sc, int prim, float hu, float hv, V3 ray_o, V3 ray_d, float t) {
    const float4* q = reinterpret_cast<const float4*>(sc.tris) + 4 * prim;
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    V3 n_g = mk3(q0.w, q1.w, q2.w);
    int mesh_id = __float_as_int(q3.y);
    int gtri = __float_as_int(q3.w);
    MeshRecord mr = sc.meshes[mesh_id];
    unsigned int i0 = sc.tri_indices[3 * gtri], i1 = sc.tri_indices[3 * gtri + 1], i2 = sc.tri_indices[3 * gtri + 2];
    SurfacePoint s;
    s.mesh = mesh_id;
    s.p = ray_o + t * ray_d;     // `p_c + t * d_c` (geometry.rs:380)
    V3 n_s;
    if (mr.flags & MESH_HAS_NORMALS) {
        V3 d0 = mk3(sc.normals[3 * i0], sc.normals[3 * i0 + 1], sc.normals[3 * i0 + 2]);
        V3 d1 = mk3(sc.normals[3 * i1], sc.normals[3 * i1 + 1], sc.normals[3 * i1 + 2]);
        V3 d2 = mk3(sc.normals[3 * i2], sc.normals[3 * i2 + 1], sc.normals[3 * i2 + 2]);
        V3 ns = d0 * (1.0f - hu - hv) + d1 * hu + d2 * hv;
        if (dot(n_g, ns) < 0.0f) n_g = -n_g;
        float l = dot(ns, ns);
        if (l == 0.0f) n_s = n_g;
        else if (l != 1.0f) n_s = ns / sqrt_rn(l);
        else n_s = ns;
    } else n_s = n_g;
    const Material& mat = sc.materials[mr.material];
    if (mat.twosided && !(mr.flags & MESH_IS_LIGHT) && dot(ray_d, n_s) > 0.0f) {   // two-sided hack (structure.rs:1006-1013)
        n_s = mk3(-n_s.x, -n_s.y, -n_s.z);
        n_g = mk3(-n_g.x, -n_g.y, -n_g.z);
    }
    s.has_uv = (mr.flags & MESH_HAS_UV) != 0;
    s.uv.x = 0.0f; s.uv.y = 0.0f;
    if (s.has_uv) {
        float w0 = 1.0f - hu - hv;
        s.uv.x = sc.uvs[2 * i0] * w0 + sc.uvs[2 * i1] * hu + sc.uvs[2 * i2] * hv;
        s.uv.y = sc.uvs[2 * i0 + 1] * w0 + sc.uvs[2 * i1 + 1] * hu + sc.uvs[2 * i2 + 1] * hv;
    }
    s.n_g = n_g; s.n_s = n_s;
    s.frame = make_frame(n_s);
    s.wi = to_local(s.frame, -ray_d);
    return s;
}

// ------------------------------------------------------------------------------------------
// bsdfs/utils.rs
RL_DEV V3 reflect_z(V3 d) { return mk3(-d.x, -d.y, d.z); }
RL_DEV float sin_theta(V3 w) { return sqrt_rn(rmax(1.0f - w.z * w.z, 0.0f)); }
RL_DEV float tan_theta(V3 w) { return div_rn(sin_theta(w), w.z); }
RL_DEV float hypot2(float a, float b) {
    if (fabsf(a) > fabsf(b)) { float r = div_rn(b, a); return fabsf(a) * sqrt_rn(1.0f + r * r); }
    else if (b != 0.0f) { float r = div_rn(a, b); return fabsf(b) * sqrt_rn(1.0f + r * r); }
    return 0.0f;
}
RL_DEV V3 reflect_vector(V3 wo, V3 n) { return (-wo) + n * 2.0f * dot(wo, n); }
RL_DEV Col fresnel_conductor(float cos_t, Col eta, Col k) {
    float c2 = cos_t * cos_t;
    float s2 = 1.0f - c2;
    float s4 = s2 * s2;
    Col temp1 = eta * eta - k * k - cval(s2);
    Col a2pb2 = safe_sqrt(temp1 * temp1 + k * k * eta * eta * 4.0f);
    Col a = safe_sqrt((a2pb2 + temp1) * 0.5f);
    Col term1 = a2pb2 + cval(c2);
    Col term2 = a * (2.0f * c2);
    Col rs2 = (term1 - term2) / (term1 + term2);
    Col term3 = a2pb2 * c2 + cval(s4);
    Col term4 = term2 * s2;
    Col rp2 = rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (rp2 + rs2);
}
RL_DEV void fresnel_dielectric(float cos_i_, float eta, float* fres, float* cos_t_out) {
    if (eta == 1.0f) { *fres = 0.0f; *cos_t_out = -cos_i_; return; }
    float scale = cos_i_ > 0.0f ? div_rn(1.0f, eta) : eta;
    float cos_t_sqr = 1.0f - (1.0f - cos_i_ * cos_i_) * (scale * scale);
    if (cos_t_sqr <= 0.0f) { *fres = 1.0f; *cos_t_out = 0.0f; return; }
    float cos_i = fabsf(cos_i_);
    float cos_t = sqrt_rn(cos_t_sqr);
    float rs = div_rn(cos_i - eta * cos_t, cos_i + eta * cos_t);
    float rp = div_rn(eta * cos_i - cos_t, eta * cos_i + cos_t);
    *cos_t_out = cos_i_ > 0.0f ? -cos_t : cos_t;
    *fres = 0.5f * (rs * rs + rp * rp);
}

// MicrofacetDistribution (bsdfs/distribution.rs)
struct Micro { int type; float au, av; };
RL_DEV float micro_eval(const Micro& d, V3 m) {
    if (m.z <= 0.0f) return 0.0f;
    float c2 = m.z * m.z;
    float bexp = div_rn(div_rn(m.x * m.x, d.au * d.au) + div_rn(m.y * m.y, d.av * d.av), c2);
    float res;
    if (d.type == MICRO_BECKMANN) res = div_rn(m_expf(-bexp), kPi * d.au * d.av * c2 * c2);
    else { float root = (1.0f + bexp) * c2; res = div_rn(1.0f, kPi * d.au * d.av * root * root); }
    if (res * m.z < 1e-20f) return 0.0f;
    return res;
}
RL_DEV float micro_pdf(const Micro& d, V3 m) { return micro_eval(d, m) * m.z; }
RL_DEV void micro_sample(const Micro& d, V2 s, V3* m, float* pdf_out) {
    float sin_phi, cos_phi;
    m_sincosf(2.0f * kPi * s.y, &sin_phi, &cos_phi);
    float alpha_sqr = d.au * d.av;
    float cos_m, pdf;
    if (d.type == MICRO_BECKMANN) {
        float tan2 = alpha_sqr * -m_logf(1.0f - s.x);
        cos_m = div_rn(1.0f, sqrt_rn(1.0f + tan2));
        pdf = div_rn(1.0f - s.x, kPi * d.au * d.av * powi_f(cos_m, 3));
    } else {
        float tan2 = div_rn(alpha_sqr * s.x, 1.0f - s.x);
        cos_m = div_rn(1.0f, sqrt_rn(1.0f + tan2));
        float tmp = 1.0f + div_rn(tan2, alpha_sqr);
        pdf = div_rn(kInvPi, d.au * d.av * powi_f(cos_m, 3) * powi_f(tmp, 2));
    }
    if (pdf < 1e-20f) pdf = 0.0f;
    float sin_m = sqrt_rn(rmax(1.0f - powi_f(cos_m, 2), 0.0f));
    *m = mk3(sin_m * cos_phi, sin_m * sin_phi, cos_m);
    *pdf_out = pdf;
}
RL_DEV float smith_g1(const Micro& d, V3 v, V3 m) {
    if (dot(v, m) * v.z <= 0.0f) return 0.0f;
    float tt = fabsf(tan_theta(v));
    if (tt == 0.0f) return 1.0f;
    float alpha = d.au;
    if (d.type == MICRO_BECKMANN) {
        float a = div_rn(1.0f, alpha * tt);
        if (a >= 1.6f) return 1.0f;
        float a2 = powi_f(a, 2);
        return div_rn(3.535f * a + 2.181f * a2, 1.0f + 2.276f * a + 2.577f * a2);
    }
    float root = alpha * tt;
    return div_rn(2.0f, 1.0f + hypot2(1.0f, root));
}
RL_DEV float micro_g(const Micro& d, V3 wi, V3 wo, V3 m) { return smith_g1(d, wi, m) * smith_g1(d, wo, m); }

struct BsdfSample { Col weight; V3 d; float pdf; int pdf_kind; };

RL_DEV Col schlick_fresnel(const DeviceScene& sc, const Material& mat, bool huv, V2 uv, float cos_t) {
    Col rs = tex_color(sc, mat.specular, huv, uv);
    return rs + (cone() - rs) * powi_f(1.0f - cos_t, 5);
}

// BSDF::pdf (solid-angle domain unless `discrete`)
template <int MAT>
RL_DEV float bsdf_pdf(const DeviceScene& sc, const Material& mat, bool huv, V2 uv, V3 wi, V3 wo, bool discrete) {
    const int type = MAT >= 0 ? MAT : mat.type;
    if (type == BSDF_DIFFUSE) {
        if (wi.z <= 0.0f) return 0.0f;
        if (wo.z <= 0.0f) return 0.0f;
        return wo.z * kInvPi;
    }
    if (type == BSDF_PHONG) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        float alpha = dot(reflect_z(wi), wo);
        float ps = 0.0f;
        if (alpha > 0.0f) ps = div_rn(mat.weight_specular * m_powf(alpha, mat.exponent) * (mat.exponent + 1.0f), 2.0f * kPi);
        float pd = (1.0f - mat.weight_specular) * wo.z * kInvPi;
        return ps + pd;
    }
    if (type == BSDF_METAL) {
        if (mat.distribution == MICRO_NONE) return 1.0f;
        V3 h = normalize(wi + wo);
        Micro d = {mat.distribution, mat.alpha_u, mat.alpha_v};
        return div_rn(micro_pdf(d, h), 4.0f * fabsf(dot(wo, h)));
    }
    if (type == BSDF_SUBSTRATE) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return 0.0f;
        V3 m = wi + wo;
        if (m.x == 0.0f && m.y == 0.0f && m.z == 0.0f) return 0.0f;
        m = normalize(m);
        if (discrete) return 0.5f;
        float pd = wo.z * kInvPi;
        float ps = 0.0f;
        if (mat.distribution != MICRO_NONE) {
            Micro d = {mat.distribution, mat.alpha_u, mat.alpha_v};
            ps = div_rn(micro_pdf(d, m), 4.0f * fabsf(dot(wo, m)));
        }
        return 0.5f * (pd + ps);
    }
    return 0.0f;   // glass: todo!() in the reference, never evaluated on this path
}

// BSDF::eval (cosine-weighted)
template <int MAT>
RL_DEV Col bsdf_eval(const DeviceScene& sc, const Material& mat, bool huv, V2 uv, V3 wi, V3 wo, bool discrete) {
    const int type = MAT >= 0 ? MAT : mat.type;
    if (type == BSDF_DIFFUSE) {
        if (wi.z <= 0.0f) return czero();
        if (wo.z > 0.0f) return tex_color(sc, mat.diffuse, huv, uv) * wo.z * kInvPi;
        return czero();
    }
    if (type == BSDF_PHONG) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return czero();
        float alpha = dot(reflect_z(wi), wo);
        Col spec = czero();
        if (alpha > 0.0f) spec = tex_color(sc, mat.specular, huv, uv) * div_rn(m_powf(alpha, mat.exponent) * (mat.exponent + 2.0f), 2.0f * kPi);
        Col diff = tex_color(sc, mat.diffuse, huv, uv) * wo.z * kInvPi;
        return spec + diff;
    }
    if (type == BSDF_METAL) {
        if (mat.distribution == MICRO_NONE)
            return tex_color(sc, mat.specular, huv, uv) * fresnel_conductor(fabsf(wi.z), tex_color(sc, mat.eta, huv, uv), tex_color(sc, mat.k, huv, uv));
        V3 h = normalize(wi + wo);
        Micro d = {mat.distribution, mat.alpha_u, mat.alpha_v};
        float dv = micro_eval(d, h);
        if (dv == 0.0f) return czero();
        Col f = tex_color(sc, mat.specular, huv, uv) * fresnel_conductor(dot(wi, h), tex_color(sc, mat.eta, huv, uv), tex_color(sc, mat.k, huv, uv));
        float g = micro_g(d, wi, wo, h);
        float model = div_rn(dv * g, 4.0f * wi.z);
        return f * model;
    }
    if (type == BSDF_SUBSTRATE) {
        if (wi.z <= 0.0f || wo.z <= 0.0f) return czero();
        V3 m = wi + wo;
        if (m.x == 0.0f && m.y == 0.0f && m.z == 0.0f) return czero();
        m = normalize(m);
        if (discrete) return schlick_fresnel(sc, mat, huv, uv, dot(wi, m));
        Col diff = tex_color(sc, mat.diffuse, huv, uv) * (cone() - tex_color(sc, mat.specular, huv, uv)) *
                   div_rn(28.0f, 23.0f * kPi) * (1.0f - powi_f(1.0f - 0.5f * fabsf(wi.z), 5)) * (1.0f - powi_f(1.0f - 0.5f * fabsf(wo.z), 5));
        Col spec = czero();
        if (mat.distribution != MICRO_NONE) {
            Micro d = {mat.distribution, mat.alpha_u, mat.alpha_v};
            float model = div_rn(micro_eval(d, m), 4.0f * fabsf(dot(wi, m)) * rmax(fabsf(wi.z), fabsf(wo.z)));
            spec = model * schlick_fresnel(sc, mat, huv, uv, dot(wi, m));
        }
        return (diff + spec) * wo.z;
    }
    return czero();
}

// BSDF::sample -> Option<SampledDirection>
template <int MAT>
RL_DEV bool bsdf_sample(const DeviceScene& sc, const Material& mat, bool huv, V2 uv, V3 wi, V2 s, BsdfSample* out) {
    const int type = MAT >= 0 ? MAT : mat.type;
    if (type == BSDF_DIFFUSE) {
        if (wi.z <= 0.0f) return false;
        V3 d = cosine_sample_hemisphere(s);
        out->weight = tex_color(sc, mat.diffuse, huv, uv);
        out->d = d; out->pdf = d.z * kInvPi; out->pdf_kind = PDF_SOLID_ANGLE;
        return true;
    }
    if (type == BSDF_PHONG) {
        if (wi.z <= 0.0f) return false;
        V3 d;
        if (s.x < mat.weight_specular) {
            s.x = div_rn(s.x, mat.weight_specular);
            float sin_a = sqrt_rn(1.0f - m_powf(s.y, div_rn(2.0f, mat.exponent + 1.0f)));
            float cos_a = m_powf(s.y, div_rn(1.0f, mat.exponent + 1.0f));
            float phi = 2.0f * kPi * s.x;
            V3 local = mk3(sin_a * m_cosf(phi), sin_a * m_sinf(phi), cos_a);
            Frame fr = make_frame(reflect_z(wi));
            d = to_world(fr, local);
            if (d.z <= 0.0f) return false;
        } else {
            s.x = div_rn(s.x - mat.weight_specular, 1.0f - mat.weight_specular);
            d = cosine_sample_hemisphere(s);
        }
        float p = bsdf_pdf<MAT>(sc, mat, huv, uv, wi, d, false);
        if (p == 0.0f) return false;
        out->weight = bsdf_eval<MAT>(sc, mat, huv, uv, wi, d, false) / p;
        out->d = d; out->pdf = p; out->pdf_kind = PDF_SOLID_ANGLE;
        return true;
    }
    if (type == BSDF_METAL) {
        if (wi.z <= 0.0f) return false;
        if (mat.distribution == MICRO_NONE) {
            out->weight = tex_color(sc, mat.specular, huv, uv) * fresnel_conductor(wi.z, tex_color(sc, mat.eta, huv, uv), tex_color(sc, mat.k, huv, uv));
            out->d = reflect_z(wi); out->pdf = 1.0f; out->pdf_kind = PDF_DISCRETE;
            return true;
        }
        Micro di = {mat.distribution, mat.alpha_u, mat.alpha_v};
        V3 m; float p;
        micro_sample(di, s, &m, &p);
        if (p == 0.0f) return false;
        V3 wo = reflect_vector(wi, m);
        if (wo.z <= 0.0f) return false;
        Col f = fresnel_conductor(dot(wi, m), tex_color(sc, mat.eta, huv, uv), tex_color(sc, mat.k, huv, uv)) * tex_color(sc, mat.specular, huv, uv);
        float w = div_rn(micro_eval(di, m) * micro_g(di, wi, wo, m) * dot(wi, m), p * wi.z);
        out->weight = w * f;
        out->d = wo; out->pdf = p; out->pdf_kind = PDF_SOLID_ANGLE;
        return true;
    }
    if (type == BSDF_GLASS) {   // Transport::Importance => no eta^2 factor (glass.rs:99-107)
        float fres, cos_t;
        fresnel_dielectric(wi.z, mat.glass_eta, &fres, &cos_t);
        if (s.x <= fres) {
            out->weight = tex_color(sc, mat.specular, huv, uv);
            out->d = reflect_z(wi);
        } else {
            float factor = 1.0f;
            float scale = cos_t < 0.0f ? -mat.glass_inv_eta : -mat.glass_eta;
            out->weight = tex_color(sc, mat.transmittance, huv, uv) * factor * factor;
            out->d = mk3(scale * wi.x, scale * wi.y, cos_t);
        }
        out->pdf = fres; out->pdf_kind = PDF_DISCRETE;
        return true;
    }
    if (type == BSDF_SUBSTRATE) {
        if (wi.z <= 0.0f) return false;
        V3 d; bool discrete = false;
        if (s.x < 0.5f) {
            s.x *= 2.0f;
            d = cosine_sample_hemisphere(s);
        } else {
            s.x = (s.x - 0.5f) * 2.0f;
            V3 m;
            if (mat.distribution == MICRO_NONE) { m = mk3(0.0f, 0.0f, 1.0f); discrete = true; }
            else {
                Micro di = {mat.distribution, mat.alpha_u, mat.alpha_v};
                float p;
                micro_sample(di, s, &m, &p);
                if (p == 0.0f) return false;
            }
            d = reflect_vector(wi, m);
            if (d.z <= 0.0f) return false;
        }
        float p = bsdf_pdf<MAT>(sc, mat, huv, uv, wi, d, discrete);
        if (p == 0.0f) return false;
        Col f = bsdf_eval<MAT>(sc, mat, huv, uv, wi, d, discrete);
        out->weight = f / p;
        out->d = d; out->pdf = p; out->pdf_kind = discrete ? PDF_DISCRETE : PDF_SOLID_ANGLE;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------
// Distribution1D::sample_discrete (src/math.rs:447-457): last index with cdf[i] <= v
RL_DEV unsigned int cdf_sample(const float* cdf, unsigned int n_entries, float v) {
    unsigned int lo = 0, hi = n_entries;
    while (lo < hi) { unsigned int mid = lo + (hi - lo) / 2; if (cdf[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo ? lo - 1 : 0u;     // cdf[0] = 0 <= v for every valid table; a NaN table (the reference panics there) must not index out of bounds
}

struct LightSample { float pdf; int pdf_kind; V3 p, n, d; Col weight; int kind; };

// ------------------------------------------------------------------------------------------
// EnvironmentLightColor (src/emitter.rs:300-425): constant colour or lat-long texture with a Distribution2D
RL_DEV float clamp_f(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }          // lib.rs:59-67
// Distribution1D::sample_continuous (math.rs:461-478) over `n` bins
RL_DEV float cdf_sample_continuous(const float* cdf, unsigned int n, float v) {
    unsigned int i = cdf_sample(cdf, n + 1, v);
    float dv = v - cdf[i];
    float p = cdf[i + 1] - cdf[i];
    if (p > 0.0f) dv = div_rn(dv, p);
    return (float)i + dv;
}
RL_DEV float env_bin_pdf(const DeviceScene& sc, unsigned long long x, unsigned long long y) {          // Distribution2D::pdf
    return div_rn(sc.env_cond_func[y * sc.env_w + x], sc.env_marg_func_int);
}
RL_DEV V2 env_to_spherical(V3 d) {                                                                     // emitter.rs:320-338
    float p = dm::atan2f_det(d.y, d.x);
    if (p < 0.0f) p = p + 2.0f * kPi;
    V2 uv; uv.x = p * kInvPi * 0.5f; uv.y = dm::acosf_det(clamp_f(d.z, -1.0f, 1.0f)) * kInvPi;
    uv.x = clamp_f(uv.x, 0.0f, 0.9999999403953552f);
    uv.y = clamp_f(uv.y, 0.0f, 0.9999999403953552f);
    return uv;
}
// Emitter::eval of the environment = scene.enviroment_luminance(d) (scene.rs:125-130)
RL_DEV Col env_eval(const DeviceScene& sc, V3 d) {
    if (sc.env_w == 0u) return mkc(sc.env_color[0], sc.env_color[1], sc.env_color[2]);
    V2 uv = env_to_spherical(d);
    float ux = modulo1(uv.x), uy = modulo1(uv.y);                                                      // Bitmap::pixel_uv
    unsigned long long x = f32_as_usize(ux * (float)sc.env_w), y = f32_as_usize(uy * (float)sc.env_h);
    unsigned long long i = (unsigned long long)sc.env_w * y + x;
    if (i >= (unsigned long long)sc.env_w * sc.env_h) return czero();
    return mkc(sc.env_texels[3 * i], sc.env_texels[3 * i + 1], sc.env_texels[3 * i + 2]);
}
// EnvironmentLight::direct_pdf (solid angle) times the emitter-selection probability (emitter.rs:407-424, 1566-1575)
RL_DEV float env_direct_pdf(const DeviceScene& sc, V3 d) {
    if (sc.env_w == 0u) return sc.env_pdf;
    V2 uv = env_to_spherical(d);
    float p = env_bin_pdf(sc, f32_as_usize(uv.x * (float)sc.env_w), f32_as_usize(uv.y * (float)sc.env_h));
    float st = m_sinf(kPi * uv.y);
    float v = st == 0.0f ? 0.0f : div_rn(p, (2.0f * powi_f(kPi, 2)) * st);
    return v * sc.env_sel_pdf;
}
// EnvironmentLightColor::sample_direction, Texture (emitter.rs:362-393)
RL_DEV void env_sample_direction(const DeviceScene& sc, V2 u, V3* d, Col* value, float* pdf) {
    float y = cdf_sample_continuous(sc.env_marg_cdf, sc.env_h, u.y);
    unsigned long long row = f32_as_usize(y);
    float x = cdf_sample_continuous(sc.env_cond_cdf + row * (sc.env_w + 1u), sc.env_w, u.x);
    x = clamp_f(x, 0.0f, (float)sc.env_w - 1.0f);
    y = clamp_f(y, 0.0f, (float)sc.env_h - 1.0f);
    unsigned long long px = f32_as_usize(x), py = f32_as_usize(y);
    const float* t = sc.env_texels + 3ull * (py * sc.env_w + px);
    float p = env_bin_pdf(sc, px, py);
    float sp, cp, st, ct;
    m_sincosf(div_rn(2.0f * kPi, (float)sc.env_w) * x, &sp, &cp);
    m_sincosf(div_rn(kPi, (float)sc.env_h) * y, &st, &ct);
    *d = mk3(st * cp, st * sp, ct);
    if (st == 0.0f) { *value = czero(); *pdf = 0.0f; }
    else { *value = mkc(t[0], t[1], t[2]); *pdf = div_rn(p, (2.0f * powi_f(kPi, 2)) * st); }
}

// solve_quadratic (src/math.rs:324-352) + BoundingSphere::intersect (src/structure.rs:894-917)
RL_DEV bool bsphere_intersect(V3 center, float radius, V3 o, V3 d, float tnear, float tfar, float* t_out) {
    V3 d_p = center - o;
    float a = length2(d);
    float b = 2.0f * dot(d_p, d);
    float c = length2(d_p) - radius * radius;
    float x0, x1;
    if (a == 0.0f) {
        if (b == 0.0f) return false;
        float v = div_rn(-c, b);
        x0 = v; x1 = v;
    } else {
        float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return false;
        float d_sqrt = sqrt_rn(disc);
        float tmp = b < 0.0f ? -0.5f * (b - d_sqrt) : -0.5f * (b + d_sqrt);
        float r0 = div_rn(tmp, a), r1 = div_rn(c, tmp);
        if (r0 > r1) { x0 = r1; x1 = r0; } else { x0 = r0; x1 = r1; }
    }
    if (x0 < tnear) { if (x1 < tfar) { *t_out = x1; return true; } return false; }
    if (x0 < tfar) { *t_out = x0; return true; }
    return false;
}

// Mesh::sample_tri + the direct_sample tail shared by Mesh::direct_sample / direct_sample_tri (geometry.rs:261-337,
// emitter.rs:609-688): point on triangle `prim` of the mesh, geometry term, solid-angle pdf from `pdf_area`.
template <int LIGHTS = LIGHTS_ANY>
RL_DEV void mesh_sample_triangle(const DeviceScene& sc, const MeshRecord& mr, unsigned int prim, float pdf_area, V3 p, V2 uv, LightSample* ls) {
    unsigned int gtri = mr.tri_base + prim;
    unsigned int i0 = sc.tri_indices[3 * gtri], i1 = sc.tri_indices[3 * gtri + 1], i2 = sc.tri_indices[3 * gtri + 2];
    V3 v0 = mk3(sc.positions[3 * i0], sc.positions[3 * i0 + 1], sc.positions[3 * i0 + 2]);
    V3 v1 = mk3(sc.positions[3 * i1], sc.positions[3 * i1 + 1], sc.positions[3 * i1 + 2]);
    V3 v2 = mk3(sc.positions[3 * i2], sc.positions[3 * i2 + 1], sc.positions[3 * i2 + 2]);
    V2 b = uniform_sample_triangle(uv);
    float w2 = 1.0f - b.x - b.y;
    V3 pos = v0 * b.x + v1 * b.y + v2 * w2;
    V3 n_g = normalize(cross(v2 - v0, v1 - v0));     // (v2-v0) x (v1-v0): geometry.rs:272-276
    if (mr.flags & MESH_HAS_NORMALS) {
        V3 n0 = mk3(sc.normals[3 * i0], sc.normals[3 * i0 + 1], sc.normals[3 * i0 + 2]);
        V3 n1 = mk3(sc.normals[3 * i1], sc.normals[3 * i1 + 1], sc.normals[3 * i1 + 2]);
        V3 n2 = mk3(sc.normals[3 * i2], sc.normals[3 * i2 + 1], sc.normals[3 * i2 + 2]);
        V3 n = n0 * b.x + n1 * b.y + n2 * w2;
        float nl = length2(n);
        if (nl == 0.0f) n = n_g;
        else if (nl != 1.0f) n = n / sqrt_rn(nl);
        if (dot(n_g, n) < 0.0f) n_g = -n_g;
    }
    if (pdf_area < 0.0f) pdf_area = div_rn(1.0f, length(cross(v1 - v0, v2 - v0)) * 0.5f);   // sample_tri: PDF::Area(1 / area_tri)
    V3 d = pos - p;
    float dist = length(d);
    if (dist != 0.0f) d = d / dist;
    float geom = dist != 0.0f ? div_rn(rmax(dot(n_g, -d), 0.0f), dist * dist) : 0.0f;
    float pdf = geom == 0.0f ? 0.0f : div_rn(pdf_area, geom);   // PDF::as_solid_angle_geom
    Col emit;
    if (LIGHTS == LIGHTS_AREA_ONLY || __builtin_expect(mr.emission_type == 0, 1)) emit = mkc(mr.emission[0], mr.emission[1], mr.emission[2]);
    else emit = mesh_emit_sampled(sc.bitmaps, sc.bitmap_texels, sc.uvs, mr.emission_type, mr.emission_scale, mr.emission_bitmap, (mr.flags & MESH_HAS_UV) != 0, i0, i1, i2, b.x, b.y, w2);
    ls->weight = pdf == 0.0f ? czero() : emit * geom / pdf_area;
    ls->pdf = pdf; ls->pdf_kind = PDF_SOLID_ANGLE;
    ls->p = pos; ls->n = n_g; ls->d = d;
}

// ------------------------------------------------------------------------------------------
// LightSamplerATS (`-x ats`, src/emitter.rs:901-1086, 1294-1367): the light tree is walked with probabilities
// proportional to LightBounds::importance_point of the two children.
// DirectionCone::subtended_directions(aabb, p).cos_theta (emitter.rs:831-846)
RL_DEV float ats_cos_subtended(V3 bmin, V3 bmax, V3 p) {
    V3 c = (bmax - bmin) * 0.5f + bmin;                 // AABB::center
    float radius = length(c - bmax);                    // AABB::to_sphere
    if (length2(p - c) < radius * radius) return -1.0f;
    float sin2 = div_rn(radius * radius, length2(c - p));
    return sqrt_rn(rmax(1.0f - sin2, 0.0f));
}
RL_DEV float ats_cos_sub(float sa, float ca, float sb, float cb) { return ca > cb ? 1.0f : ca * cb + sa * sb; }
RL_DEV float ats_sin_sub(float sa, float ca, float sb, float cb) { return ca > cb ? 1.0f : sa * cb - ca * sb; }
// LightBounds::importance_point (emitter.rs:1024-1086); two_sided is never set for mesh proxies
RL_DEV float ats_importance(const LightNode& nd, V3 p, bool has_n, V3 n) {
    V3 bmin = mk3(nd.bmin[0], nd.bmin[1], nd.bmin[2]), bmax = mk3(nd.bmax[0], nd.bmax[1], nd.bmax[2]);
    V3 pc = (bmax - bmin) * 0.5f + bmin;
    float d2 = rmax(length2(p - pc), 0.0001f);
    V3 wi = normalize(p - pc);
    float cos_theta = dot(mk3(nd.axis[0], nd.axis[1], nd.axis[2]), wi);
    float sin_theta = sqrt_rn(rmax(1.0f - cos_theta * cos_theta, 0.0f));
    float cos_u = ats_cos_subtended(bmin, bmax, p);
    float sin_u = sqrt_rn(rmax(1.0f - cos_u * cos_u, 0.0f));
    float sin_o = sqrt_rn(rmax(1.0f - nd.cos_theta_o * nd.cos_theta_o, 0.0f));
    float cos_x = ats_cos_sub(sin_theta, cos_theta, sin_o, nd.cos_theta_o);
    float sin_x = ats_sin_sub(sin_theta, cos_theta, sin_o, nd.cos_theta_o);
    float cos_p = ats_cos_sub(sin_x, cos_x, sin_u, cos_u);
    if (cos_p <= nd.cos_theta_e) return 0.0f;
    float imp = div_rn(nd.phi * cos_p, d2);
    if (has_n) {
        float cos_i = fabsf(dot(wi, n));
        float sin_i = sqrt_rn(rmax(1.0f - cos_i * cos_i, 0.0f));
        imp = imp * ats_cos_sub(sin_i, cos_i, sin_u, cos_u);
    }
    return rmax(imp, 0.0f);
}
RL_DEV float ats_prob_left(const DeviceScene& sc, const LightNode& nd, V3 p, bool has_n, V3 n) {
    float il = ats_importance(sc.ats_nodes[nd.left], p, has_n, n), ir = ats_importance(sc.ats_nodes[nd.right], p, has_n, n);
    return (il == 0.0f && ir == 0.0f) ? 0.5f : div_rn(il, il + ir);
}
// LightSamplerATS::sample (emitter.rs:1330-1367): returns the light proxy index, *pdf_sel its probability
RL_DEV int ats_sample(const DeviceScene& sc, float r, V3 p, bool has_n, V3 n, float* pdf_sel) {
    float pdf = 1.0f;
    int ni = sc.ats_root;
    for (;;) {
        const LightNode nd = sc.ats_nodes[ni];
        if (nd.left < 0 && nd.right < 0) { *pdf_sel = pdf; return nd.light; }
        float pl = ats_prob_left(sc, nd, p, has_n, n);
        if (r < pl) { r = div_rn(r, pl); ni = nd.left; pdf = pdf * pl; }
        else { r = div_rn(r - pl, 1.0f - pl); ni = nd.right; pdf = pdf * (1.0f - pl); }
    }
}
// LightSamplerATS::pdf (emitter.rs:1294-1328): product of the branch probabilities from the leaf up to the root
RL_DEV float ats_pdf(const DeviceScene& sc, unsigned int leaf, V3 p, bool has_n, V3 n) {
    int id = (int)leaf;
    float pdf = 1.0f;
    for (;;) {
        int ip = sc.ats_nodes[id].parent;
        if (ip < 0) break;
        const LightNode nd = sc.ats_nodes[ip];
        float pl = ats_prob_left(sc, nd, p, has_n, n);
        pdf = nd.left == id ? pdf * pl : pdf * (1.0f - pl);
        id = ip;
    }
    return pdf;
}

// EmitterSampler::sample_light (src/emitter.rs:1604-1639) -> Emitter::direct_sample of the emitter picked from the flux cdf,
// or direct_sample_tri of the (emitter, triangle) picked by the light tree.  `n`: Some(&its.n_s) at surfaces, None in the medium.
// LIGHTS: what the scene's emitter set may contain, known when the kernel is picked — LIGHTS_AREA_ONLY compiles the light tree, point,
// directional and environment code out of the caller (the emitter kinds are then not even looked at); results are those of LIGHTS_ANY.
template <int LIGHTS = LIGHTS_ANY>
RL_DEV LightSample sample_light(const DeviceScene& sc, V3 p, bool has_n, V3 n, float r_sel, float r, V2 uv) {
    if (LIGHTS != LIGHTS_AREA_ONLY && sc.ats_root >= 0) {
        float pdf_sel;
        int li = ats_sample(sc, r_sel, p, has_n, n, &pdf_sel);
        LightSample ls;
        ls.kind = EMITTER_MESH;
        mesh_sample_triangle<LIGHTS>(sc, sc.meshes[sc.ats_light_mesh[li]], (unsigned int)sc.ats_light_prim[li], -1.0f, p, uv, &ls);
        ls.weight = div_unguarded(ls.weight, pdf_sel);
        ls.pdf = ls.pdf * pdf_sel;
        return ls;
    }
    unsigned int id = cdf_sample(sc.emitters_cdf, sc.n_emitters + 1, r_sel);
    float pdf_sel = sc.emitters_cdf[id + 1] - sc.emitters_cdf[id];
    const EmitterRecord em = sc.emitters[id];
    LightSample ls;
    ls.kind = LIGHTS == LIGHTS_AREA_ONLY ? (int)EMITTER_MESH : em.kind;
    if (LIGHTS == LIGHTS_AREA_ONLY || em.kind == EMITTER_MESH) {
        // Mesh::direct_sample -> Mesh::sample: triangle by area cdf, pdf = Area(1 / cdf.total()) (emitter.rs:652-688, geometry.rs:340-348)
        MeshRecord mr = sc.meshes[em.mesh];
        unsigned int prim = cdf_sample(sc.mesh_cdf + mr.cdf_base, mr.n_tris + 1, r);
        mesh_sample_triangle<LIGHTS>(sc, mr, prim, mr.inv_area, p, uv, &ls);
    } else if (em.kind == EMITTER_POINT) {                 // PointEmitter::direct_sample (emitter.rs:194-213)
        V3 lp = mk3(em.v[0], em.v[1], em.v[2]);
        V3 d = lp - p;
        float dist = length(d);
        d = d / dist;
        ls.pdf = 1.0f; ls.pdf_kind = PDF_DISCRETE;
        ls.p = lp; ls.n = mk3(0.0f, 0.0f, 0.0f); ls.d = d;
        ls.weight = mkc(em.c[0], em.c[1], em.c[2]) / powi_f(dist, 2);
    } else if (em.kind == EMITTER_DIRECTIONAL) {           // DirectionalLight::direct_sample (emitter.rs:116-134)
        V3 dir = mk3(em.v[0], em.v[1], em.v[2]);
        ls.pdf = 1.0f; ls.pdf_kind = PDF_DISCRETE;
        ls.p = p - em.radius * dir; ls.n = dir; ls.d = -dir;
        ls.weight = mkc(em.c[0], em.c[1], em.c[2]);
    } else {                                               // EnvironmentLight::direct_sample (emitter.rs:473-518)
        V3 d; float pdf; Col lum = mkc(em.c[0], em.c[1], em.c[2]);
        if (sc.env_w) env_sample_direction(sc, uv, &d, &lum, &pdf);
        else { d = sample_uniform_sphere(uv); pdf = div_rn(1.0f, kPi * 4.0f); }
        float t;
        ls.pdf = pdf; ls.pdf_kind = PDF_SOLID_ANGLE; ls.d = d;
        if (!bsphere_intersect(mk3(em.center[0], em.center[1], em.center[2]), em.radius, p, d, kEps, kF32Max, &t)) {
            ls.p = mk3(0.0f, 0.0f, 0.0f); ls.n = mk3(0.0f, 0.0f, 0.0f); ls.weight = czero();
        } else {
            ls.p = p + d * t;
            ls.n = normalize(mk3(em.center[0], em.center[1], em.center[2]) - ls.p);
            ls.weight = lum / pdf;
        }
    }
    ls.weight = div_unguarded(ls.weight, pdf_sel);         // res.weight /= pdf_sel
    ls.pdf = ls.pdf * pdf_sel;                             // res.pdf = res.pdf * pdf_sel
    return ls;
}

// EmitterSampler::direct_pdf for a mesh light hit by a BSDF-sampled ray (emitter.rs:571-589, 1566-1603).  With the light
// tree: pdf of the hit triangle (1 / its area) x the probability of reaching its leaf; `n` as the caller passes it
// (None from the path tracer's MIS, Some(&its.n_s) from `direct`).
template <int LIGHTS = LIGHTS_ANY>
RL_DEV float light_direct_pdf(const DeviceScene& sc, const MeshRecord& mr, int prim_in_mesh, V3 o, V3 p, V3 n, V3 dir, bool has_ns, V3 ns) {
    float cos_light = rmax(dot(n, -dir), 0.0f);
    if (LIGHTS != LIGHTS_AREA_ONLY && sc.ats_root >= 0) {
        float tri = 0.0f;
        if (cos_light != 0.0f) {
            float geom = div_rn(cos_light, length2(p - o));
            unsigned int gtri = mr.tri_base + (unsigned int)prim_in_mesh;
            unsigned int i0 = sc.tri_indices[3 * gtri], i1 = sc.tri_indices[3 * gtri + 1], i2 = sc.tri_indices[3 * gtri + 2];
            V3 v0 = mk3(sc.positions[3 * i0], sc.positions[3 * i0 + 1], sc.positions[3 * i0 + 2]);
            V3 v1 = mk3(sc.positions[3 * i1], sc.positions[3 * i1 + 1], sc.positions[3 * i1 + 2]);
            V3 v2 = mk3(sc.positions[3 * i2], sc.positions[3 * i2 + 1], sc.positions[3 * i2 + 2]);
            float area_tri = length(cross(v1 - v0, v2 - v0)) * 0.5f;                 // Mesh::pdf_tri (geometry.rs:226-234)
            tri = div_rn(div_rn(1.0f, area_tri), geom);
        }
        return tri * ats_pdf(sc, sc.ats_leaf_of[mr.ats_base + (unsigned int)prim_in_mesh], o, has_ns, ns);
    }
    if (cos_light == 0.0f) return 0.0f * mr.emitter_pdf;
    float geom = div_rn(cos_light, length2(p - o));
    return div_rn(mr.inv_area, geom) * mr.emitter_pdf;
}

// ------------------------------------------------------------------------------------------
// HomogenousVolume (src/volume.rs:95-141) and PhaseFunction (12-68)
struct MediumSample { float t; Col w; bool exited; };
// exp per channel.  The reference CLI only ever creates grey media (`-m sigma_s[:sigma_a[:g]]` fills all three channels with one value,
// examples/cli.rs:381-385), so the three arguments are usually equal and one evaluation of the (f64, ~40-instruction) recipe serves all of them —
// same inputs, same bits.  The test is per lane but all lanes of a grey medium agree, so the branch never diverges.
RL_DEV Col cexp(Col c) {
#ifndef RL_NO_GREY_EXP
    if ((c.r == c.g) & (c.g == c.b)) { const float e = m_expf(c.r); return mkc(e, e, e); }
#endif
    return mkc(m_expf(c.r), m_expf(c.g), m_expf(c.b));
}
RL_DEV MediumSample medium_sample(const MediumRecord& m, float max_t, float u) {
    Col sigma_t = mkc(m.sigma_t[0], m.sigma_t[1], m.sigma_t[2]);
    Col sigma_s = mkc(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]);
    float u3 = u * 3.0f;
    int component = u3 != u3 ? 0 : (u3 <= 0.0f ? 0 : (u3 >= 255.0f ? 255 : (int)u3));   // `as u8`
    u = u * 3.0f - (float)component;
    float sigma_t_c = cget(sigma_t, component);
    float t = div_rn(-m_logf(1.0f - u), sigma_t_c);
    float t_min = rmin(t, max_t);
    bool exited = t >= max_t;
    Col tau = t_min * sigma_t;
    Col w = cexp(-tau);
    float pdf;
    if (exited) pdf = cavg(cexp(-tau));
    else { w = w * sigma_s; pdf = cavg(sigma_t * cexp(-tau)); }
    w = div_unguarded(w, pdf);
    MediumSample r; r.t = t_min; r.w = w; r.exited = exited;
    return r;
}
RL_DEV Col medium_transmittance(const MediumRecord& m, float tfar) {
    Col sigma_t = mkc(m.sigma_t[0], m.sigma_t[1], m.sigma_t[2]);
    Col tau = sigma_t * tfar;
    return cexp(-tau);
}
RL_DEV Col phase_eval(const MediumRecord& m, V3 w_i, V3 w_o) {
    if (m.phase == 0) return cval(div_rn(1.0f, kPi * 4.0f));
    float g = m.g;
    float tmp = 1.0f + g * g + 2.0f * g * dot(w_i, w_o);
    return cval(div_rn(kInvPi * 0.25f * (1.0f - g * g), tmp * sqrt_rn(tmp)));
}
RL_DEV float phase_pdf(const MediumRecord& m, V3 w_i, V3 w_o) { return cavg(phase_eval(m, w_i, w_o)); }
RL_DEV void phase_sample(const MediumRecord& m, V3 d_in, V2 u, V3* d, Col* weight, float* pdf) {
    if (m.phase == 0) { *d = sample_uniform_sphere(u); *weight = cone(); *pdf = div_rn(1.0f, kPi * 4.0f); return; }
    float g = m.g;
    float cos_t;
    if (fabsf(g) < 0.000001f) cos_t = 1.0f - 2.0f * u.x;
    else { float sq = div_rn(1.0f - g * g, 1.0f - g + 2.0f * g * u.x); cos_t = div_rn(1.0f + g * g - sq * sq, 2.0f * g); }
    float sin_t = sqrt_rn(rmax(1.0f - cos_t * cos_t, 0.0f));
    float sp, cp;
    m_sincosf(2.0f * kPi * u.y, &sp, &cp);
    V3 rev = d_in * -1.0f;
    *d = to_world(make_frame(rev), mk3(sin_t * cp, sin_t * sp, cos_t));
    *weight = cone();
    *pdf = phase_pdf(m, d_in, *d);
}

}  // namespace rl
