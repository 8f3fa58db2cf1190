// hostmath.h — f32 vector / matrix helpers for the host-side scene preparation.
//
// The host computes, once and outside the timed region, every ray-independent quantity the
// reference recomputes per ray or per sample (triangle edges and normals, areas, cdfs, camera
// matrices).  To stay bit-compatible with rustlight these helpers use the operation order of
// cgmath 0.18 (dot = (x*x' + y*y') + z*z', normalize = v * (1/|v|), column-major Matrix4).
// Compile without FMA contraction.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace rl {

struct Vec3 {
    float x, y, z;
    float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    void set(int i, float v) { (i == 0 ? x : (i == 1 ? y : z)) = v; }
};
inline Vec3 vadd(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 vsub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 vscale(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 vdiv(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float vdot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 vcross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float vlen(Vec3 a) { return std::sqrt(vdot(a, a)); }
inline Vec3 vnormalize(Vec3 a) { return vscale(a, 1.0f / vlen(a)); }

// AABB (src/structure.rs:759-878), with Rust's NaN-ignoring min/max
struct Box3 {
    Vec3 lo{FLT_MAX, FLT_MAX, FLT_MAX}, hi{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(Vec3 p) {
        lo = {std::fmin(lo.x, p.x), std::fmin(lo.y, p.y), std::fmin(lo.z, p.z)};
        hi = {std::fmax(hi.x, p.x), std::fmax(hi.y, p.y), std::fmax(hi.z, p.z)};
    }
    void grow(const Box3& b) {
        lo = {std::fmin(lo.x, b.lo.x), std::fmin(lo.y, b.lo.y), std::fmin(lo.z, b.lo.z)};
        hi = {std::fmax(hi.x, b.hi.x), std::fmax(hi.y, b.hi.y), std::fmax(hi.z, b.hi.z)};
    }
    Vec3 extent() const { return vsub(hi, lo); }
    Vec3 centre() const { return vadd(vscale(extent(), 0.5f), lo); }
    // AABB::surface_area: sum over axes of the product of the two other extents ("half area")
    float half_area() const {
        Vec3 d = extent();
        float a = (1.0f * d.y) * d.z, b = (1.0f * d.x) * d.z, c = (1.0f * d.x) * d.y;
        return ((0.0f + a) + b) + c;
    }
    // compute_aabb / compute_aabb_tri: pad degenerate axes by EPSILON (src/geometry.rs:430-437)
    void pad_degenerate(float eps) {
        Vec3 s = extent();
        for (int k = 0; k < 3; k++)
            if (s.get(k) < eps) { hi.set(k, hi.get(k) + eps); lo.set(k, lo.get(k) - eps); }
    }
};

// cgmath Matrix4<f32>, column-major: m[col][row]
struct Mat4 {
    float m[4][4];
    static Mat4 identity() {
        Mat4 r{};
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
        return r;
    }
    static Mat4 scale(float x, float y, float z) { Mat4 r = identity(); r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z; return r; }
    static Mat4 translate(float x, float y, float z) { Mat4 r = identity(); r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z; return r; }
    static Mat4 from_cols(const float* p) { Mat4 r; for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) r.m[c][k] = p[4 * c + k]; return r; }
    void to_cols(float* p) const { for (int c = 0; c < 4; c++) for (int k = 0; k < 4; k++) p[4 * c + k] = m[c][k]; }
    // Matrix4 * Vector4 = c0*x + c1*y + c2*z + c3*w
    void apply(const float v[4], float out[4]) const {
        for (int k = 0; k < 4; k++) out[k] = ((m[0][k] * v[0] + m[1][k] * v[1]) + m[2][k] * v[2]) + m[3][k] * v[3];
    }
    Mat4 times(const Mat4& r) const { Mat4 o; for (int c = 0; c < 4; c++) apply(r.m[c], o.m[c]); return o; }
    Vec3 xform_point(Vec3 p) const {
        float v[4] = {p.x, p.y, p.z, 1.0f}, o[4];
        apply(v, o);
        float inv = 1.0f / o[3];
        return {o[0] * inv, o[1] * inv, o[2] * inv};
    }
    Vec3 xform_vector(Vec3 d) const {
        float v[4] = {d.x, d.y, d.z, 0.0f}, o[4];
        apply(v, o);
        return {o[0], o[1], o[2]};
    }
    // general inverse by cofactors of the transpose (cgmath's Matrix4::invert structure)
    bool inverse(Mat4* out) const {
        float t[4][4];  // transpose, t[col][row]
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) t[c][r] = m[r][c];
        auto cof = [&](int i, int j) {
            float s[3][3];
            int ci = 0;
            for (int c = 0; c < 4; c++) {
                if (c == i) continue;
                int ri = 0;
                for (int r = 0; r < 4; r++) { if (r == j) continue; s[ci][ri++] = t[c][r]; }
                ci++;
            }
            float d = s[0][0] * (s[1][1] * s[2][2] - s[2][1] * s[1][2]) - s[1][0] * (s[0][1] * s[2][2] - s[2][1] * s[0][2]) +
                      s[2][0] * (s[0][1] * s[1][2] - s[1][1] * s[0][2]);
            return ((i + j) & 1) ? -d : d;
        };
        float det = 0.0f;
        for (int j = 0; j < 4; j++) det += m[j][0] * cof(0, j) * 1.0f;
        if (det == 0.0f) return false;
        float inv_det = 1.0f / det;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out->m[i][j] = cof(i, j) * inv_det;
        return true;
    }
};

// cgmath::perspective(fovy, aspect, near, far)
inline Mat4 perspective(float fovy_rad, float aspect, float n, float f) {
    float c = 1.0f / std::tan(fovy_rad / 2.0f);
    Mat4 r{};
    r.m[0][0] = c / aspect;
    r.m[1][1] = c;
    r.m[2][2] = (f + n) / (n - f);
    r.m[2][3] = -1.0f;
    r.m[3][2] = (2.0f * f * n) / (n - f);
    return r;
}

// rand 0.8.5 SmallRng (Xoshiro256++) + rand_core 0.6.4 seed_from_u64 — host copy used for the
// master sampler and the per-block seeds (src/samplers/independent.rs, src/integrators/mod.rs:357-371)
struct Xoshiro {
    uint64_t s[4];
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    float next_f32() { return (float)((uint32_t)(next() >> 32) >> 8) * (1.0f / 16777216.0f); }
    void seed(uint64_t state, int variant) {
        if (variant == 1) {
            for (int i = 0; i < 4; i++) {
                state += 0x9e3779b97f4a7c15ull;
                uint64_t z = state;
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
                s[i] = z ^ (z >> 31);
            }
            return;
        }
        uint32_t w[8];
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            w[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
        for (int i = 0; i < 4; i++) s[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        if ((s[0] | s[1] | s[2] | s[3]) == 0) seed(0, variant);
    }
};

}  // namespace rl
