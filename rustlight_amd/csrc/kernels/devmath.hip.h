// devmath.hip.h — device-side f32 vector / colour algebra, deterministic transcendentals and the
// Xoshiro256++ sampler for the gfx950 wavefront path tracer.
//
// Numerics contract: every f32 operation on the path is a single IEEE-754
// operation in the order rustlight's Rust code performs it — no FMA contraction (the build
// passes -ffp-contract=off and this header pins it again), correctly rounded divide / sqrt,
// denormals kept.  Transcendentals are evaluated in f64 with +,-,*,/ only and rounded once to
// f32, so the CPU oracle (which restates the same recipe) and these kernels agree bit-for-bit
// and branch decisions (`q < xi`, `t < its.t`, `u + v <= 1`) never diverge.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../detmath_shared.h"

// RL_FAST_MATH (fused_*_fast.hip only): the opt-in tolerance mode `numerics = fast` — FMA contraction, v_rcp / v_rsq / v_sqrt (1 ulp) instead of
// the correctly rounded sequences, hardware sin / cos / exp2 / log2.  The RNG and every integer operation are untouched (the draw sequence of
// a path stays bit-exact as long as its branch decisions do); pixels agree with the exact build within BASELINE.json's tolerance, not bit for bit.
#ifdef RL_FAST_MATH
#pragma clang fp contract(fast)
#define RL_NUMERICS_ID 1
#else
#pragma clang fp contract(off)
#define RL_NUMERICS_ID 0
#endif

#define RL_DEV __device__ __forceinline__

namespace rl {

static constexpr float kEps = 0.0001f;                  // constants::EPSILON (src/lib.rs:51)
static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.318309886183790671538f;
static constexpr float kPi2 = 1.57079632679489661923f;
static constexpr float kPi4 = 0.785398163397448309616f;
static constexpr float kF32Max = 3.402823466e+38f;

RL_DEV float f32_inf() { return __int_as_float(0x7f800000); }
RL_DEV float f32_nan() { return __int_as_float(0x7fc00000); }
// is_finite: a class test, NOT `x - x == 0`: when x is itself a product, FMA contraction (tolerance build) may fuse that idiom into
// fma(a, b, -x) — the rounding error of the product, non-zero — and every guarded `Color * f32` then silently returns black (this cost the
// glossy scenes 13 % of their energy in the first fast build)
RL_DEV bool finite_f(float x) { return __builtin_isfinite(x); }
RL_DEV float rmax(float a, float b) { return fmaxf(a, b); }     // Rust f32::max (non-NaN operand wins)
RL_DEV float rmin(float a, float b) { return fminf(a, b); }
RL_DEV float signum_f(float x) { return x != x ? x : copysignf(1.0f, x); }   // Rust f32::signum
#ifdef RL_FAST_MATH
RL_DEV float sqrt_rn(float x) { return __builtin_amdgcn_sqrtf(x); }              // v_sqrt_f32, 1 ulp
RL_DEV float div_rn(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }   // v_rcp_f32 (1 ulp) + one multiply
#else
RL_DEV float sqrt_rn(float x) { return __builtin_sqrtf(x); }   // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, the default); __fsqrt_rn is the 1-ulp native sqrt
RL_DEV float div_rn(float a, float b) { return a / b; }             // correctly rounded under the same default
#endif

// ------------------------------------------------------------------------------------------
struct V2 { float x, y; };
struct V3 { float x, y, z; };
RL_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RL_DEV V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RL_DEV V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RL_DEV V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
RL_DEV V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RL_DEV V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
#ifdef RL_FAST_MATH
RL_DEV V3 operator/(V3 a, float s) { const float r = __builtin_amdgcn_rcpf(s); return mk3(a.x * r, a.y * r, a.z * r); }
#else
RL_DEV V3 operator/(V3 a, float s) { return mk3(div_rn(a.x, s), div_rn(a.y, s), div_rn(a.z, s)); }
#endif
RL_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RL_DEV V3 cross(V3 a, V3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RL_DEV float length2(V3 a) { return dot(a, a); }
RL_DEV float length(V3 a) { return sqrt_rn(dot(a, a)); }
#ifdef RL_FAST_MATH
RL_DEV V3 normalize(V3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }
#else
RL_DEV V3 normalize(V3 a) { return a * div_rn(1.0f, length(a)); }   // cgmath: v * (1 / |v|)
#endif

// Color with rustlight's guards (src/structure.rs:249-303)
struct Col { float r, g, b; };
RL_DEV Col mkc(float r, float g, float b) { Col c; c.r = r; c.g = g; c.b = b; return c; }
RL_DEV Col czero() { return mkc(0.0f, 0.0f, 0.0f); }
RL_DEV Col cone() { return mkc(1.0f, 1.0f, 1.0f); }
RL_DEV Col cval(float v) { return mkc(v, v, v); }
RL_DEV bool is_zero(Col c) { return c.r == 0.0f && c.g == 0.0f && c.b == 0.0f; }
RL_DEV float channel_max(Col c) { return rmax(c.r, rmax(c.g, c.b)); }
RL_DEV float cavg(Col c) { return div_rn(c.r + c.g + c.b, 3.0f); }
RL_DEV Col operator+(Col a, Col b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
RL_DEV Col operator-(Col a, Col b) { return mkc(a.r - b.r, a.g - b.g, a.b - b.b); }
RL_DEV Col operator-(Col a) { return mkc(-a.r, -a.g, -a.b); }
RL_DEV Col operator*(Col a, Col b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
RL_DEV Col operator/(Col a, Col b) { return mkc(div_rn(a.r, b.r), div_rn(a.g, b.g), div_rn(a.b, b.b)); }
RL_DEV Col operator*(Col a, float s) { return finite_f(s) ? mkc(a.r * s, a.g * s, a.b * s) : czero(); }   // Mul<f32> for Color
RL_DEV Col operator*(float s, Col a) { return mkc(a.r * s, a.g * s, a.b * s); }                            // Mul<Color> for f32
RL_DEV Col operator/(Col a, float s) {                                                                     // Div<f32> for Color
    return (s == 0.0f || !finite_f(s)) ? czero() : mkc(div_rn(a.r, s), div_rn(a.g, s), div_rn(a.b, s));
}
RL_DEV Col div_unguarded(Col a, float s) { return mkc(div_rn(a.r, s), div_rn(a.g, s), div_rn(a.b, s)); }   // DivAssign<f32>
RL_DEV Col scale_unguarded(Col a, float s) { return mkc(a.r * s, a.g * s, a.b * s); }                       // Scale<f32>
RL_DEV Col safe_sqrt(Col c) { return mkc(sqrt_rn(rmax(c.r, 0.0f)), sqrt_rn(rmax(c.g, 0.0f)), sqrt_rn(rmax(c.b, 0.0f))); }
RL_DEV float cget(Col c, int i) { return i == 0 ? c.r : (i == 1 ? c.g : c.b); }

// ------------------------------------------------------------------------------------------
// deterministic transcendentals (f64 evaluation, one rounding to f32)
namespace dm {
// bits_f64 / f64_bits / sincosf_det / sinf_det / cosf_det: ../detmath_shared.h

RL_DEV double exp_d(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10, INV_LN2 = 1.44269504088896338700e+00;
    if (x != x) return x;
    if (x > 709.0) return bits_f64(0x7ff0000000000000ull);
    if (x < -745.0) return 0.0;
    double kd = floor(x * INV_LN2 + 0.5);
    double r = (x - kd * LN2_HI) - kd * LN2_LO;
    double p = 1.0 / 6227020800.0;
    p = 1.0 / 479001600.0 + r * p; p = 1.0 / 39916800.0 + r * p; p = 1.0 / 3628800.0 + r * p;
    p = 1.0 / 362880.0 + r * p; p = 1.0 / 40320.0 + r * p; p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p; p = 1.0 / 120.0 + r * p; p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p; p = 0.5 + r * p; p = 1.0 + r * p; p = 1.0 + r * p;
    long long k = (long long)kd;
    long long k1 = k / 2, k2 = k - k1;
    double s1 = bits_f64((unsigned long long)(k1 + 1023) << 52);
    double s2 = bits_f64((unsigned long long)(k2 + 1023) << 52);
    return (p * s1) * s2;
}
RL_DEV double log_d(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    unsigned long long b = f64_bits(x);
    long long e = (long long)((b >> 52) & 0x7ff);
    if (e == 0) { x = x * 18014398509481984.0; b = f64_bits(x); e = (long long)((b >> 52) & 0x7ff) - 54; }
    e -= 1023;
    double m = bits_f64((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = 1.0 / 21.0 + z * p; p = 1.0 / 19.0 + z * p; p = 1.0 / 17.0 + z * p; p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p; p = 1.0 / 11.0 + z * p; p = 1.0 / 9.0 + z * p; p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p; p = 1.0 / 3.0 + z * p; p = 1.0 + z * p;
    double lm = (2.0 * s) * p;
    double ed = (double)e;
    return (ed * LN2_HI + lm) + ed * LN2_LO;
}
RL_DEV float expf_det(float x) {
    if (x != x) return x;
    if (x > 89.0f) return f32_inf();
    if (x < -104.0f) return 0.0f;
    return (float)exp_d((double)x);
}
RL_DEV float logf_det(float x) {
    if (x != x) return x;
    if (x < 0.0f) return f32_nan();
    if (x == 0.0f) return -f32_inf();
    if (x - x != 0.0f) return x;
    return (float)log_d((double)x);
}
RL_DEV float powf_det(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return f32_nan();
    if (x == 0.0f) return y > 0.0f ? 0.0f : f32_inf();
    if (x - x != 0.0f) return y > 0.0f ? x : 0.0f;
    double a = (double)y * log_d((double)x);
    return (float)exp_d(a);
}
// atan_d / sqrt_d / atan2f_det / acosf_det / asinf_det: ../detmath_shared.h (the host builds the light tree with them)
}  // namespace dm

// transcendentals as the stage functions call them: the deterministic f64 recipes above, or the hardware units in the tolerance build
#ifdef RL_FAST_MATH
RL_DEV void m_sincosf(float x, float* s, float* c) { const float r = x * 0.15915494309189535f; *s = __builtin_amdgcn_sinf(r); *c = __builtin_amdgcn_cosf(r); }
RL_DEV float m_sinf(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }
RL_DEV float m_cosf(float x) { return __builtin_amdgcn_cosf(x * 0.15915494309189535f); }
RL_DEV float m_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
RL_DEV float m_logf(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
RL_DEV float m_powf(float x, float y) {
    if (y == 0.0f || x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return f32_nan();
    if (x == 0.0f) return y > 0.0f ? 0.0f : f32_inf();
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
#else
RL_DEV void m_sincosf(float x, float* s, float* c) { dm::sincosf_det(x, s, c); }
RL_DEV float m_sinf(float x) { return dm::sinf_det(x); }
RL_DEV float m_cosf(float x) { return dm::cosf_det(x); }
RL_DEV float m_expf(float x) { return dm::expf_det(x); }
RL_DEV float m_logf(float x) { return dm::logf_det(x); }
RL_DEV float m_powf(float x, float y) { return dm::powf_det(x, y); }
#endif

// f32::powi = llvm.powi (binary exponentiation, compiler-rt __powisf2)
RL_DEV float powi_f(float a, int b) {
    float r = 1.0f;
    for (;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
    return r;
}

// ------------------------------------------------------------------------------------------
// rand 0.8.5 SmallRng = Xoshiro256++ (src/samplers/independent.rs:5-22)
struct Rng { unsigned long long s0, s1, s2, s3; };
RL_DEV unsigned long long rotl64(unsigned long long x, int k) { return (x << k) | (x >> (64 - k)); }
RL_DEV unsigned long long rng_next_u64(Rng& r) {
    unsigned long long result = rotl64(r.s0 + r.s3, 23) + r.s0;
    unsigned long long t = r.s1 << 17;
    r.s2 ^= r.s0; r.s3 ^= r.s1; r.s1 ^= r.s2; r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return result;
}
// Sampler::next(): Standard f32 = (next_u32() >> 8) * 2^-24, next_u32 = next_u64() >> 32
RL_DEV float rng_next_f32(Rng& r) {
    unsigned int v = (unsigned int)(rng_next_u64(r) >> 32);
    return (float)(v >> 8) * (1.0f / 16777216.0f);
}
// SeedableRng::seed_from_u64 (rand_core 0.6.4 PCG32 fill; variant 1 = SplitMix64)
RL_DEV Rng rng_seed(unsigned long long state, int variant) {
    Rng r;
    if (variant == 1) {
        unsigned long long o[4];
        for (int i = 0; i < 4; i++) {
            state += 0x9e3779b97f4a7c15ull;
            unsigned long long z = state;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            o[i] = z ^ (z >> 31);
        }
        r.s0 = o[0]; r.s1 = o[1]; r.s2 = o[2]; r.s3 = o[3];
        return r;
    }
    unsigned int w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        unsigned int xs = (unsigned int)(((state >> 18) ^ state) >> 27);
        unsigned int rot = (unsigned int)(state >> 59);
        w[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    r.s0 = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    r.s1 = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
    r.s2 = (unsigned long long)w[4] | ((unsigned long long)w[5] << 32);
    r.s3 = (unsigned long long)w[6] | ((unsigned long long)w[7] << 32);
    if ((r.s0 | r.s1 | r.s2 | r.s3) == 0ull) {   // Xoshiro256PlusPlus::from_seed: all-zero seed => seed_from_u64(0)
        r.s0 = 0x45cdb581f973f2ecull; r.s1 = 0xad6cad067346f087ull; r.s2 = 0x67e71733e3a3d0d0ull; r.s3 = 0xfe7d8ad772ea9bf2ull;
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// sampling math (src/math.rs:37-72, 388-394) and the Pixar frame (src/math.rs:357-384)
RL_DEV V2 concentric_sample_disk(V2 u) {
    V2 o; o.x = u.x * 2.0f - 1.0f; o.y = u.y * 2.0f - 1.0f;
    V2 r0; r0.x = 0.0f; r0.y = 0.0f;
    if (o.x == 0.0f && o.y == 0.0f) return r0;
    float theta, r;
    if (fabsf(o.x) > fabsf(o.y)) { r = o.x; theta = kPi4 * div_rn(o.y, o.x); }
    else { r = o.y; theta = kPi2 - kPi4 * div_rn(o.x, o.y); }
    float s, c;
    m_sincosf(theta, &s, &c);
    V2 out; out.x = c * r; out.y = s * r;
    return out;
}
RL_DEV V3 cosine_sample_hemisphere(V2 u) {
    V2 d = concentric_sample_disk(u);
    float z = sqrt_rn(rmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return mk3(d.x, d.y, z);
}
RL_DEV V3 sample_uniform_sphere(V2 u) {
    float z = 1.0f - 2.0f * u.x;
    float r = sqrt_rn(rmax(1.0f - z * z, 0.0f));
    float phi = 2.0f * kPi * u.y;
    float s, c;
    m_sincosf(phi, &s, &c);
    return mk3(r * c, r * s, z);
}
RL_DEV V2 uniform_sample_triangle(V2 u) {
    float su0 = sqrt_rn(u.x);
    V2 b; b.x = 1.0f - su0; b.y = u.y * su0;
    return b;
}

struct Frame { V3 x, y, z; };
RL_DEV Frame make_frame(V3 n) {
    float sign = signum_f(n.z);
    float a = div_rn(-1.0f, sign + n.z);
    float b = n.x * n.y * a;
    Frame f;
    f.x = mk3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    f.y = mk3(b, sign + n.y * n.y * a, -n.y);
    f.z = n;
    return f;
}
RL_DEV V3 to_world(const Frame& f, V3 v) { return f.x * v.x + f.y * v.y + f.z * v.z; }
RL_DEV V3 to_local(const Frame& f, V3 v) { return mk3(dot(v, f.x), dot(v, f.y), dot(v, f.z)); }

}  // namespace rl
