// oracle/detmath.h — TEST INFRASTRUCTURE (part of the CPU oracle; never linked into the product).
//
// Deterministic f32 transcendentals used by the oracle wherever rustlight calls Rust's
// f32::{sin,cos,sin_cos,exp,ln,powf,acos,atan2} (e.g. src/math.rs:37-72, src/volume.rs:95-141,
// src/bsdfs/phong.rs:27-30, src/bsdfs/distribution.rs:62-108).
//
// Why not libm: Rust forwards these to the platform libm, whose results are *not* specified
// bit-for-bit (Rust std docs: "precision is unspecified"), and the GPU has a different libm
// (ocml).  A 1-ulp difference in cos() flips rare branch decisions and changes whole paths
// (SURVEY.md H2).  The oracle therefore evaluates every transcendental in f64 with only
// IEEE-754 +,-,*,/ and a final f64->f32 rounding; the HIP kernels restate the same sequence of
// operations (rustlight_amd/csrc/kernels/detmath.hip.h), so oracle and GPU agree bit-for-bit.
// Each result is the correctly rounded f32 value except when the exact value lies within
// ~1e-15 (relative) of a rounding boundary; tests/test_oracle_math.py measures the distance to
// glibc's sinf/cosf/expf/logf/powf (<= 1 ulp, >99.9 % identical).
//
// Compile with -ffp-contract=off (no FMA contraction) — the Makefile does.
#pragma once
// ORC_TIMING_BUILD (Makefile target librl_oracle_timing.so): the entry points below forward to libm — that build is only
// ever timed as the CPU baseline of bench.py (-O3, native float transcendentals), never used as the checker.
#include <cmath>
#include <cstdint>
#include <cstring>

namespace detmath {

static inline double bits_to_f64(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }
static inline uint64_t f64_to_bits(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }

// fdlibm argument-reduction constants (pi/2 split into a 33-bit head and a tail).
static const double INV_PIO2 = 6.36619772367581382433e-01;
static const double PIO2_HI = 1.57079632673412561417e+00;
static const double PIO2_LO = 6.07710050650619224932e-11;

static inline double k_sin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S5 + z * S6;
    p = S4 + z * p;
    p = S3 + z * p;
    p = S2 + z * p;
    p = S1 + z * p;
    return r + (r * z) * p;
}
static inline double k_cos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = C5 + z * C6;
    p = C4 + z * p;
    p = C3 + z * p;
    p = C2 + z * p;
    p = C1 + z * p;
    return (1.0 - 0.5 * z) + (z * z) * p;
}

// (sin x, cos x) in f64 for a finite f32-ranged argument.
static inline void sincos_d(double x, double* s, double* c) {
    double kd = std::floor(x * INV_PIO2 + 0.5);
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    long long k = (long long)kd;
    double sr = k_sin(r), cr = k_cos(r);
    switch ((int)(k & 3)) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
static inline float sinf_det(float x) {
#ifdef ORC_TIMING_BUILD
    return std::sin(x);
#endif

    if (!(x - x == 0.0f)) return x - x;  // NaN / inf -> NaN
    double s, c; sincos_d((double)x, &s, &c); return (float)s;
}
static inline float cosf_det(float x) {
#ifdef ORC_TIMING_BUILD
    return std::cos(x);
#endif

    if (!(x - x == 0.0f)) return x - x;
    double s, c; sincos_d((double)x, &s, &c); return (float)c;
}
static inline void sincosf_det(float x, float* s, float* c) {
#ifdef ORC_TIMING_BUILD
    *s = std::sin(x); *c = std::cos(x); return;
#endif

    if (!(x - x == 0.0f)) { *s = *c = x - x; return; }
    double sd, cd; sincos_d((double)x, &sd, &cd); *s = (float)sd; *c = (float)cd;
}

static const double LN2_HI = 6.93147180369123816490e-01;
static const double LN2_LO = 1.90821492927058770002e-10;
static const double INV_LN2 = 1.44269504088896338700e+00;

// e^x in f64, |x| <= 745.
static inline double exp_d(double x) {
    if (x != x) return x;
    if (x > 709.0) return bits_to_f64(0x7ff0000000000000ull);
    if (x < -745.0) return 0.0;
    double kd = std::floor(x * INV_LN2 + 0.5);
    double r = (x - kd * LN2_HI) - kd * LN2_LO;
    // Taylor to degree 13 on |r| <= 0.347 (remainder < 4e-18)
    double p = 1.0 / 6227020800.0;
    p = 1.0 / 479001600.0 + r * p;
    p = 1.0 / 39916800.0 + r * p;
    p = 1.0 / 3628800.0 + r * p;
    p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p;
    p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p;
    p = 1.0 / 120.0 + r * p;
    p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p;
    p = 0.5 + r * p;
    p = 1.0 + r * p;
    p = 1.0 + r * p;
    long long k = (long long)kd;
    // scale by 2^k in two steps so that results in the f64 subnormal range stay finite/correct
    long long k1 = k / 2, k2 = k - k1;
    double s1 = bits_to_f64((uint64_t)(k1 + 1023) << 52);
    double s2 = bits_to_f64((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}
// ln x in f64 for x > 0 finite.
static inline double log_d(double x) {
    uint64_t b = f64_to_bits(x);
    long long e = (long long)((b >> 52) & 0x7ff);
    if (e == 0) {  // f64 subnormal (cannot come from an f32 input, kept for completeness)
        x = x * 18014398509481984.0;  // 2^54
        b = f64_to_bits(x);
        e = (long long)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    double m = bits_to_f64((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = 1.0 / 21.0 + z * p;
    p = 1.0 / 19.0 + z * p;
    p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    p = 1.0 + z * p;
    double lm = (2.0 * s) * p;
    double ed = (double)e;
    return (ed * LN2_HI + lm) + ed * LN2_LO;
}
static inline float expf_det(float x) {
#ifdef ORC_TIMING_BUILD
    return std::exp(x);
#endif

    if (x != x) return x;
    if (x > 89.0f) return (float)bits_to_f64(0x7ff0000000000000ull);
    if (x < -104.0f) return 0.0f;
    return (float)exp_d((double)x);
}
static inline float logf_det(float x) {
#ifdef ORC_TIMING_BUILD
    return std::log(x);
#endif

    if (x != x) return x;
    if (x < 0.0f) return (float)bits_to_f64(0x7ff8000000000000ull);
    if (x == 0.0f) return -(float)bits_to_f64(0x7ff0000000000000ull);
    if (x - x != 0.0f) return x;  // +inf
    return (float)log_d((double)x);
}
// powf for the cases rustlight produces (x >= 0; see phong.rs:27-30,81-83,107-110).
static inline float powf_det(float x, float y) {
#ifdef ORC_TIMING_BUILD
    return std::pow(x, y);
#endif

    if (y == 0.0f) return 1.0f;
    if (x == 1.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return (float)bits_to_f64(0x7ff8000000000000ull);  // non-integer exponents only on this path
    if (x == 0.0f) return y > 0.0f ? 0.0f : (float)bits_to_f64(0x7ff0000000000000ull);
    if (x - x != 0.0f) return y > 0.0f ? x : 0.0f;  // x = +inf
    double a = (double)y * log_d((double)x);
    return (float)exp_d(a);
}

// atan on f64 via argument reduction to [0, tan(pi/8)] + odd series; used by acos/atan2.
static inline double atan_d(double x) {
    bool neg = x < 0.0; if (neg) x = -x;
    bool inv = x > 1.0; if (inv) x = 1.0 / x;
    // second reduction: atan(x) = atan(c) + atan((x-c)/(1+x*c)), c in {0, tan(pi/8), 1}
    // so that |t| <= tan(pi/16) = 0.19891
    const double C = 0.41421356237309503;        // tan(pi/8)
    double base = 0.0, t = x;
    if (x > 0.66817863791929890) { base = 0.78539816339744827900; t = (x - 1.0) / (1.0 + x); }
    else if (x > 0.19891236737965800) { base = 0.39269908169872413950; t = (x - C) / (1.0 + x * C); }
    bool red = base != 0.0;
    const double ATAN_C = base;
    double z = t * t;
    // odd series up to t^25 : t^27/27 ~ 1e-19
    double p = 1.0 / 25.0;
    p = 1.0 / 23.0 - z * p;
    p = 1.0 / 21.0 - z * p;
    p = 1.0 / 19.0 - z * p;
    p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p;
    p = 1.0 / 13.0 - z * p;
    p = 1.0 / 11.0 - z * p;
    p = 1.0 / 9.0 - z * p;
    p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p;
    p = 1.0 / 3.0 - z * p;
    p = 1.0 - z * p;
    double r = t * p;
    if (red) r = ATAN_C + r;
    if (inv) r = 1.57079632679489655800 - r;
    return neg ? -r : r;
}
static inline float atan2f_det(float y, float x) {
#ifdef ORC_TIMING_BUILD
    return std::atan2(y, x);
#endif

    if (x != x || y != y) return x + y;
    const double PI = 3.14159265358979311600;
    double yd = y, xd = x;
    if (xd == 0.0 && yd == 0.0) return std::signbit(x) ? (std::signbit(y) ? (float)-PI : (float)PI) : y;
    double r;
    if (std::fabs(xd) >= std::fabs(yd)) {
        r = atan_d(yd / xd);
        if (xd < 0.0) r = (yd >= 0.0 && !std::signbit(y)) ? r + PI : r - PI;
    } else {
        r = atan_d(xd / yd);
        r = (yd > 0.0 ? 0.5 * PI : -0.5 * PI) - r;
    }
    return (float)r;
}
// sqrt on f64 from the correctly rounded f32 square root and two Newton steps (+,-,*,/ only, so the GPU's f64 sqrt
// rounding never enters)
static inline double sqrt_d(double a) {
    if (!(a > 0.0)) return 0.0;
    double s = (double)std::sqrt((float)a);
    s = 0.5 * (s + a / s);
    s = 0.5 * (s + a / s);
    return s;
}
static inline float asinf_det(float x);
static inline float acosf_det(float x) {
#ifdef ORC_TIMING_BUILD
    return std::acos(x);
#endif

    if (x != x) return x;
    if (x > 1.0f || x < -1.0f) return (float)bits_to_f64(0x7ff8000000000000ull);
    double xd = x;
    double s = sqrt_d((1.0 - xd) * (1.0 + xd));
    // acos(x) = atan2(sqrt(1-x^2), x)
    const double PI = 3.14159265358979311600;
    double r;
    if (std::fabs(xd) >= s) { r = atan_d(s / xd); if (xd < 0.0) r = r + PI; }
    else { r = 0.5 * PI - atan_d(xd / s); }
    return (float)r;
}

static inline float asinf_det(float x) {
#ifdef ORC_TIMING_BUILD
    return std::asin(x);
#endif
    if (x != x) return x;
    if (x > 1.0f || x < -1.0f) return (float)bits_to_f64(0x7ff8000000000000ull);
    double xd = x;
    double c = sqrt_d((1.0 - xd) * (1.0 + xd));      // asin(x) = atan2(x, sqrt(1 - x^2))
    const double PI = 3.14159265358979311600;
    double r;
    if (c >= std::fabs(xd)) r = atan_d(xd / c);
    else { r = atan_d(c / xd); r = (xd > 0.0 ? 0.5 * PI : -0.5 * PI) - r; }
    return (float)r;
}
}  // namespace detmath
