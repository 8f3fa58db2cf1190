// detmath_shared.h — the part of the deterministic transcendental recipe that host code needs too (the environment
// map's sin(theta) row weights, emitter.rs:340-353, and the light tree's cone algebra, emitter.rs:783-898, are built on
// the host): sin / cos / asin / acos / atan2 evaluated in f64 with +,-,*,/ only and rounded once to f32.  Compiled by hipcc (host + device) and by g++ (host); both builds pass
// -ffp-contract=off, so the three instantiations agree bit for bit with each other and with oracle/detmath.h.
#pragma once
#if defined(__HIPCC__)
#define RL_HD __host__ __device__ __forceinline__
#else
#define RL_HD inline
#endif

namespace rl {
namespace dm {
RL_HD double bits_f64(unsigned long long u) { return __builtin_bit_cast(double, u); }
RL_HD unsigned long long f64_bits(double d) { return __builtin_bit_cast(unsigned long long, d); }

RL_HD double k_sin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S5 + z * S6;
    p = S4 + z * p; p = S3 + z * p; p = S2 + z * p; p = S1 + z * p;
    return r + (r * z) * p;
}
RL_HD double k_cos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = C5 + z * C6;
    p = C4 + z * p; p = C3 + z * p; p = C2 + z * p; p = C1 + z * p;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
RL_HD void sincos_d(double x, double* s, double* c) {
    const double INV_PIO2 = 6.36619772367581382433e-01, PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11;
    double kd = __builtin_floor(x * INV_PIO2 + 0.5);
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    long long k = (long long)kd;
    double sr = k_sin(r), cr = k_cos(r);
    int q = (int)(k & 3);
    *s = q == 0 ? sr : (q == 1 ? cr : (q == 2 ? -sr : -cr));
    *c = q == 0 ? cr : (q == 1 ? -sr : (q == 2 ? -cr : sr));
}
RL_HD void sincosf_det(float x, float* s, float* c) {
    if (!(x - x == 0.0f)) { *s = *c = x - x; return; }
    double sd, cd;
    sincos_d((double)x, &sd, &cd);
    *s = (float)sd; *c = (float)cd;
}
RL_HD float sinf_det(float x) { float s, c; sincosf_det(x, &s, &c); return s; }
RL_HD float cosf_det(float x) { float s, c; sincosf_det(x, &s, &c); return c; }
// atan on f64: argument reduction to |t| <= tan(pi/16) + odd series up to t^25; used by acos / atan2
RL_HD double atan_d(double x) {
    bool neg = x < 0.0; if (neg) x = -x;
    bool inv = x > 1.0; if (inv) x = 1.0 / x;
    const double C = 0.41421356237309503;        // tan(pi/8)
    double base = 0.0, t = x;
    if (x > 0.66817863791929890) { base = 0.78539816339744827900; t = (x - 1.0) / (1.0 + x); }
    else if (x > 0.19891236737965800) { base = 0.39269908169872413950; t = (x - C) / (1.0 + x * C); }
    bool red = base != 0.0;
    double z = t * t;
    double p = 1.0 / 25.0;
    p = 1.0 / 23.0 - z * p; p = 1.0 / 21.0 - z * p; p = 1.0 / 19.0 - z * p; p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p; p = 1.0 / 13.0 - z * p; p = 1.0 / 11.0 - z * p; p = 1.0 / 9.0 - z * p;
    p = 1.0 / 7.0 - z * p; p = 1.0 / 5.0 - z * p; p = 1.0 / 3.0 - z * p; p = 1.0 - z * p;
    double r = t * p;
    if (red) r = base + r;
    if (inv) r = 1.57079632679489655800 - r;
    return neg ? -r : r;
}
// sqrt on f64 from the correctly rounded f32 square root and two Newton steps (+,-,*,/ only: no reliance on how
// the f64 sqrt instruction rounds)
RL_HD double sqrt_d(double a) {
    if (!(a > 0.0)) return 0.0;
    double s = (double)__builtin_sqrtf((float)a);
    s = 0.5 * (s + a / s);
    s = 0.5 * (s + a / s);
    return s;
}
RL_HD float atan2f_det(float y, float x) {
    if (x != x || y != y) return x + y;
    const double PI = 3.14159265358979311600;
    double yd = y, xd = x;
    if (xd == 0.0 && yd == 0.0) return __builtin_signbit(x) ? (__builtin_signbit(y) ? (float)-PI : (float)PI) : y;
    double r;
    if (__builtin_fabs(xd) >= __builtin_fabs(yd)) {
        r = atan_d(yd / xd);
        if (xd < 0.0) r = (yd >= 0.0 && !__builtin_signbit(y)) ? r + PI : r - PI;
    } else {
        r = atan_d(xd / yd);
        r = (yd > 0.0 ? 0.5 * PI : -0.5 * PI) - r;
    }
    return (float)r;
}
RL_HD float acosf_det(float x) {
    if (x != x) return x;
    if (x > 1.0f || x < -1.0f) return __builtin_bit_cast(float, 0x7fc00000u);
    double xd = x;
    double s = sqrt_d((1.0 - xd) * (1.0 + xd));
    const double PI = 3.14159265358979311600;
    double r;
    if (__builtin_fabs(xd) >= s) { r = atan_d(s / xd); if (xd < 0.0) r = r + PI; }   // acos(x) = atan2(sqrt(1 - x^2), x)
    else { r = 0.5 * PI - atan_d(xd / s); }
    return (float)r;
}
RL_HD float asinf_det(float x) {
    if (x != x) return x;
    if (x > 1.0f || x < -1.0f) return __builtin_bit_cast(float, 0x7fc00000u);
    double xd = x;
    double c = sqrt_d((1.0 - xd) * (1.0 + xd));      // asin(x) = atan2(x, sqrt(1 - x^2))
    const double PI = 3.14159265358979311600;
    double r;
    if (c >= __builtin_fabs(xd)) r = atan_d(xd / c);
    else { r = atan_d(c / xd); r = (xd > 0.0 ? 0.5 * PI : -0.5 * PI) - r; }
    return (float)r;
}
}  // namespace dm
}  // namespace rl
