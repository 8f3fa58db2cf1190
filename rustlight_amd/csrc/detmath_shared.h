// detmath_shared.h — the part of the deterministic transcendental recipe that host code needs too (the environment
// map's sin(theta) row weights are built on the host, emitter.rs:340-353): sin / cos evaluated in f64 with +,-,*,/
// only and rounded once to f32.  Compiled by hipcc (host + device) and by g++ (host); both builds pass
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
}  // namespace dm
}  // namespace rl
