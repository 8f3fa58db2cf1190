// oracle/rl_oracle.cpp — TEST INFRASTRUCTURE.  NOT part of the product.
//
// CPU restatement (C++17, scalar) of rustlight's sensor-side `path` integrator, written from
// the reference sources under /root/reference (paths below are relative to that tree).  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
// only as the checker / the timed CPU baseline — never as the thing shipped.
//
// PARITY UNPINNED: the reference has no unit tests, golden vectors or fixtures for this path
//  and cannot be compiled here (no Rust toolchain).  The only externally
// published known-answer vector this oracle is pinned to is the Xoshiro256++ reference output
// (SURVEY.md App. B.1).  The SmallRng::seed_from_u64 expansion (rand 0.8.5 / rand_core 0.6.4,
// third-party, not vendored) is restated from its documented algorithm and is switchable
// between the PCG32-fill default and Xoshiro's SplitMix64 variant.
// HOW TO PIN IT: oracle/pin/README.md — with a Rust toolchain, `cargo run --release --example=pin_draws` (64 draws of
// SmallRng::seed_from_u64(0), oracle/pin/pin_draws.rs) and `cargo run --release --example=cli -- -t 1 -r independent:0 -n 4 -o ref.pfm
// oracle/pin/out/cbox_64.xml path` produce the two artefacts `python oracle/pin/diff_pin.py ref_draws.txt ref.pfm` compares with this
// oracle's expected outputs (oracle/pin/expected.json holds their hashes; tests/test_pin_kit.py keeps the kit alive).
//
// Deliberate, documented deviations from the reference:
//   * transcendentals go through oracle/detmath.h instead of the platform libm (see there);
//   * a path is cut when `generate`'s depth counter reaches 2048 (the reference loops forever on NaN
//     throughput: `rr_weight < next()` is false for NaN, directional.rs:77-85);
//   * the SAH builder sorts primitive boxes by centre with std::stable_sort and the comparator `a < b` (accel.rs:125-131: `sort_by(|a, b| if a < b {Less} else
//     {Greater})`).  For finite centres that is the order Rust's stable merge sort gives.  With a NaN centre (non-finite vertices) `<` is not a strict weak order:
//     libstdc++'s merge sort and Rust's then both return SOME permutation, not provably the same one — this oracle and the product's host builder (csrc/host/bvh.cpp,
//     same libstdc++ call) agree with each other on such meshes (tests: test_non_finite_and_degenerate_geometry_parity), nothing says they agree with rustlight;
//   * `eval_order = 1` ("forward") evaluates the same path graph front-to-back the way the GPU
//     wavefront accumulates it; `eval_order = 0` is the reference's inner-first recursion
//     (explicit/path.rs:113-184).  They differ only in f32 rounding of the final radiance.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no fast-math = the checker; a second, -O3 / libm / FMA build,
// librl_oracle_timing.so, is only ever timed as bench.py's CPU baseline).

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../include/rustlight_amd.h"  // POD descriptor structs only
#include "detmath.h"
#include "rl_oracle.h"

namespace orc {

// ------------------------------------------------------------------------------------------
// constants (src/lib.rs:50-53)
static const float EPSILON = 0.0001f;
static const float PI_F = 3.14159265358979323846f;          // std::f32::consts::PI
static const float FRAC_1_PI = 0.318309886183790671538f;    // FRAC_1_PI
static const float FRAC_PI_2 = 1.57079632679489661923f;
static const float FRAC_PI_4 = 0.785398163397448309616f;
static const float F32_MAX = FLT_MAX;
static const float F32_INF = INFINITY;
static const uint32_t ORC_DEPTH_CAP = 2048;  // `generate` depth counter at which a path is cut (see header)

// Rust f32::max / f32::min: return the non-NaN operand
static inline float rmax(float a, float b) { return std::fmax(a, b); }
static inline float rmin(float a, float b) { return std::fmin(a, b); }
// Rust f32::signum: +-1 including for +-0, NaN for NaN (used by Frame::new, src/math.rs:361)
static inline float signum(float x) { return x != x ? x : std::copysign(1.0f, x); }
static inline bool is_finite(float x) { return x - x == 0.0f; }

// ------------------------------------------------------------------------------------------
// cgmath 0.18 vectors (semantics: SURVEY.md App. E)
struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float magnitude2(V3 a) { return dot(a, a); }
static inline float magnitude(V3 a) { return std::sqrt(dot(a, a)); }
// InnerSpace::normalize = self * (1 / magnitude)
static inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }

// ------------------------------------------------------------------------------------------
// Color (src/structure.rs:105-380) — note the non-finite guards on Mul<f32> / Div<f32>
struct Color {
    float r, g, b;
    static Color zero() { return {0.f, 0.f, 0.f}; }
    static Color one() { return {1.f, 1.f, 1.f}; }
    static Color value(float v) { return {v, v, v}; }
    bool is_zero() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
    float channel_max() const { return rmax(r, rmax(g, b)); }
    float luminance() const { return r * 0.212671f + g * 0.715160f + b * 0.072169f; }   // structure.rs:173-176
    float avg() const { return (r + g + b) / 3.0f; }
    float get(int c) const { return c == 0 ? r : (c == 1 ? g : b); }
    Color safe_sqrt() const { return {std::sqrt(rmax(r, 0.f)), std::sqrt(rmax(g, 0.f)), std::sqrt(rmax(b, 0.f))}; }
    Color exp() const { return {detmath::expf_det(r), detmath::expf_det(g), detmath::expf_det(b)}; }
    void scale(float v) { r *= v; g *= v; b *= v; }   // Scale<f32>, unguarded (structure.rs:184-190)
};
static inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
static inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
static inline Color operator-(Color a) { return {-a.r, -a.g, -a.b}; }
static inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
static inline Color operator/(Color a, Color b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
// Mul<f32> for Color: zero if the scalar is not finite (structure.rs:278-292)
static inline Color operator*(Color a, float s) {
    if (is_finite(s)) return {a.r * s, a.g * s, a.b * s};
    return Color::zero();
}
// Mul<Color> for f32: unguarded (structure.rs:294-303)
static inline Color operator*(float s, Color a) { return {a.r * s, a.g * s, a.b * s}; }
// Div<f32> for Color: zero if the scalar is 0 or not finite (structure.rs:249-265)
static inline Color operator/(Color a, float s) {
    if (s == 0.0f || !is_finite(s)) return Color::zero();
    return {a.r / s, a.g / s, a.b / s};
}
static inline void div_assign(Color& a, float s) { a.r /= s; a.g /= s; a.b /= s; }  // DivAssign<f32>, unguarded
static inline void mul_assign(Color& a, Color b) { a.r *= b.r; a.g *= b.g; a.b *= b.b; }
static inline void add_assign(Color& a, Color b) { a.r += b.r; a.g += b.g; a.b += b.b; }

// ------------------------------------------------------------------------------------------
// PDF (src/structure.rs:19-94)
struct PDF {
    enum Kind { SolidAngle, Area, Discrete } kind;
    float v;
    static PDF solid_angle(float v) { return {SolidAngle, v}; }
    static PDF area(float v) { return {Area, v}; }
    static PDF discrete(float v) { return {Discrete, v}; }
    bool is_zero() const { return v == 0.0f; }
    float value() const { return v; }
    PDF mul(float o) const { return {kind, v * o}; }
    // as_solid_angle_geom (structure.rs:27-40)
    PDF as_solid_angle_geom(float g_ad) const {
        if (kind == SolidAngle) return *this;
        // Area (Discrete panics in the reference)
        if (g_ad == 0.0f) return solid_angle(0.0f);
        return solid_angle(v / g_ad);
    }
};

// ------------------------------------------------------------------------------------------
// rand 0.8.5 SmallRng = Xoshiro256++ ; rand_core 0.6.4 seed_from_u64
struct Rng {
    uint64_t s[4];
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next_u64() {
        uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    // Standard f32: (next_u32() >> 8) * 2^-24 with next_u32 = next_u64() >> 32
    float next_f32() {
        uint32_t v = (uint32_t)(next_u64() >> 32);
        return (float)(v >> 8) * (1.0f / 16777216.0f);
    }
    static Rng seed_from_u64(uint64_t state, int variant) {
        Rng r;
        if (variant == 1) {  // Xoshiro256PlusPlus::seed_from_u64: SplitMix64
            for (int i = 0; i < 4; i++) {
                state += 0x9e3779b97f4a7c15ull;
                uint64_t z = state;
                z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
                r.s[i] = z ^ (z >> 31);
            }
            return r;
        }
        // rand_core default: PCG32 output fills the 32 seed bytes in 8 little-endian chunks
        uint32_t w[8];
        for (int i = 0; i < 8; i++) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            w[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        }
        for (int i = 0; i < 4; i++) r.s[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        if ((r.s[0] | r.s[1] | r.s[2] | r.s[3]) == 0) return seed_from_u64(0, variant);  // from_seed all-zero rule
        return r;
    }
};

// IndependentSampler (src/samplers/independent.rs:5-34)
struct Sampler {
    Rng rnd;
    int variant = 0;
    uint64_t draws = 0;
    float next() { draws++; return rnd.next_f32(); }
    V2 next2d() { float x = next(); float y = next(); return {x, y}; }
    Sampler clone_box() { Sampler s; s.rnd = Rng::seed_from_u64(rnd.next_u64(), variant); s.variant = variant; return s; }
};

// ------------------------------------------------------------------------------------------
// sampling math (src/math.rs:37-72, 388-394)
static inline V2 concentric_sample_disk(V2 u) {
    V2 o = {u.x * 2.0f - 1.0f, u.y * 2.0f - 1.0f};
    if (o.x == 0.0f && o.y == 0.0f) return {0.0f, 0.0f};
    float theta, r;
    if (std::fabs(o.x) > std::fabs(o.y)) {
        r = o.x;
        theta = FRAC_PI_4 * (o.y / o.x);
    } else {
        r = o.y;
        theta = FRAC_PI_2 - FRAC_PI_4 * (o.x / o.y);
    }
    float s, c;
    detmath::sincosf_det(theta, &s, &c);
    return {c * r, s * r};
}
static inline V3 cosine_sample_hemisphere(V2 u) {
    V2 d = concentric_sample_disk(u);
    float z = std::sqrt(rmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
static inline V3 sample_uniform_sphere(V2 u) {
    float z = 1.0f - 2.0f * u.x;
    float r = std::sqrt(rmax(1.0f - z * z, 0.0f));
    float phi = 2.0f * PI_F * u.y;
    float s, c;
    detmath::sincosf_det(phi, &s, &c);
    return {r * c, r * s, z};
}
static inline V2 uniform_sample_triangle(V2 u) {
    float su0 = std::sqrt(u.x);
    return {1.0f - su0, u.y * su0};
}

// Frame (src/math.rs:357-384)
struct Frame {
    V3 x, y, z;
    static Frame make(V3 n) {
        float sign = signum(n.z);
        float a = -1.0f / (sign + n.z);
        float b = n.x * n.y * a;
        Frame f;
        f.x = {1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x};
        f.y = {b, sign + n.y * n.y * a, -n.y};
        f.z = n;
        return f;
    }
    V3 to_world(V3 v) const { return x * v.x + y * v.y + z * v.z; }
    V3 to_local(V3 v) const { return {dot(v, x), dot(v, y), dot(v, z)}; }
};

// Distribution1D (src/math.rs:398-487)
struct Distribution1D {
    std::vector<float> cdf, func;
    float func_int = 0.0f;
    static Distribution1D normalize(const std::vector<float>& elements) {
        Distribution1D d;
        d.cdf.reserve(elements.size() + 1);
        float cur = 0.0f;
        float len = (float)elements.size();
        for (float e : elements) { d.cdf.push_back(cur); cur += e / len; }
        d.cdf.push_back(cur);
        if (cur != 0.0f) for (float& x : d.cdf) x /= cur;
        d.cdf.back() = 1.0f;
        d.func = elements;
        d.func_int = cur;
        return d;
    }
    // binary_search_by(partial_cmp): Ok(x) => x, Err(x) => x - 1.  For strictly increasing cdf
    // entries this is "last index with cdf[i] <= v" (duplicates are implementation-defined in Rust).
    size_t sample_discrete(float v) const {
        size_t lo = 0, hi = cdf.size();
        while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (cdf[mid] <= v) lo = mid + 1; else hi = mid; }
        return lo ? lo - 1 : 0;      // (a NaN table makes the reference panic; stay in bounds)
    }
    float pdf(size_t i) const { return cdf[i + 1] - cdf[i]; }
    float total() const { return func_int * (float)(cdf.size() - 1); }
    // sample_continuous (math.rs:461-478): index + position inside the bin, in [0, n]
    float sample_continuous(float v) const {
        size_t i = sample_discrete(v);
        float dv = v - cdf[i];
        float p = pdf(i);
        if (p > 0.0f) dv = dv / p;
        return (float)i + dv;
    }
};

// ------------------------------------------------------------------------------------------
// Bitmap texture lookup (src/structure.rs:434-453) and BSDFColor (src/bsdfs/mod.rs:11-116)
struct Bitmap { uint32_t w = 0, h = 0; std::vector<Color> colors; };
static inline float modulo1(float a) { return std::fmod(std::fmod(a, 1.0f) + 1.0f, 1.0f); }  // tools.rs:32-45
static inline size_t as_usize(float f) { if (!(f > 0.0f)) return 0; if (f >= 1.8446744e19f) return SIZE_MAX; return (size_t)f; }
static inline int32_t as_i32(float f) { if (f != f) return 0; if (f >= 2147483648.0f) return INT32_MAX; if (f <= -2147483648.0f) return INT32_MIN; return (int32_t)f; }

static inline float powi(float a, int b) { float r = 1.0f; for (;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; } return r; }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // lib.rs:59-67
static const float ONE_MINUS_EPSILON = 0.9999999403953552f;                                            // lib.rs:52

// Distribution2D (src/math.rs:489-532)
struct Distribution2D {
    Distribution1D marginal;
    std::vector<Distribution1D> conditionals;
    static Distribution2D from_bitmap(const Bitmap& image) {
        Distribution2D d;
        std::vector<float> marg;
        for (uint32_t y = 0; y < image.h; y++) {
            std::vector<float> row;
            for (uint32_t x = 0; x < image.w; x++) row.push_back(image.colors[(size_t)y * image.w + x].luminance());
            d.conditionals.push_back(Distribution1D::normalize(row));
            marg.push_back(d.conditionals.back().func_int);
        }
        d.marginal = Distribution1D::normalize(marg);
        return d;
    }
    V2 sample_continuous(V2 uv) const {
        float y = marginal.sample_continuous(uv.y);
        float x = conditionals[as_usize(y)].sample_continuous(uv.x);
        return {x, y};
    }
    float pdf(size_t x, size_t y) const { return conditionals[y].func[x] / marginal.func_int; }
};

// EnvironmentLightColor::Texture (src/emitter.rs:300-425): lat-long image, z up, importance sampled by luminance * sin(theta)
struct EnvTexture {
    Bitmap image;
    Distribution2D cdf;
    void build() {   // new_texture (emitter.rs:340-353)
        Bitmap image_pdf = image;
        for (uint32_t y = 0; y < image.h; y++) {
            float w = detmath::sinf_det(((float)y + 0.5f) * PI_F / (float)image.h);
            for (uint32_t x = 0; x < image.w; x++) { Color& c = image_pdf.colors[(size_t)y * image.w + x]; c.r *= w; c.g *= w; c.b *= w; }
        }
        cdf = Distribution2D::from_bitmap(image_pdf);
    }
    static V2 to_spherical_coordinates(V3 d) {   // emitter.rs:320-338
        float p = detmath::atan2f_det(d.y, d.x);
        if (p < 0.0f) p = p + 2.0f * PI_F;
        V2 uv{p * FRAC_1_PI * 0.5f, detmath::acosf_det(clampf(d.z, -1.0f, 1.0f)) * FRAC_1_PI};
        uv.x = clampf(uv.x, 0.0f, ONE_MINUS_EPSILON);
        uv.y = clampf(uv.y, 0.0f, ONE_MINUS_EPSILON);
        return uv;
    }
    Color pixel_uv(V2 uv) const {   // Bitmap::pixel_uv (structure.rs:434-453)
        float ux = modulo1(uv.x), uy = modulo1(uv.y);
        size_t x = as_usize(ux * (float)image.w), y = as_usize(uy * (float)image.h);
        size_t i = (size_t)image.w * y + x;
        return i >= image.colors.size() ? Color::zero() : image.colors[i];
    }
    void sample_direction(V2 u, V3* d, Color* value, float* pdf) const {   // emitter.rs:355-394
        V2 uv = cdf.sample_continuous(u);
        uv.x = clampf(uv.x, 0.0f, (float)image.w - 1.0f);
        uv.y = clampf(uv.y, 0.0f, (float)image.h - 1.0f);
        size_t px = as_usize(uv.x), py = as_usize(uv.y);
        Color v = image.colors[py * image.w + px];
        float p = cdf.pdf(px, py);
        float sp, cp, st, ct;
        detmath::sincosf_det((2.0f * PI_F / (float)image.w) * uv.x, &sp, &cp);
        detmath::sincosf_det((PI_F / (float)image.h) * uv.y, &st, &ct);
        *d = {st * cp, st * sp, ct};
        if (st == 0.0f) { *value = Color::zero(); *pdf = 0.0f; }
        else { *value = v; *pdf = p / (2.0f * powi(PI_F, 2) * st); }
    }
    Color eval(V3 d) const { return pixel_uv(to_spherical_coordinates(d)); }
    float pdf(V3 d) const {   // emitter.rs:407-424
        V2 uv = to_spherical_coordinates(d);
        float p = cdf.pdf(as_usize(uv.x * (float)image.w), as_usize(uv.y * (float)image.h));
        float st = detmath::sinf_det(PI_F * uv.y);
        return st == 0.0f ? 0.0f : p / (2.0f * powi(PI_F, 2) * st);
    }
};

struct Scene;
struct BSDFColor {
    int type = RL_TEX_CONSTANT;
    Color c0{1, 1, 1}, c1{0, 0, 0};
    V2 offset{0, 0}, scale{1, 1};
    float line_width = 0;
    const Bitmap* img = nullptr;
    Color color(bool has_uv, V2 uv) const {
        switch (type) {
            case RL_TEX_CONSTANT: return c0;
            case RL_TEX_BITMAP: {
                if (!has_uv || !img) return Color::zero();
                float ux = modulo1(uv.x), uy = modulo1(uv.y);
                size_t x = as_usize(ux * (float)img->w), y = as_usize(uy * (float)img->h);
                size_t i = (size_t)img->w * y + x;
                if (i >= img->colors.size()) return Color::zero();
                return img->colors[i];
            }
            case RL_TEX_CHECKERBOARD: {
                if (!has_uv) return Color::zero();
                V2 p = {uv.x * scale.x + offset.x, uv.y * scale.y + offset.y};
                int x = 2 * (as_i32(p.x * 2.0f) % 2) - 1;
                int y = 2 * (as_i32(p.y * 2.0f) % 2) - 1;
                return (x * y == 1) ? c0 : c1;
            }
            case RL_TEX_GRID: {
                if (!has_uv) return Color::zero();
                V2 p = {uv.x * scale.x + offset.x, uv.y + scale.y + offset.y};  // sic: `uv.y + scale.y` (bsdfs/mod.rs:82)
                float x = p.x - std::floor(p.x), y = p.y - std::floor(p.y);
                if (x > 0.5f) x -= 1.0f;
                if (y > 0.5f) y -= 1.0f;
                return (std::fabs(x) < line_width || std::fabs(y) < line_width) ? c0 : c1;
            }
        }
        return Color::zero();
    }
};

// ------------------------------------------------------------------------------------------
// BSDFs (src/bsdfs/*.rs)
enum BsdfTypeBits { BT_NULL = 1, BT_DIFFUSE = 2, BT_GLOSSY = 4, BT_DELTA = 8 };
enum Domain { DomSolidAngle, DomDiscrete };

struct SampledDirection { Color weight; V3 d; PDF pdf; };

static inline V3 reflect_z(V3 d) { return {-d.x, -d.y, d.z}; }  // bsdfs/mod.rs:125-127
// bsdfs/utils.rs
static inline float cos_theta(V3 w) { return w.z; }
static inline float cos_2_theta(V3 w) { return w.z * w.z; }
static inline float abs_cos_theta(V3 w) { return std::fabs(w.z); }
static inline float sin_2_theta(V3 w) { return rmax(1.0f - cos_2_theta(w), 0.0f); }
static inline float sin_theta(V3 w) { return std::sqrt(sin_2_theta(w)); }
static inline float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
static inline float hypot2(float a, float b) {
    if (std::fabs(a) > std::fabs(b)) { float r = b / a; return std::fabs(a) * std::sqrt(1.0f + r * r); }
    else if (b != 0.0f) { float r = a / b; return std::fabs(b) * std::sqrt(1.0f + r * r); }
    return 0.0f;
}
static inline V3 reflect_vector(V3 wo, V3 n) { return (-wo) + n * 2.0f * dot(wo, n); }
static inline bool check_reflection_condition(V3 wi, V3 wo) {
    return std::fabs(wi.z * wo.z - wi.x * wo.x - wi.y * wo.y - 1.0f) < 0.0001f;
}
// f32::powi = llvm.powi.f32 -> compiler-rt __powisf2 / SelectionDAG ExpandPowI: binary exponentiation
static Color fresnel_conductor(float cos_t, Color eta, Color k) {
    float c2 = cos_t * cos_t;
    float s2 = 1.0f - c2;
    float s4 = s2 * s2;
    Color temp1 = eta * eta - k * k - Color::value(s2);
    Color a2pb2 = (temp1 * temp1 + k * k * eta * eta * 4.0f).safe_sqrt();
    Color a = ((a2pb2 + temp1) * 0.5f).safe_sqrt();
    Color term1 = a2pb2 + Color::value(c2);
    Color term2 = a * (2.0f * c2);
    Color rs2 = (term1 - term2) / (term1 + term2);
    Color term3 = a2pb2 * c2 + Color::value(s4);
    Color term4 = term2 * s2;
    Color rp2 = rs2 * (term3 - term4) / (term3 + term4);
    return 0.5f * (rp2 + rs2);
}
static void fresnel_dielectric(float cos_i_, float eta, float* fres, float* cos_t_out) {
    if (eta == 1.0f) { *fres = 0.0f; *cos_t_out = -cos_i_; return; }
    float scale = cos_i_ > 0.0f ? 1.0f / eta : eta;
    float cos_t_sqr = 1.0f - (1.0f - cos_i_ * cos_i_) * (scale * scale);
    if (cos_t_sqr <= 0.0f) { *fres = 1.0f; *cos_t_out = 0.0f; return; }
    float cos_i = std::fabs(cos_i_);
    float cos_t = std::sqrt(cos_t_sqr);
    float rs = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
    float rp = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
    *cos_t_out = cos_i_ > 0.0f ? -cos_t : cos_t;
    *fres = 0.5f * (rs * rs + rp * rp);
}

// MicrofacetDistribution (bsdfs/distribution.rs:19-145)
struct Microfacet {
    int type;  // RL_MICROFACET_*
    float alpha_u, alpha_v;
    float eval(V3 m) const {
        if (cos_theta(m) <= 0.0f) return 0.0f;
        float c2 = cos_2_theta(m);
        float bexp = ((m.x * m.x) / (alpha_u * alpha_u) + (m.y * m.y) / (alpha_v * alpha_v)) / c2;
        float res;
        if (type == RL_MICROFACET_BECKMANN) res = detmath::expf_det(-bexp) / (PI_F * alpha_u * alpha_v * c2 * c2);
        else { float root = (1.0f + bexp) * c2; res = 1.0f / (PI_F * alpha_u * alpha_v * root * root); }
        if (res * cos_theta(m) < 1e-20f) return 0.0f;
        return res;
    }
    float pdf(V3 m) const { return eval(m) * cos_theta(m); }
    void sample(V2 s, V3* m, float* pdf_out) const {
        float sin_phi, cos_phi;
        detmath::sincosf_det(2.0f * PI_F * s.y, &sin_phi, &cos_phi);
        float alpha_sqr = alpha_u * alpha_v;
        float cos_m, pdf;
        if (type == RL_MICROFACET_BECKMANN) {
            float tan2 = alpha_sqr * -detmath::logf_det(1.0f - s.x);
            cos_m = 1.0f / std::sqrt(1.0f + tan2);
            pdf = (1.0f - s.x) / (PI_F * alpha_u * alpha_v * powi(cos_m, 3));
        } else {
            float tan2 = alpha_sqr * s.x / (1.0f - s.x);
            cos_m = 1.0f / std::sqrt(1.0f + tan2);
            float tmp = 1.0f + tan2 / alpha_sqr;
            pdf = FRAC_1_PI / (alpha_u * alpha_v * powi(cos_m, 3) * powi(tmp, 2));
        }
        if (pdf < 1e-20f) pdf = 0.0f;
        float sin_m = std::sqrt(rmax(1.0f - powi(cos_m, 2), 0.0f));
        *m = {sin_m * cos_phi, sin_m * sin_phi, cos_m};
        *pdf_out = pdf;
    }
    float smith_g1(V3 v, V3 m) const {
        if (dot(v, m) * cos_theta(v) <= 0.0f) return 0.0f;
        float tt = std::fabs(tan_theta(v));
        if (tt == 0.0f) return 1.0f;
        float alpha = alpha_u;
        if (type == RL_MICROFACET_BECKMANN) {
            float a = 1.0f / (alpha * tt);
            if (a >= 1.6f) return 1.0f;
            float a2 = powi(a, 2);
            return (3.535f * a + 2.181f * a2) / (1.0f + 2.276f * a + 2.577f * a2);
        }
        float root = alpha * tt;
        return 2.0f / (1.0f + hypot2(1.0f, root));
    }
    float g(V3 wi, V3 wo, V3 m) const { return smith_g1(wi, m) * smith_g1(wo, m); }
};

struct BSDF {
    int type = RL_BSDF_DIFFUSE;
    BSDFColor diffuse, specular, transmittance, eta, k;
    float exponent = 0, weight_specular = 0;
    int distribution = RL_MICROFACET_NONE;
    float alpha_u = 0, alpha_v = 0;
    float g_eta = 1, g_inv_eta = 1;  // glass

    bool is_twosided() const { return type != RL_BSDF_GLASS; }
    int bsdf_type() const {
        switch (type) {
            case RL_BSDF_DIFFUSE: return BT_DIFFUSE;
            case RL_BSDF_PHONG: return BT_GLOSSY;
            case RL_BSDF_METAL: return distribution ? BT_GLOSSY : BT_DELTA;
            case RL_BSDF_GLASS: return BT_DELTA;
            case RL_BSDF_SUBSTRATE: return distribution ? (BT_GLOSSY | BT_DIFFUSE) : (BT_DELTA | BT_DIFFUSE);
        }
        return BT_NULL;
    }
    bool is_smooth() const { return (bsdf_type() & (BT_DELTA | BT_NULL)) != 0; }  // bsdfs/mod.rs:157-161
    Microfacet distr() const { return {distribution, alpha_u, alpha_v}; }

    Color schlick_fresnel(bool huv, V2 uv, float cos_t) const {  // substrate.rs:14-19
        Color rs = specular.color(huv, uv);
        return rs + (Color::one() - rs) * powi(1.0f - cos_t, 5);
    }

    // fn pdf (domain explicit as in the trait)
    PDF pdf(bool huv, V2 uv, V3 wi, V3 wo, Domain domain) const {
        switch (type) {
            case RL_BSDF_DIFFUSE:  // diffuse.rs:35-53
                if (wi.z <= 0.0f) return PDF::solid_angle(0.0f);
                if (wo.z <= 0.0f) return PDF::solid_angle(0.0f);
                return PDF::solid_angle(wo.z * FRAC_1_PI);
            case RL_BSDF_PHONG: {  // phong.rs:66-91
                if (wi.z <= 0.0f || wo.z <= 0.0f) return PDF::solid_angle(0.0f);
                float alpha = dot(reflect_z(wi), wo);
                float pdf_spec = 0.0f;
                if (alpha > 0.0f)
                    pdf_spec = weight_specular * detmath::powf_det(alpha, exponent) * (exponent + 1.0f) / (2.0f * PI_F);
                float pdf_diff = (1.0f - weight_specular) * wo.z * FRAC_1_PI;
                return PDF::solid_angle(pdf_spec + pdf_diff);
            }
            case RL_BSDF_METAL: {  // metal.rs:76-107
                if (!distribution) return PDF::discrete(1.0f);
                V3 h = normalize(wi + wo);
                return PDF::solid_angle(distr().pdf(h) / (4.0f * std::fabs(dot(wo, h))));
            }
            case RL_BSDF_GLASS: return PDF::discrete(0.0f);  // todo!() in the reference, never called
            case RL_BSDF_SUBSTRATE: {  // substrate.rs:95-149
                auto zero = [&]() { return domain == DomSolidAngle ? PDF::solid_angle(0.0f) : PDF::discrete(0.0f); };
                if (wi.z <= 0.0f || wo.z <= 0.0f) return zero();
                V3 m = wi + wo;
                if (m.x == 0.0f && m.y == 0.0f && m.z == 0.0f) return zero();
                m = normalize(m);
                if (domain == DomDiscrete) return PDF::discrete(0.5f);
                float pdf_diff = wo.z * FRAC_1_PI;
                float pdf_spec = 0.0f;
                if (distribution) pdf_spec = distr().pdf(m) / (4.0f * std::fabs(dot(wo, m)));
                return PDF::solid_angle(0.5f * (pdf_diff + pdf_spec));
            }
        }
        return PDF::solid_angle(0.0f);
    }

    Color eval(bool huv, V2 uv, V3 wi, V3 wo, Domain domain) const {
        switch (type) {
            case RL_BSDF_DIFFUSE:  // diffuse.rs:55-73
                if (wi.z <= 0.0f) return Color::zero();
                if (wo.z > 0.0f) return diffuse.color(huv, uv) * wo.z * FRAC_1_PI;
                return Color::zero();
            case RL_BSDF_PHONG: {  // phong.rs:93-121
                if (wi.z <= 0.0f || wo.z <= 0.0f) return Color::zero();
                float alpha = dot(reflect_z(wi), wo);
                Color spec = Color::zero();
                if (alpha > 0.0f)
                    spec = specular.color(huv, uv) * (detmath::powf_det(alpha, exponent) * (exponent + 2.0f) / (2.0f * PI_F));
                Color diff = diffuse.color(huv, uv) * wo.z * FRAC_1_PI;
                return spec + diff;
            }
            case RL_BSDF_METAL: {  // metal.rs:109-155
                if (!distribution)
                    return specular.color(huv, uv) * fresnel_conductor(std::fabs(wi.z), eta.color(huv, uv), k.color(huv, uv));
                V3 h = normalize(wi + wo);
                Microfacet di = distr();
                float d = di.eval(h);
                if (d == 0.0f) return Color::zero();
                Color f = specular.color(huv, uv) * fresnel_conductor(dot(wi, h), eta.color(huv, uv), k.color(huv, uv));
                float g = di.g(wi, wo, h);
                float model = d * g / (4.0f * cos_theta(wi));
                return f * model;
            }
            case RL_BSDF_GLASS: return Color::zero();  // eval on a delta BSDF is never reached on this path
            case RL_BSDF_SUBSTRATE: {  // substrate.rs:151-205
                if (wi.z <= 0.0f || wo.z <= 0.0f) return Color::zero();
                V3 m = wi + wo;
                if (m.x == 0.0f && m.y == 0.0f && m.z == 0.0f) return Color::zero();
                m = normalize(m);
                if (domain == DomDiscrete) return schlick_fresnel(huv, uv, dot(wi, m));
                Color diff = diffuse.color(huv, uv) * (Color::one() - specular.color(huv, uv)) *
                             (28.0f / (23.0f * PI_F)) *
                             (1.0f - powi(1.0f - 0.5f * abs_cos_theta(wi), 5)) *
                             (1.0f - powi(1.0f - 0.5f * abs_cos_theta(wo), 5));
                Color spec = Color::zero();
                if (distribution) {
                    float model = distr().eval(m) /
                                  (4.0f * std::fabs(dot(wi, m)) * rmax(std::fabs(cos_theta(wi)), std::fabs(cos_theta(wo))));
                    spec = model * schlick_fresnel(huv, uv, dot(wi, m));
                }
                return (diff + spec) * wo.z;
            }
        }
        return Color::zero();
    }

    // fn sample -> Option<SampledDirection>
    bool sample(bool huv, V2 uv, V3 wi, V2 s, SampledDirection* out) const {
        switch (type) {
            case RL_BSDF_DIFFUSE: {  // diffuse.rs:11-33
                if (wi.z <= 0.0f) return false;
                V3 d = cosine_sample_hemisphere(s);
                *out = {diffuse.color(huv, uv), d, PDF::solid_angle(d.z * FRAC_1_PI)};
                return true;
            }
            case RL_BSDF_PHONG: {  // phong.rs:14-64
                if (wi.z <= 0.0f) return false;
                V3 d;
                if (s.x < weight_specular) {
                    s.x /= weight_specular;
                    float sin_a = std::sqrt(1.0f - detmath::powf_det(s.y, 2.0f / (exponent + 1.0f)));
                    float cos_a = detmath::powf_det(s.y, 1.0f / (exponent + 1.0f));
                    float phi = 2.0f * PI_F * s.x;
                    V3 local = {sin_a * detmath::cosf_det(phi), sin_a * detmath::sinf_det(phi), cos_a};
                    Frame fr = Frame::make(reflect_z(wi));
                    d = fr.to_world(local);
                    if (d.z <= 0.0f) return false;
                } else {
                    s.x = (s.x - weight_specular) / (1.0f - weight_specular);
                    d = cosine_sample_hemisphere(s);
                }
                PDF p = pdf(huv, uv, wi, d, DomSolidAngle);
                if (p.value() == 0.0f) return false;
                *out = {eval(huv, uv, wi, d, DomSolidAngle) / p.value(), d, p};
                return true;
            }
            case RL_BSDF_METAL: {  // metal.rs:15-74
                if (wi.z <= 0.0f) return false;
                if (!distribution) {
                    *out = {specular.color(huv, uv) * fresnel_conductor(wi.z, eta.color(huv, uv), k.color(huv, uv)),
                            reflect_z(wi), PDF::discrete(1.0f)};
                    return true;
                }
                Microfacet di = distr();
                V3 m; float p;
                di.sample(s, &m, &p);
                if (p == 0.0f) return false;
                V3 wo = reflect_vector(wi, m);
                if (cos_theta(wo) <= 0.0f) return false;
                Color f = fresnel_conductor(dot(wi, m), eta.color(huv, uv), k.color(huv, uv)) * specular.color(huv, uv);
                float w = di.eval(m) * di.g(wi, wo, m) * dot(wi, m) / (p * cos_theta(wi));
                *out = {w * f, wo, PDF::solid_angle(p)};
                return true;
            }
            case RL_BSDF_GLASS: {  // glass.rs:76-121 (Transport::Importance => factor 1)
                float fres, cos_t;
                fresnel_dielectric(wi.z, g_eta, &fres, &cos_t);
                if (s.x <= fres) {
                    *out = {specular.color(huv, uv), reflect_z(wi), PDF::discrete(fres)};
                } else {
                    float factor = 1.0f;
                    float scale = cos_t < 0.0f ? -g_inv_eta : -g_eta;
                    V3 d = {scale * wi.x, scale * wi.y, cos_t};
                    *out = {transmittance.color(huv, uv) * factor * factor, d, PDF::discrete(fres)};
                }
                return true;
            }
            case RL_BSDF_SUBSTRATE: {  // substrate.rs:22-93
                if (wi.z <= 0.0f) return false;
                V3 d; Domain dom;
                if (s.x < 0.5f) {
                    s.x *= 2.0f;
                    d = cosine_sample_hemisphere(s);
                    dom = DomSolidAngle;
                } else {
                    s.x = (s.x - 0.5f) * 2.0f;
                    V3 m;
                    if (!distribution) { m = {0.0f, 0.0f, 1.0f}; dom = DomDiscrete; }
                    else { float p; distr().sample(s, &m, &p); if (p == 0.0f) return false; dom = DomSolidAngle; }
                    d = reflect_vector(wi, m);
                    if (cos_theta(d) <= 0.0f) return false;
                }
                PDF p = pdf(huv, uv, wi, d, dom);
                if (p.value() == 0.0f) return false;
                Color f = eval(huv, uv, wi, d, dom);
                *out = {f / p.value(), d, p};
                return true;
            }
        }
        return false;
    }
};

// ------------------------------------------------------------------------------------------
// Ray / AABB (src/structure.rs:696-878)
struct Ray {
    V3 o, d; float tnear, tfar;
    static Ray make(V3 o, V3 d) { return {o, d, EPSILON, F32_MAX}; }
};
struct AABB {
    V3 p_min{F32_MAX, F32_MAX, F32_MAX}, p_max{-F32_MAX, -F32_MAX, -F32_MAX};
    AABB union_aabb(const AABB& b) const {
        AABB r;
        r.p_min = {rmin(p_min.x, b.p_min.x), rmin(p_min.y, b.p_min.y), rmin(p_min.z, b.p_min.z)};
        r.p_max = {rmax(p_max.x, b.p_max.x), rmax(p_max.y, b.p_max.y), rmax(p_max.z, b.p_max.z)};
        return r;
    }
    AABB union_vec(V3 v) const {
        AABB r;
        r.p_min = {rmin(p_min.x, v.x), rmin(p_min.y, v.y), rmin(p_min.z, v.z)};
        r.p_max = {rmax(p_max.x, v.x), rmax(p_max.y, v.y), rmax(p_max.z, v.z)};
        return r;
    }
    V3 size() const { return p_max - p_min; }
    V3 center() const { return size() * 0.5f + p_min; }
    float surface_area() const { V3 d = size(); return (0.0f + (1.0f * d.y) * d.z + (1.0f * d.x) * d.z) + (1.0f * d.x) * d.y; }
    // structure.rs:849-869: returns entry distance, or negative-one flag via `ok`
    bool intersect(const Ray& r, float* t_out) const {
        float t_max = r.tfar, t_min = r.tnear;
        for (int d = 0; d < 3; d++) {
            float inv_d = 1.0f / r.d[d];
            float t0 = (p_min[d] - r.o[d]) * inv_d;
            float t1 = (p_max[d] - r.o[d]) * inv_d;
            if (inv_d < 0.0f) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        *t_out = t_min;
        return true;
    }
};
struct BoundingSphere { V3 center{0, 0, 0}; float radius = 0; };

// ------------------------------------------------------------------------------------------
// Mesh (src/geometry.rs:107-457)
struct IntersectionUV { float t; V3 p, n; float u, v; };

struct Mesh {
    std::vector<V3> vertices;
    std::vector<uint32_t> indices;  // 3 per triangle
    bool has_normals = false, has_uv = false;
    std::vector<V3> normals;
    std::vector<V2> uv;
    BSDF bsdf;
    bool is_light = false;   // EmissionType::{Color, HSV, Texture}: anything but Zero
    Color emission = Color::zero();
    int emission_type = 0;           // 0: EmissionType::Color { v = emission }; 1: HSV { scale }; 2: Texture { scale, img } (geometry.rs:99-104)
    float emission_scale = 1.0f;
    const Bitmap* emission_img = nullptr;
    Distribution1D cdf;
    size_t n_tris() const { return indices.size() / 3; }

    // Mesh::new (geometry.rs:122-182); returns false for an empty mesh
    bool finish() {
        std::vector<float> areas;
        areas.reserve(n_tris());
        for (size_t i = 0; i < n_tris(); i++) {
            V3 v0 = vertices[indices[3 * i]], v1 = vertices[indices[3 * i + 1]], v2 = vertices[indices[3 * i + 2]];
            areas.push_back(magnitude(cross(v1 - v0, v2 - v0)) * 0.5f);
        }
        if (has_normals) {
            size_t wrong = 0;
            for (V3& n : normals) {
                float l = dot(n, n);
                if (l == 0.0f) wrong++;
                else if (l != 1.0f) n = n / std::sqrt(l);
            }
            if (wrong > 0 && wrong == normals.size()) { has_normals = false; normals.clear(); }
        }
        if (areas.empty()) return false;
        cdf = Distribution1D::normalize(areas);
        return true;
    }
    // Mesh::emit (geometry.rs:184-206).  HSV / Texture `uv.unwrap()`: a light mesh without uv panics in the reference — the API refuses to put such an
    // emission on a mesh without uv, so has_uv is true whenever it matters.
    Color emit(bool huv, V2 uv) const {
        if (!is_light) return Color::zero();
        if (emission_type == 1) {
            const Color c1{1.0f, 0.0f, 0.0f}, c2{0.0f, 1.0f, 0.0f};
            float x = std::fmod(std::fabs(uv.x), 1.0f);                 // uv.x.abs() % 1.0
            Color c = x * c1 + (1.0f - x) * c2;
            return c * emission_scale;                                  // Color * f32: guarded (structure.rs:278-292)
        }
        if (emission_type == 2) {
            if (!huv || !emission_img) return Color::zero();
            float ux = modulo1(uv.x), uy = modulo1(uv.y);               // Bitmap::pixel_uv (structure.rs:434-453)
            size_t x = as_usize(ux * (float)emission_img->w), y = as_usize(uy * (float)emission_img->h);
            size_t i = (size_t)emission_img->w * y + x;
            Color c = i >= emission_img->colors.size() ? Color::zero() : emission_img->colors[i];
            return c * emission_scale;
        }
        return emission;
    }
    // the colour Emitter::flux uses (emitter.rs:591-599): the constant emission, or Color::value(scale) for the uv-dependent kinds ("TODO" there)
    Color flux_emission() const { return !is_light ? Color::zero() : (emission_type == 0 ? emission : Color{emission_scale, emission_scale, emission_scale}); }
    float pdf() const { return 1.0f / cdf.total(); }

    // geometry.rs:358-410
    bool intersection_tri(size_t i, V3 p_c, V3 d_c, IntersectionUV* its) const {
        V3 v0 = vertices[indices[3 * i]], v1 = vertices[indices[3 * i + 1]], v2 = vertices[indices[3 * i + 2]];
        V3 e1 = v1 - v0, e2 = v2 - v0;
        V3 n_geo = normalize(cross(e1, e2));
        float denom = dot(d_c, n_geo);
        if (denom == 0.0f) return false;
        float t = -dot(p_c - v0, n_geo) / denom;
        if (t < 0.0f) return false;
        V3 p = p_c + t * d_c;
        float det = magnitude(cross(e1, e2));
        V3 u0 = cross(e1, p - v0);
        V3 w0 = cross(p - v0, e2);
        if (dot(u0, n_geo) < 0.0f || dot(w0, n_geo) < 0.0f) return false;
        float v = magnitude(u0) / det;
        float u = magnitude(w0) / det;
        if (u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) return false;
        if (u + v <= 1.0f) {
            if (t < its->t && t > 0.00001f) {
                its->t = t; its->u = u; its->v = v; its->p = p; its->n = n_geo;
                return true;
            }
        }
        return false;
    }
    // geometry.rs:423-439
    AABB compute_aabb_tri(size_t i) const {
        AABB aabb;
        for (int s = 0; s < 3; s++) aabb = aabb.union_vec(vertices[indices[3 * i + s]]);
        V3 s = aabb.size();
        for (int k = 0; k < 3; k++) if (s[k] < EPSILON) { aabb.p_max.at(k) += EPSILON; aabb.p_min.at(k) -= EPSILON; }
        return aabb;
    }
    AABB compute_aabb() const {
        AABB aabb;
        for (const V3& v : vertices) aabb = aabb.union_vec(v);
        V3 s = aabb.size();
        for (int k = 0; k < 3; k++) if (s[k] < EPSILON) { aabb.p_max.at(k) += EPSILON; aabb.p_min.at(k) -= EPSILON; }
        return aabb;
    }

    struct SampledPosition { V3 p, n; bool has_uv; V2 uv; PDF pdf; };
    // geometry.rs:261-337
    SampledPosition sample_tri(size_t prim, V2 v) const {
        V3 v0 = vertices[indices[3 * prim]], v1 = vertices[indices[3 * prim + 1]], v2 = vertices[indices[3 * prim + 2]];
        V2 b = uniform_sample_triangle(v);
        V3 pos = v0 * b.x + v1 * b.y + v2 * (1.0f - b.x - b.y);
        V3 n_g = normalize(cross(v2 - v0, v1 - v0));
        if (has_normals) {
            V3 n0 = normals[indices[3 * prim]], n1 = normals[indices[3 * prim + 1]], n2 = normals[indices[3 * prim + 2]];
            V3 n = n0 * b.x + n1 * b.y + n2 * (1.0f - b.x - b.y);
            float n_l = magnitude2(n);
            if (n_l == 0.0f) n = n_g;
            else if (n_l != 1.0f) n = n / std::sqrt(n_l);
            if (dot(n_g, n) < 0.0f) n_g = -n_g;
        }
        SampledPosition sp;
        sp.has_uv = has_uv;
        sp.uv = {0, 0};
        if (has_uv) {
            V2 a0 = uv[indices[3 * prim]], a1 = uv[indices[3 * prim + 1]], a2 = uv[indices[3 * prim + 2]];
            float w2 = 1.0f - b.x - b.y;
            V2 t = {a0.x * b.x + a1.x * b.y + a2.x * w2, a0.y * b.x + a1.y * b.y + a2.y * w2};
            float inv = 1.0f / std::sqrt(t.x * t.x + t.y * t.y);  // `.normalize()` (sic, geometry.rs:322)
            sp.uv = {t.x * inv, t.y * inv};
        }
        float area_tri = magnitude(cross(v1 - v0, v2 - v0)) * 0.5f;
        sp.p = pos; sp.n = n_g; sp.pdf = PDF::area(1.0f / area_tri);
        return sp;
    }
    // geometry.rs:340-348
    SampledPosition sample(float s, V2 v) const {
        size_t prim = cdf.sample_discrete(s);
        SampledPosition r = sample_tri(prim, v);
        r.pdf = PDF::area(1.0f / cdf.total());
        return r;
    }
};

// ------------------------------------------------------------------------------------------
// LightSampling & the mesh area emitter (src/emitter.rs:10-44, 570-688)
struct LightSampling { int emitter; PDF pdf; V3 p, n, d; Color weight; bool has_uv = false; V2 uv{0, 0}; bool is_valid() const { return !pdf.is_zero(); } };
struct LightSamplingPDF { V3 o, p, n, dir; };

static PDF mesh_direct_pdf(const Mesh& m, const LightSamplingPDF& ls) {  // emitter.rs:571-579
    float cos_light = rmax(dot(ls.n, -ls.dir), 0.0f);
    if (cos_light == 0.0f) return PDF::solid_angle(0.0f);
    float geom = cos_light / magnitude2(ls.p - ls.o);
    return PDF::solid_angle(m.pdf() / geom);
}
static LightSampling mesh_direct_sample(const Mesh& m, V3 p, float r, V2 uv) {  // emitter.rs:652-688
    Mesh::SampledPosition sp = m.sample(r, uv);
    V3 d = sp.p - p;
    float dist = magnitude(d);
    if (dist != 0.0f) d = d / dist;
    float geom = dist != 0.0f ? rmax(dot(sp.n, -d), 0.0f) / (dist * dist) : 0.0f;
    float pdf_area = sp.pdf.value();
    PDF pdf = sp.pdf.as_solid_angle_geom(geom);
    Color weight = pdf.is_zero() ? Color::zero() : m.emit(sp.has_uv, sp.uv) * geom / pdf_area;
    return {-1, pdf, sp.p, sp.n, d, weight, sp.has_uv, sp.uv};
}
static Color mesh_flux(const Mesh& m) { return m.cdf.total() * m.flux_emission() * PI_F; }  // emitter.rs:591-599

// solve_quadratic (src/math.rs:324-352) and BoundingSphere::intersect (src/structure.rs:894-917)
static bool solve_quadratic(float a, float b, float c, float* x0, float* x1) {
    if (a == 0.0f) { if (b != 0.0f) { float v = -c / b; *x0 = v; *x1 = v; return true; } return false; }
    float d = b * b - 4.0f * a * c;
    if (d < 0.0f) return false;
    float d_sqrt = std::sqrt(d);
    float tmp = b < 0.0f ? -0.5f * (b - d_sqrt) : -0.5f * (b + d_sqrt);
    float r0 = tmp / a, r1 = c / tmp;
    if (r0 > r1) { *x0 = r1; *x1 = r0; } else { *x0 = r0; *x1 = r1; }
    return true;
}
static bool bsphere_intersect(const BoundingSphere& s, const Ray& r, float* t) {
    V3 d_p = s.center - r.o;
    float a = magnitude2(r.d);
    float b = 2.0f * dot(d_p, r.d);
    float c = magnitude2(d_p) - s.radius * s.radius;
    float t0, t1;
    if (!solve_quadratic(a, b, c, &t0, &t1)) return false;
    if (t0 < r.tnear) { if (t1 < r.tfar) { *t = t1; return true; } return false; }
    if (t0 < r.tfar) { *t = t0; return true; }
    return false;
}

// the non-mesh emitters (src/emitter.rs:96-250, 300-568); EnvironmentLightColor::Constant only
enum EmitterKind { EM_MESH = 0, EM_ENV = 1, EM_POINT = 2, EM_DIRECTIONAL = 3 };
struct EmitterRec {
    int kind = EM_MESH;
    int mesh = -1;
    V3 v{0, 0, 0};            // point position / light direction
    Color c = Color::zero();  // intensity / environment luminance
    BoundingSphere bsphere;   // preprocess(): scene.bsphere with radius * 1.1
};

// ------------------------------------------------------------------------------------------
// cgmath Matrix4 (column-major) — restated from memory of cgmath 0.18 (third party, unpinned)
struct V4 { float x, y, z, w; };
static inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
struct M4 {
    V4 c[4];
    static M4 identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
    static M4 from_nonuniform_scale(float x, float y, float z) { return {{{x, 0, 0, 0}, {0, y, 0, 0}, {0, 0, z, 0}, {0, 0, 0, 1}}}; }
    static M4 from_translation(V3 v) { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {v.x, v.y, v.z, 1}}}; }
    float at(int col, int row) const { const V4& v = c[col]; return row == 0 ? v.x : row == 1 ? v.y : row == 2 ? v.z : v.w; }
    V4 mulv(V4 v) const { return c[0] * v.x + c[1] * v.y + c[2] * v.z + c[3] * v.w; }
    M4 mul(const M4& r) const { M4 o; for (int j = 0; j < 4; j++) o.c[j] = mulv(r.c[j]); return o; }
    V3 transform_vector(V3 v) const { V4 r = mulv({v.x, v.y, v.z, 0.0f}); return {r.x, r.y, r.z}; }
    V3 transform_point(V3 p) const {
        V4 r = mulv({p.x, p.y, p.z, 1.0f});
        float inv = 1.0f / r.w;
        return {r.x * inv, r.y * inv, r.z * inv};
    }
    static float det3(float a00, float a01, float a02, float a10, float a11, float a12, float a20, float a21, float a22) {
        // Matrix3::determinant with a[col][row]
        return a00 * (a11 * a22 - a21 * a12) - a10 * (a01 * a22 - a21 * a02) + a20 * (a01 * a12 - a11 * a02);
    }
    // cofactor of (col i, row j) of the transpose, as cgmath's Matrix4::invert builds it
    float cofactor(const M4& t, int i, int j) const {
        float m[3][3];
        int ci = 0;
        for (int col = 0; col < 4; col++) {
            if (col == i) continue;
            int ri = 0;
            for (int row = 0; row < 4; row++) { if (row == j) continue; m[ci][ri++] = t.at(col, row); }
            ci++;
        }
        float d = det3(m[0][0], m[0][1], m[0][2], m[1][0], m[1][1], m[1][2], m[2][0], m[2][1], m[2][2]);
        return ((i + j) & 1) ? -d : d;
    }
    M4 transpose() const { M4 t; for (int a = 0; a < 4; a++) t.c[a] = {at(0, a), at(1, a), at(2, a), at(3, a)}; return t; }
    float determinant() const {
        M4 t = transpose();
        float d = 0.0f;
        for (int j = 0; j < 4; j++) d += at(j, 0) * cofactor(t, 0, j) * 1.0f;  // expansion along row 0
        return d;
    }
    bool invert(M4* out) const {
        float det = determinant();
        if (det == 0.0f) return false;
        float inv_det = 1.0f / det;
        M4 t = transpose();
        float e[4][4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) e[i][j] = cofactor(t, i, j) * inv_det;
        for (int i = 0; i < 4; i++) out->c[i] = {e[i][0], e[i][1], e[i][2], e[i][3]};
        return true;
    }
};
static M4 perspective(float fovy_rad, float aspect, float near, float far) {
    float f = 1.0f / std::tan(fovy_rad / 2.0f);
    M4 m;
    m.c[0] = {f / aspect, 0, 0, 0};
    m.c[1] = {0, f, 0, 0};
    m.c[2] = {0, 0, (far + near) / (near - far), -1.0f};
    m.c[3] = {0, 0, (2.0f * far * near) / (near - far), 0};
    return m;
}

// Camera (src/camera.rs:5-91, 140-142)
struct Camera {
    uint32_t w = 0, h = 0;
    M4 camera_to_sample, sample_to_camera, to_world, to_local;
    bool init(uint32_t w_, uint32_t h_, float fov, int fov_axis, const M4& mat, bool flip) {
        w = w_; h = h_;
        to_world = mat;
        if (!to_world.invert(&to_local)) return false;
        float x_v = flip ? 1.0f : -1.0f;
        float aspect = (float)w / (float)h;
        float fov_rad = fov_axis == 0 ? fov * PI_F / 180.0f : fov * aspect * PI_F / 180.0f;
        camera_to_sample = M4::from_nonuniform_scale(-0.5f, -0.5f * aspect, 1.0f)
                               .mul(M4::from_translation({-1.0f, -1.0f / aspect, 0.0f}))
                               .mul(perspective(fov_rad, 1.0f, 1e-2f, 1000.0f))
                               .mul(M4::from_nonuniform_scale(x_v, 1.0f, -1.0f));
        return camera_to_sample.invert(&sample_to_camera);
    }
    V3 position() const { return to_world.transform_point({0, 0, 0}); }
    Ray generate(V2 px) const {
        V3 near_p = sample_to_camera.transform_point({px.x / (float)w, px.y / (float)h, 0.0f});
        V3 d = normalize(near_p);
        return Ray::make(position(), to_world.transform_vector(d));
    }
};

// ------------------------------------------------------------------------------------------
// ATS — adaptive tree splitting light sampler, the `-x ats` option (src/emitter.rs:783-1488, 1540-1639): a BVH over the
// emissive triangles whose nodes carry an orientation cone and a flux; a light is drawn by descending with probabilities
// proportional to LightBounds::importance_point.  Only `sample` / `pdf` are on the path / direct integrators.
struct DirectionCone { V3 w{0, 0, 1}; float cos_theta = -1.0f; bool empty = false; };
static inline float safe_acos(float v) { return detmath::acosf_det(rmin(rmax(v, -1.0f), 1.0f)); }
static inline float safe_asin(float v) { return detmath::asinf_det(rmin(rmax(v, -1.0f), 1.0f)); }
static inline float angle_between(V3 v1, V3 v2) {   // emitter.rs:792-798
    if (dot(v1, v2) < 0.0f) return PI_F - 2.0f * safe_asin(magnitude(v2 + v1) / 2.0f);
    return 2.0f * safe_asin(magnitude(v2 - v1) / 2.0f);
}
static inline float to_degrees(float r) { return r * 57.2957795130823208767981548141051703f; }        // f32::to_degrees
static inline float to_radians(float d) { const float k = PI_F / 180.0f; return d * k; }               // f32::to_radians
static M4 rotate_sc(float sin_theta, float cos_theta, V3 axis) {   // emitter.rs:800-819 (built column-wise, then transposed)
    V3 a = normalize(axis);
    M4 m;
    m.c[0] = {a.x * a.x + (1.0f - a.x * a.x) * cos_theta, a.x * a.y * (1.0f - cos_theta) - a.z * sin_theta, a.x * a.z * (1.0f - cos_theta) + a.y * sin_theta, 0.0f};
    m.c[1] = {a.x * a.y * (1.0f - cos_theta) + a.z * sin_theta, a.y * a.y + (1.0f - a.y * a.y) * cos_theta, a.y * a.z * (1.0f - cos_theta) - a.x * sin_theta, 0.0f};
    m.c[2] = {a.x * a.z * (1.0f - cos_theta) - a.y * sin_theta, a.y * a.z * (1.0f - cos_theta) + a.x * sin_theta, a.z * a.z + (1.0f - a.z * a.z) * cos_theta, 0.0f};
    m.c[3] = {0.0f, 0.0f, 0.0f, 1.0f};
    return m.transpose();
}
static M4 rotate_angle_axis(float theta_deg, V3 axis) {
    return rotate_sc(detmath::sinf_det(to_radians(theta_deg)), detmath::cosf_det(to_radians(theta_deg)), axis);
}
static BoundingSphere aabb_to_sphere(const AABB& b) { V3 c = b.center(); return {c, magnitude(c - b.p_max)}; }   // structure.rs:871-877
static DirectionCone cone_subtended(const AABB& b, V3 p) {   // emitter.rs:831-846
    BoundingSphere s = aabb_to_sphere(b);
    if (magnitude2(p - s.center) < s.radius * s.radius) return DirectionCone();
    DirectionCone c;
    c.w = normalize(s.center - p);
    float sin2 = s.radius * s.radius / magnitude2(s.center - p);
    c.cos_theta = std::sqrt(rmax(1.0f - sin2, 0.0f));
    return c;
}
static DirectionCone cone_union(const DirectionCone& a, const DirectionCone& b) {   // emitter.rs:848-888
    if (a.empty) return b;
    if (b.empty) return a;
    float theta_a = safe_acos(a.cos_theta), theta_b = safe_acos(b.cos_theta), theta_d = angle_between(a.w, b.w);
    if (rmin(theta_d + theta_b, PI_F) <= theta_a) return a;
    if (rmin(theta_d + theta_a, PI_F) <= theta_b) return b;
    float theta_o = (theta_a + theta_d + theta_b) / 2.0f;
    if (theta_o >= PI_F) return DirectionCone();
    float theta_r = theta_o - theta_a;
    V3 wr = cross(a.w, b.w);
    if (magnitude2(wr) == 0.0f) return DirectionCone();
    DirectionCone c;
    c.w = rotate_angle_axis(to_degrees(theta_r), wr).transform_vector(a.w);
    c.cos_theta = detmath::cosf_det(theta_o);
    return c;
}
struct LightBounds {   // emitter.rs:901-935
    AABB aabb;
    V3 w{0, 0, 1};
    float phi = 0, theta_o = 0, theta_e = 0, cos_theta_o = 1, cos_theta_e = 1;
    bool two_sided = false;
    size_t number_lights = 0;
    float phi_sqr = 0;
    static LightBounds merge(const LightBounds& a, const LightBounds& b) {   // LightBounds::union (emitter.rs:947-973)
        if (a.phi == 0.0f) return b;
        if (b.phi == 0.0f) return a;
        DirectionCone ca; ca.w = a.w; ca.cos_theta = a.cos_theta_o;
        DirectionCone cb; cb.w = b.w; cb.cos_theta = b.cos_theta_o;
        DirectionCone c = cone_union(ca, cb);
        LightBounds r;
        r.theta_o = safe_acos(c.cos_theta);
        r.theta_e = rmax(a.theta_e, b.theta_e);
        r.aabb = a.aabb.union_aabb(b.aabb);
        r.w = c.w;
        r.phi = a.phi + b.phi;
        r.cos_theta_o = detmath::cosf_det(r.theta_o);
        r.cos_theta_e = detmath::cosf_det(r.theta_e);
        r.two_sided = a.two_sided | b.two_sided;
        r.number_lights = a.number_lights + b.number_lights;
        r.phi_sqr = a.phi_sqr + b.phi_sqr;
        return r;
    }
    // importance_point (emitter.rs:1024-1086)
    float importance_point(V3 p, const V3* n) const {
        V3 pc = aabb.center();
        float d2 = rmax(magnitude2(p - pc), 0.0001f);
        V3 wi = normalize(p - pc);
        float cos_theta = dot(w, wi);
        if (two_sided) cos_theta = std::fabs(cos_theta);
        float sin_theta = std::sqrt(rmax(1.0f - cos_theta * cos_theta, 0.0f));
        auto cos_sub = [](float sa, float ca, float sb, float cb) { return ca > cb ? 1.0f : ca * cb + sa * sb; };
        auto sin_sub = [](float sa, float ca, float sb, float cb) { return ca > cb ? 1.0f : sa * cb - ca * sb; };
        float cos_theta_u = cone_subtended(aabb, p).cos_theta;
        float sin_theta_u = std::sqrt(rmax(1.0f - cos_theta_u * cos_theta_u, 0.0f));
        float sin_theta_o = std::sqrt(rmax(1.0f - cos_theta_o * cos_theta_o, 0.0f));
        float cos_theta_x = cos_sub(sin_theta, cos_theta, sin_theta_o, cos_theta_o);
        float sin_theta_x = sin_sub(sin_theta, cos_theta, sin_theta_o, cos_theta_o);
        float cos_theta_p = cos_sub(sin_theta_x, cos_theta_x, sin_theta_u, cos_theta_u);
        if (cos_theta_p <= cos_theta_e) return 0.0f;
        float imp = phi * cos_theta_p / d2;
        if (n) {
            float cos_theta_i = std::fabs(dot(wi, *n));
            float sin_theta_i = std::sqrt(rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f));
            imp *= cos_sub(sin_theta_i, cos_theta_i, sin_theta_u, cos_theta_u);
        }
        return rmax(imp, 0.0f);
    }
};
struct LightProxy { size_t emitter_id, primitive_idx; LightBounds bounds; };
struct LightBVHNode { long left = -1, right = -1, parent = -1; LightBounds bounds; long light = -1; bool is_leaf() const { return left < 0 && right < 0; } };
struct LightSamplerATS {
    long root = -1;
    std::vector<LightBVHNode> nodes;
    std::vector<LightProxy> lights;
    std::map<std::pair<size_t, size_t>, size_t> query_to_nodes;
    static float max_component(V3 v) { return rmax(rmax(v.x, v.y), v.z); }
    static V3 aabb_offset(const AABB& b, V3 v) {   // structure.rs:811-820
        V3 o = v - b.p_min, s = b.size();
        return {s.x != 0.0f ? o.x / s.x : 0.0f, s.y != 0.0f ? o.y / s.y : 0.0f, s.z != 0.0f ? o.z / s.z : 0.0f};
    }
    // build_bvh (emitter.rs:1117-1262)
    size_t build(size_t index, LightProxy* lt, size_t n) {
        if (n == 1) {
            LightBVHNode nd; nd.bounds = lt[0].bounds; nd.light = (long)index;
            nodes.push_back(nd);
            query_to_nodes[{lt[0].emitter_id, lt[0].primitive_idx}] = nodes.size() - 1;
            return nodes.size() - 1;
        }
        AABB bounds, centroid_bounds;
        for (size_t i = 0; i < n; i++) { bounds = bounds.union_aabb(lt[i].bounds.aabb); centroid_bounds = centroid_bounds.union_vec(lt[i].bounds.aabb.center()); }
        float min_cost = F32_MAX; int min_cost_bucket = -1, min_cost_dim = -1;
        const size_t NB = 12;
        auto bucket_of = [&](const LightProxy& l, int dim) { size_t i = as_usize((float)NB * aabb_offset(centroid_bounds, l.bounds.aabb.center())[dim]); return i < NB - 1 ? i : NB - 1; };
        for (int dim = 0; dim < 3; dim++) {
            if (centroid_bounds.p_max[dim] == centroid_bounds.p_min[dim]) continue;
            std::vector<LightBounds> bb(NB);
            for (size_t i = 0; i < n; i++) { size_t k = bucket_of(lt[i], dim); bb[k] = LightBounds::merge(bb[k], lt[i].bounds); }
            for (size_t i = 0; i + 1 < NB; i++) {
                LightBounds b0, b1;
                for (size_t j = 0; j < i + 1; j++) b0 = LightBounds::merge(b0, bb[j]);
                for (size_t j = i + 1; j < NB; j++) b1 = LightBounds::merge(b1, bb[j]);
                auto momega = [](const LightBounds& b) {
                    float theta_w = rmin(b.theta_o + b.theta_e, PI_F);
                    return 2.0f * PI_F * (1.0f - detmath::cosf_det(b.theta_o))
                         + 1.57079632679489661923f * (2.0f * theta_w * detmath::sinf_det(b.theta_o) - detmath::cosf_det(b.theta_o - 2.0f * theta_w)
                                                      - 2.0f * b.theta_o * detmath::sinf_det(b.theta_o) + detmath::cosf_det(b.theta_o));
                };
                float kr = max_component(bounds.size()) / bounds.size()[dim];
                float c = kr * (b0.phi * momega(b0) * b0.aabb.surface_area() + b1.phi * momega(b1) * b1.aabb.surface_area());
                if (c > 0.0f && c < min_cost) { min_cost = c; min_cost_bucket = (int)i; min_cost_dim = dim; }
            }
        }
        size_t mid;
        if (min_cost_dim == -1) mid = n / 2;
        else {   // itertools::partition: swap the first failing element with the last passing one, repeatedly
            auto pred = [&](const LightProxy& l) { return bucket_of(l, min_cost_dim) <= (size_t)min_cost_bucket; };
            size_t split = 0, front = 0, back = n;
            for (;;) {
                if (front == back) break;
                size_t f = front++;
                if (!pred(lt[f])) {
                    bool found = false;
                    while (back > front) { back--; if (pred(lt[back])) { std::swap(lt[f], lt[back]); found = true; break; } }
                    if (!found) break;
                }
                split++;
            }
            mid = split;
        }
        size_t left = build(index, lt, mid);
        size_t right = build(index + mid, lt + mid, n - mid);
        LightBVHNode nd; nd.left = (long)left; nd.right = (long)right;
        nd.bounds = LightBounds::merge(nodes[left].bounds, nodes[right].bounds);
        nodes.push_back(nd);
        size_t id = nodes.size() - 1;
        nodes[left].parent = (long)id; nodes[right].parent = (long)id;
        return id;
    }
    template <class F> float prob_left(const LightBVHNode& nd, F imp) const {
        float il = imp(nodes[nd.left].bounds), ir = imp(nodes[nd.right].bounds);
        return (il == 0.0f && ir == 0.0f) ? 0.5f : il / (il + ir);
    }
    // sample (emitter.rs:1330-1367)
    template <class F> const LightProxy& sample(float r, F imp, float* pdf_sel) const {
        float pdf = 1.0f; long ni = root;
        for (;;) {
            const LightBVHNode& nd = nodes[ni];
            if (nd.is_leaf()) { *pdf_sel = pdf; return lights[nd.light]; }
            float pl = prob_left(nd, imp);
            if (r < pl) { r = r / pl; ni = nd.left; pdf *= pl; }
            else { r = (r - pl) / (1.0f - pl); ni = nd.right; pdf *= 1.0f - pl; }
        }
    }
    // pdf (emitter.rs:1294-1328)
    template <class F> float pdf(size_t id_emitter, size_t id_primitive, F imp) const {
        long id = (long)query_to_nodes.at({id_emitter, id_primitive});
        float pdf = 1.0f;
        while (nodes[id].parent >= 0) {
            long ip = nodes[id].parent;
            float pl = prob_left(nodes[ip], imp);
            if (nodes[ip].left == id) pdf *= pl; else pdf *= 1.0f - pl;
            id = ip;
        }
        return pdf;
    }
};

// ------------------------------------------------------------------------------------------
// HomogenousVolume + PhaseFunction (src/volume.rs)
struct SampledDistance { float t; Color w; float pdf; bool exited; };
struct Volume {
    Color sigma_a, sigma_s, sigma_t;
    int phase = RL_PHASE_ISOTROPIC;
    float g = 0;
    SampledDistance sample(const Ray& r, float u) const {  // volume.rs:95-135
        float max_t = r.tfar;
        float u3 = u * 3.0f;
        int component = u3 != u3 ? 0 : (u3 <= 0.0f ? 0 : (u3 >= 255.0f ? 255 : (int)u3));  // `as u8` saturates
        u = u * 3.0f - (float)component;
        float sigma_t_c = sigma_t.get(component);
        float t = -detmath::logf_det(1.0f - u) / sigma_t_c;
        float t_min = rmin(t, max_t);
        bool exited = t >= max_t;
        Color tau = t_min * sigma_t;
        Color w = (-tau).exp();
        float pdf;
        if (exited) pdf = (-tau).exp().avg();
        else { mul_assign(w, sigma_s); pdf = (sigma_t * (-tau).exp()).avg(); }
        div_assign(w, pdf);
        return {t_min, w, pdf, exited};
    }
    Color transmittance(float tfar) const { Color tau = sigma_t * tfar; return (-tau).exp(); }  // volume.rs:137-141
    Color phase_eval(V3 w_i, V3 w_o) const {  // volume.rs:19-30
        if (phase == RL_PHASE_ISOTROPIC) return Color::value(1.0f / (PI_F * 4.0f));
        float tmp = 1.0f + g * g + 2.0f * g * dot(w_i, w_o);
        return Color::value(FRAC_1_PI * 0.25f * (1.0f - g * g) / (tmp * std::sqrt(tmp)));
    }
    float phase_pdf(V3 w_i, V3 w_o) const { return phase_eval(w_i, w_o).avg(); }
    void phase_sample(V3 d_in, V2 u, V3* d, Color* weight, float* pdf) const {  // volume.rs:36-67
        if (phase == RL_PHASE_ISOTROPIC) { *d = sample_uniform_sphere(u); *weight = Color::one(); *pdf = 1.0f / (PI_F * 4.0f); return; }
        float cos_t;
        if (std::fabs(g) < 0.000001f) cos_t = 1.0f - 2.0f * u.x;
        else { float sq = (1.0f - g * g) / (1.0f - g + 2.0f * g * u.x); cos_t = (1.0f + g * g - sq * sq) / (2.0f * g); }
        float sin_t = std::sqrt(rmax(1.0f - cos_t * cos_t, 0.0f));
        float sp, cp;
        detmath::sincosf_det(2.0f * PI_F * u.y, &sp, &cp);
        V3 rev = d_in * -1.0f;
        *d = Frame::make(rev).to_world({sin_t * cp, sin_t * sp, cos_t});
        *weight = Color::one();
        *pdf = phase_pdf(d_in, *d);
    }
};

// ------------------------------------------------------------------------------------------
// Intersection (src/structure.rs:926-1059)
struct Intersection {
    float dist; V3 n_g, n_s, p; bool has_uv; V2 uv; int mesh; Frame frame; V3 wi; size_t primitive_id;
};

struct Scene {
    Camera camera;
    std::vector<Mesh> meshes;
    std::vector<std::unique_ptr<Bitmap>> bitmaps;
    bool has_volume = false;
    Volume volume;
    // EmitterSampler (non-ATS): emitters = mesh indices in mesh order
    bool emitters_built = false;
    std::vector<EmitterRec> emitters;
    std::vector<EmitterRec> other_emitters;   // EmittersState::Unbuild(..): point / directional lights in insertion order
    bool has_env = false; Color env_color = Color::zero();
    std::unique_ptr<EnvTexture> env_tex;      // EnvironmentLightColor::Texture when set, else Constant(env_color)
    int env_emitter = -1;
    std::vector<int> mesh_to_emitter;
    Distribution1D emitters_cdf;
    bool want_ats = false;                    // build_emitters(build_ats) (scene.rs:53, cli `-x ats`)
    std::unique_ptr<LightSamplerATS> ats;
    BoundingSphere bsphere;
    // BVHAccel (src/accel.rs:79-113)
    struct Node { AABB aabb; size_t info, count; bool is_leaf() const { return count != 0; } };
    struct TriRef { int id_mesh; size_t id_tri; };
    std::vector<Node> nodes;
    std::vector<TriRef> primitives;
    bool bvh_built = false;

    // Scene::build_emitters(false) (src/scene.rs:53-123)
    void build_emitters() {
        AABB aabb;
        for (const Mesh& m : meshes) aabb = aabb.union_aabb(m.compute_aabb());
        aabb = aabb.union_vec(camera.position());
        V3 c = aabb.center();
        bsphere.center = c;
        bsphere.radius = magnitude(c - aabb.p_max);
        emitters.clear();
        env_emitter = -1;
        mesh_to_emitter.assign(meshes.size(), -1);
        for (size_t i = 0; i < meshes.size(); i++) if (meshes[i].is_light) {
            mesh_to_emitter[i] = (int)emitters.size();
            EmitterRec e; e.kind = EM_MESH; e.mesh = (int)i; emitters.push_back(e);
        }
        BoundingSphere big = bsphere; big.radius *= 1.1f;       // Emitter::preprocess
        if (has_env) { EmitterRec e; e.kind = EM_ENV; e.c = env_color; e.bsphere = big; env_emitter = (int)emitters.size(); emitters.push_back(e); }
        for (EmitterRec e : other_emitters) { e.bsphere = big; emitters.push_back(e); }
        emitters_built = true;
        if (emitters.empty()) return;
        std::vector<float> flux;
        for (const EmitterRec& e : emitters) {
            Color f;
            switch (e.kind) {
                case EM_MESH: f = mesh_flux(meshes[e.mesh]); break;
                case EM_POINT: f = e.c * 4.0f * PI_F; break;                                  // emitter.rs:238-240
                case EM_DIRECTIONAL: f = (PI_F * powi(e.bsphere.radius, 2)) * e.c; break;      // emitter.rs:163-167
                default:                                                                        // emitter.rs:519-531
                    if (env_tex) f = Color::value(PI_F * powi(e.bsphere.radius, 2) * env_tex->cdf.marginal.func_int);
                    else f = PI_F * powi(e.bsphere.radius, 2) * e.c;
                    break;
            }
            flux.push_back(f.channel_max());
        }
        emitters_cdf = Distribution1D::normalize(flux);
        ats.reset();
        if (want_ats) build_ats();
    }
    // EmitterSampler::build_ats -> LightSamplerATS::new (emitter.rs:1264-1292) with Mesh::convert_light_proxy (726-781)
    void build_ats() {
        ats.reset(new LightSamplerATS());
        for (size_t e = 0; e < emitters.size(); e++) {
            assert(emitters[e].kind == EM_MESH);      // assert!(e.is_surface())
            const Mesh& m = meshes[emitters[e].mesh];
            for (size_t i = 0; i < m.n_tris(); i++) {
                V3 v0 = m.vertices[m.indices[3 * i]], v1 = m.vertices[m.indices[3 * i + 1]], v2 = m.vertices[m.indices[3 * i + 2]];
                V3 n = cross(v1 - v0, v2 - v0);
                LightProxy lp; lp.emitter_id = e; lp.primitive_idx = i;
                LightBounds& b = lp.bounds;
                b.w = normalize(n);
                b.theta_o = 0.0f; b.theta_e = 1.57079632679489661923f;
                V2 uvc{0, 0};                                           // "For now we interpolate at the middle" (emitter.rs:741-750)
                if (m.has_uv) { V2 a0 = m.uv[m.indices[3 * i]], a1 = m.uv[m.indices[3 * i + 1]], a2 = m.uv[m.indices[3 * i + 2]]; uvc = {((a0.x + a1.x) + a2.x) / 3.0f, ((a0.y + a1.y) + a2.y) / 3.0f}; }
                b.phi = m.emit(m.has_uv, uvc).channel_max() * magnitude(n) * 0.5f;
                b.aabb = AABB().union_vec(v0).union_vec(v1).union_vec(v2);
                b.cos_theta_o = detmath::cosf_det(b.theta_o); b.cos_theta_e = detmath::cosf_det(b.theta_e);
                b.two_sided = false; b.number_lights = 1; b.phi_sqr = powi(b.phi, 2);
                ats->lights.push_back(lp);
            }
        }
        if (ats->lights.empty()) { ats.reset(); return; }
        ats->root = (long)ats->build(0, ats->lights.data(), ats->lights.size());
    }
    float emitter_pdf(int mesh_id) const { return emitters_cdf.pdf((size_t)mesh_to_emitter[mesh_id]); }  // emitter.rs:1510-1526
    // EmitterSampler::direct_pdf (emitter.rs:1566-1575)
    PDF direct_pdf(int mesh_id, const LightSamplingPDF& ls, const V3* n = nullptr, size_t id_primitive = 0) const {
        if (!ats) return mesh_direct_pdf(meshes[mesh_id], ls).mul(emitter_pdf(mesh_id));
        // with the light tree: pdf of the triangle x probability of reaching its leaf (emitter.rs:1595-1600)
        const Mesh& m = meshes[mesh_id];
        float cos_light = rmax(dot(ls.n, -ls.dir), 0.0f);
        PDF tri = PDF::solid_angle(0.0f);
        if (cos_light != 0.0f) {   // Mesh::direct_pdf_tri (emitter.rs:581-589), pdf_tri (geometry.rs:226-234)
            float geom = cos_light / magnitude2(ls.p - ls.o);
            V3 v0 = m.vertices[m.indices[3 * id_primitive]], v1 = m.vertices[m.indices[3 * id_primitive + 1]], v2 = m.vertices[m.indices[3 * id_primitive + 2]];
            float area_tri = magnitude(cross(v1 - v0, v2 - v0)) * 0.5f;
            tri = PDF::solid_angle((1.0f / area_tri) / geom);
        }
        V3 o = ls.o;
        return tri.mul(ats->pdf((size_t)mesh_to_emitter[mesh_id], id_primitive, [&](const LightBounds& b) { return b.importance_point(o, n); }));
    }
    PDF direct_pdf_env(V3 d) const { return PDF::solid_angle(env_tex ? env_tex->pdf(d) : 1.0f / (PI_F * 4.0f)).mul(emitters_cdf.pdf((size_t)env_emitter)); }
    Color environment_luminance(V3 d) const { return !has_env ? Color::zero() : (env_tex ? env_tex->eval(d) : env_color); }   // scene.rs:125-130
    // EmitterSampler::sample_light (emitter.rs:1604-1620); LightSampling.emitter = index into `emitters`
    LightSampling sample_light(V3 p, const V3* n, float r_sel, float r, V2 uv) const {
        if (ats) {   // emitter.rs:1621-1639: the tree picks (emitter, triangle); Mesh::direct_sample_tri (emitter.rs:609-650)
            float pdf_sel;
            const LightProxy& lp = ats->sample(r_sel, [&](const LightBounds& b) { return b.importance_point(p, n); }, &pdf_sel);
            const Mesh& m = meshes[emitters[lp.emitter_id].mesh];
            Mesh::SampledPosition sp = m.sample_tri(lp.primitive_idx, uv);
            V3 d = sp.p - p;
            float dist = magnitude(d);
            if (dist != 0.0f) d = d / dist;
            float geom = dist != 0.0f ? rmax(dot(sp.n, -d), 0.0f) / (dist * dist) : 0.0f;
            float pdf_area = sp.pdf.value();
            PDF pdf = sp.pdf.as_solid_angle_geom(geom);
            Color weight = pdf.is_zero() ? Color::zero() : m.emit(sp.has_uv, sp.uv) * geom / pdf_area;
            LightSampling res = {(int)lp.emitter_id, pdf, sp.p, sp.n, d, weight, sp.has_uv, sp.uv};
            div_assign(res.weight, pdf_sel);
            res.pdf = res.pdf.mul(pdf_sel);
            return res;
        }
        size_t id = emitters_cdf.sample_discrete(r_sel);
        float pdf_sel = emitters_cdf.pdf(id);
        const EmitterRec& e = emitters[id];
        LightSampling res;
        if (e.kind == EM_MESH) res = mesh_direct_sample(meshes[e.mesh], p, r, uv);
        else if (e.kind == EM_POINT) {                                     // emitter.rs:194-213
            V3 d = e.v - p;
            float dist = magnitude(d);
            d = d / dist;
            res = {-1, PDF::discrete(1.0f), e.v, {0, 0, 0}, d, e.c / powi(dist, 2)};
        } else if (e.kind == EM_DIRECTIONAL) {                             // emitter.rs:116-134
            V3 lp = p - e.bsphere.radius * e.v;
            res = {-1, PDF::discrete(1.0f), lp, e.v, -e.v, e.c};
        } else {                                                           // emitter.rs:473-518
            V3 d; float pdf; Color lum = e.c;
            if (env_tex) env_tex->sample_direction(uv, &d, &lum, &pdf);
            else { d = sample_uniform_sphere(uv); pdf = 1.0f / (PI_F * 4.0f); }
            float t;
            if (!bsphere_intersect(e.bsphere, Ray::make(p, d), &t)) res = {-1, PDF::solid_angle(pdf), {0, 0, 0}, {0, 0, 0}, d, Color::zero()};
            else {
                V3 lp = p + d * t;
                V3 n = normalize(e.bsphere.center - lp);
                res = {-1, PDF::solid_angle(pdf), lp, n, d, lum / pdf};
            }
        }
        res.emitter = (int)id;
        div_assign(res.weight, pdf_sel);
        res.pdf = res.pdf.mul(pdf_sel);
        return res;
    }

    // ---- BVHAccel::new + subdivide_node (src/accel.rs:115-239)
    struct CachedAABB { AABB aabb; TriRef info; };
    static AABB compute_aabb(const std::vector<CachedAABB>& a, size_t start, size_t count) {
        AABB r; for (size_t i = 0; i < count; i++) r = r.union_aabb(a[i + start].aabb); return r;
    }
    void subdivide_node(size_t id_node, std::vector<CachedAABB>& aabbs) {
        if (nodes[id_node].count <= 2) return;
        size_t nb_prim = nodes[id_node].count, first_prim = nodes[id_node].info;
        nodes[id_node].count = 0;
        nodes[id_node].info = nodes.size();
        size_t best_pos = 0; float best_cost = F32_INF; int best_axis = 3;
        {
            std::vector<float> scores(nb_prim - 1, 0.0f);
            for (int o = 0; o < 3; o++) {
                // slice::sort_by with `if a < b {Less} else {Greater}`: Rust's stable sort only asks
                // "is a less than b", so this equals a stable sort by centre[o].
                std::stable_sort(aabbs.begin() + first_prim, aabbs.begin() + first_prim + nb_prim,
                                 [o](const CachedAABB& a, const CachedAABB& b) { return a.aabb.center()[o] < b.aabb.center()[o]; });
                AABB tmp;
                for (size_t id = 0; id < nb_prim - 1; id++) {
                    size_t id_left = nb_prim - id - 1;
                    tmp = tmp.union_aabb(aabbs[id_left + first_prim].aabb);
                    scores[id_left - 1] = tmp.surface_area() * (float)(id + 1);
                }
                tmp = AABB();
                for (size_t id = 0; id < nb_prim - 1; id++) {
                    tmp = tmp.union_aabb(aabbs[id + first_prim].aabb);
                    scores[id] += tmp.surface_area() * (float)(id + 1);
                    if (scores[id] < best_cost) { best_cost = scores[id]; best_axis = o; best_pos = id + 1; }
                }
            }
        }
        if (best_axis < 3) {  // (indexing a Vector3 with 3 would panic in the reference; NaN boxes only)
            int ax = best_axis;
            std::stable_sort(aabbs.begin() + first_prim, aabbs.begin() + first_prim + nb_prim,
                             [ax](const CachedAABB& a, const CachedAABB& b) { return a.aabb.center()[ax] < b.aabb.center()[ax]; });
        }
        size_t offset = (best_pos == nb_prim || best_pos == 0) ? std::max((size_t)((float)nb_prim * 0.5f), (size_t)1) : best_pos;
        Node left{compute_aabb(aabbs, first_prim, offset), first_prim, offset};
        Node right{compute_aabb(aabbs, first_prim + offset, nb_prim - offset), first_prim + offset, nb_prim - offset};
        size_t id_left = nodes.size();
        nodes.push_back(left);
        nodes.push_back(right);
        subdivide_node(id_left, aabbs);
        subdivide_node(id_left + 1, aabbs);
    }
    void build_bvh() {
        AABB root_aabb;
        std::vector<CachedAABB> cached;
        for (size_t m = 0; m < meshes.size(); m++)
            for (size_t i = 0; i < meshes[m].n_tris(); i++) {
                cached.push_back({meshes[m].compute_aabb_tri(i), {(int)m, i}});
                root_aabb = root_aabb.union_aabb(cached.back().aabb);
            }
        nodes.clear();
        nodes.push_back({root_aabb, 0, cached.size()});
        subdivide_node(0, cached);
        primitives.clear();
        for (auto& c : cached) primitives.push_back(c.info);
        bvh_built = true;
    }

    // ---- BVHAccel::intersect (src/accel.rs:243-288)
    bool bvh_intersect(size_t id_node, const Ray& ray, IntersectionUV* its, TriRef* res) const {
        const Node& node = nodes[id_node];
        if (node.is_leaf()) {
            bool found = false;
            for (size_t k = 0; k < node.count; k++) {
                const TriRef& e = primitives[node.info + k];
                if (meshes[e.id_mesh].intersection_tri(e.id_tri, ray.o, ray.d, its)) { *res = e; found = true; }
            }
            return found;
        }
        size_t id1 = node.info, id2 = node.info + 1;
        float d1, d2;
        if (!nodes[id1].aabb.intersect(ray, &d1)) d1 = F32_INF;
        if (!nodes[id2].aabb.intersect(ray, &d2)) d2 = F32_INF;
        if (d1 > d2) { std::swap(d1, d2); std::swap(id1, id2); }
        bool found = false;
        if (d1 < its->t) found = bvh_intersect(id1, ray, its, res);
        if (d2 < its->t) { TriRef r2; if (bvh_intersect(id2, ray, its, &r2)) { *res = r2; found = true; } }
        return found;
    }
    // Acceleration::trace without fill_intersection (src/accel.rs:292-315)
    bool trace_uv(const Ray& ray, IntersectionUV* its, TriRef* res) const {
        float t;
        if (!nodes[0].aabb.intersect(ray, &t)) return false;
        *its = {F32_MAX, {0, 0, 0}, {0, 0, 0}, 0, 0};
        bvh_intersect(0, ray, its, res);
        return its->t != F32_MAX;
    }
    // Intersection::fill_intersection (src/structure.rs:965-1059)
    Intersection fill_intersection(int mesh_id, size_t tri, float hu, float hv, const Ray& ray, V3 n_g, float dist, V3 p) const {
        const Mesh& mesh = meshes[mesh_id];
        uint32_t i0 = mesh.indices[3 * tri], i1 = mesh.indices[3 * tri + 1], i2 = mesh.indices[3 * tri + 2];
        V3 n_s;
        if (mesh.has_normals) {
            V3 d0 = mesh.normals[i0], d1 = mesh.normals[i1], d2 = mesh.normals[i2];
            V3 ns = d0 * (1.0f - hu - hv) + d1 * hu + d2 * hv;
            if (dot(n_g, ns) < 0.0f) n_g = -n_g;
            float l = dot(ns, ns);
            if (l == 0.0f) n_s = n_g;
            else if (l != 1.0f) n_s = ns / std::sqrt(l);
            else n_s = ns;
        } else n_s = n_g;
        if (mesh.bsdf.is_twosided() && !mesh.is_light && dot(ray.d, n_s) > 0.0f) { n_s = {-n_s.x, -n_s.y, -n_s.z}; n_g = {-n_g.x, -n_g.y, -n_g.z}; }
        Intersection its;
        its.has_uv = mesh.has_uv; its.uv = {0, 0};
        if (mesh.has_uv) {
            V2 a0 = mesh.uv[i0], a1 = mesh.uv[i1], a2 = mesh.uv[i2];
            float w0 = 1.0f - hu - hv;
            its.uv = {a0.x * w0 + a1.x * hu + a2.x * hv, a0.y * w0 + a1.y * hu + a2.y * hv};
        }
        its.frame = Frame::make(n_s);
        its.wi = its.frame.to_local(-ray.d);
        its.dist = dist; its.n_g = n_g; its.n_s = n_s; its.p = p; its.mesh = mesh_id; its.primitive_id = tri;
        return its;
    }
    bool trace(const Ray& ray, Intersection* out) const {
        IntersectionUV its; TriRef res{0, 0};
        if (!trace_uv(ray, &its, &res)) return false;
        *out = fill_intersection(res.id_mesh, res.id_tri, its.u, its.v, ray, its.n, its.t, its.p);
        return true;
    }
    // Acceleration::visible (src/accel.rs:316-343)
    bool visible(V3 p0, V3 p1) const {
        const float SHADOW_EPS = 0.00001f;
        V3 d = p1 - p0;
        float length = magnitude(d);
        d = d / length;
        IntersectionUV its = {length * (1.0f - SHADOW_EPS), {0, 0, 0}, {0, 0, 0}, 0, 0};
        Ray ray = {p0, d, EPSILON, length * (1.0f - SHADOW_EPS)};
        float t;
        if (!nodes[0].aabb.intersect(ray, &t)) return false;
        TriRef r;
        return !bvh_intersect(0, ray, &its, &r);
    }
};

// ------------------------------------------------------------------------------------------
// Path graph (src/paths/{path,vertex,edge}.rs)
struct Edge {
    bool has_dist = false; float dist = 0;
    V3 d;
    int v0 = -1, v1 = -1;  // vertices (v1 = -1: None)
    PDF pdf_direction = PDF::solid_angle(0);
    Color weight;
    bool has_contrib = false; Color contrib;
    float rr_weight = 1;
    int id_sampling = 0;
};
struct Vertex {
    enum Kind { Sensor, Surface, Light, Volume } kind;
    V2 uv_px;          // Sensor
    V3 pos;            // Sensor / Light / Volume
    Intersection its;  // Surface
    V3 n;              // Light
    int emitter = -1;  // Light (mesh id)
    V3 d_in;           // Volume
    int edge_out[2] = {-1, -1};
    int n_out = 0;
    V3 position() const { return kind == Surface ? its.p : pos; }
    bool on_surface() const { return kind == Surface || kind == Light; }
};
struct Path {
    std::vector<Vertex> vertices;
    std::vector<Edge> edges;
    void clear() { vertices.clear(); edges.clear(); }
};

struct PathParams {
    bool has_min = true; uint32_t min_depth = 0;
    bool has_max = false; uint32_t max_depth = 0;
    bool has_rr = true; uint32_t rr_depth = 0;
    int strategy = RL_STRATEGY_ALL;
    bool single_scattering = false;
    int eval_order = 0;
};
struct Counters { uint64_t vertices = 0, extension_rays = 0, shadow_rays = 0, draws = 0, samples = 0; };

struct PathTracer {
    const Scene& scene;
    PathParams prm;
    Counters cnt;
    Path path;
    explicit PathTracer(const Scene& s, const PathParams& p) : scene(s), prm(p) {}

    bool on_light_source(int v) const {  // vertex.rs:60-66
        const Vertex& vx = path.vertices[v];
        if (vx.kind == Vertex::Surface) return scene.meshes[vx.its.mesh].is_light;
        return vx.kind == Vertex::Light;
    }
    Color vertex_contribution(int v, const Edge& e) const {  // vertex.rs:69-82
        const Vertex& vx = path.vertices[v];
        if (vx.kind == Vertex::Surface) {
            if (dot(vx.its.n_s, -e.d) >= 0.0f) return scene.meshes[vx.its.mesh].emit(vx.its.has_uv, vx.its.uv);
            return Color::zero();
        }
        if (vx.kind == Vertex::Light) { const EmitterRec& em = scene.emitters[vx.emitter]; return em.kind == EM_MESH ? scene.meshes[em.mesh].emit(vx.its.has_uv, vx.its.uv) : em.c; }  // emitter.eval(-d, uv): the sampled point's uv, kept in its.uv of the light vertex
        return Color::zero();
    }
    bool next_on_light_source(const Edge& e) const { return e.v1 >= 0 ? on_light_source(e.v1) : scene.has_env; }  // edge.rs:191-197
    Color edge_contribution(const Edge& e) const {  // edge.rs:201-210
        if (e.v1 >= 0) {
            if (e.has_contrib) return e.contrib * e.weight * e.rr_weight;
            return e.weight * e.rr_weight * vertex_contribution(e.v1, e);
        }
        return e.weight * e.rr_weight * scene.environment_luminance(e.d);  // scene.enviroment_luminance(self.d)
    }

    // Edge::from_ray (edge.rs:65-189)
    void edge_from_ray(const Ray& ray, int org, PDF pdf, Color weight, float rr, Sampler& sampler, int id_sampling, int* edge_out, int* vertex_out) {
        Edge e;
        e.d = ray.d; e.v0 = org; e.pdf_direction = pdf; e.weight = weight; e.rr_weight = rr; e.id_sampling = id_sampling;
        int eid = (int)path.edges.size();
        path.edges.push_back(e);
        *edge_out = eid;
        Intersection its;
        cnt.extension_rays++;
        bool hit = scene.trace(ray, &its);
        if (!hit) {
            if (scene.has_volume) {
                SampledDistance mrec = scene.volume.sample(ray, sampler.next());
                Vertex nv; nv.kind = Vertex::Volume;
                nv.pos = ray.o + ray.d * mrec.t;
                nv.d_in = -ray.d;
                int vid = (int)path.vertices.size();
                path.vertices.push_back(nv);
                Edge& ed = path.edges[eid];
                ed.has_dist = true; ed.dist = mrec.t; ed.v1 = vid; mul_assign(ed.weight, mrec.w);
                *vertex_out = vid;
                return;
            }
            *vertex_out = -1;
            return;
        }
        float intersection_distance = its.dist;
        Vertex nv;
        bool has_mrec = false; SampledDistance mrec{};
        if (scene.has_volume) {
            Ray ray_med = ray; ray_med.tfar = intersection_distance;
            mrec = scene.volume.sample(ray_med, sampler.next());
            has_mrec = true;
            if (!mrec.exited) {
                intersection_distance = mrec.t;
                nv.kind = Vertex::Volume; nv.pos = ray.o + ray.d * mrec.t; nv.d_in = -ray.d;
            } else { nv.kind = Vertex::Surface; nv.its = its; }
        } else { nv.kind = Vertex::Surface; nv.its = its; }
        int vid = (int)path.vertices.size();
        path.vertices.push_back(nv);
        Edge& ed = path.edges[eid];
        ed.has_dist = true; ed.dist = intersection_distance; ed.v1 = vid;
        if (has_mrec) mul_assign(ed.weight, mrec.w);
        *vertex_out = vid;
    }

    // DirectionalSamplingStrategy::{bounce,sample} (strategies/directional.rs:13-257); returns next vertex or -1
    int directional_sample(int vid, Color& throughput, Sampler& sampler, uint32_t depth) {
        int edge = -1, nv = -1;
        Vertex::Kind kind = path.vertices[vid].kind;
        if (kind == Vertex::Sensor) {
            Ray ray = scene.camera.generate(path.vertices[vid].uv_px);
            edge_from_ray(ray, vid, PDF::solid_angle(1.0f), Color::one(), 1.0f, sampler, 0, &edge, &nv);
        } else if (kind == Vertex::Surface) {
            Intersection its = path.vertices[vid].its;
            const BSDF& bsdf = scene.meshes[its.mesh].bsdf;
            SampledDirection sd;
            V2 s2 = sampler.next2d();
            if (bsdf.sample(its.has_uv, its.uv, its.wi, s2, &sd)) {
                V3 d_out_global = its.frame.to_world(sd.d);
                mul_assign(throughput, sd.weight);
                if (throughput.is_zero()) return -1;
                bool do_rr = prm.has_rr ? prm.rr_depth <= depth : true;
                float rr_weight = 1.0f;
                if (do_rr) {
                    float q = rmin(throughput.channel_max(), 0.95f);
                    if (q < sampler.next()) return -1;
                    rr_weight = 1.0f / q;
                }
                throughput.scale(rr_weight);
                Ray ray = {its.p, d_out_global, EPSILON, F32_MAX};  // Ray::spawn_ray
                edge_from_ray(ray, vid, sd.pdf, sd.weight, rr_weight, sampler, 0, &edge, &nv);
            }
        } else if (kind == Vertex::Volume) {
            V3 d_in = path.vertices[vid].d_in, pos = path.vertices[vid].pos;
            V3 d; Color w; float pdf;
            scene.volume.phase_sample(d_in, sampler.next2d(), &d, &w, &pdf);
            mul_assign(throughput, w);
            if (throughput.is_zero()) return -1;
            bool do_rr = prm.has_rr ? prm.rr_depth <= depth : true;
            float rr_weight = 1.0f;
            if (do_rr) {
                float q = rmin(throughput.channel_max(), 0.95f);
                if (q < sampler.next()) return -1;
                rr_weight = 1.0f / q;
            }
            throughput.scale(rr_weight);
            Ray ray = Ray::make(pos, d);
            edge_from_ray(ray, vid, PDF::solid_angle(pdf), w, rr_weight, sampler, 0, &edge, &nv);
        }
        if (edge >= 0) { Vertex& v = path.vertices[vid]; v.edge_out[v.n_out++] = edge; }
        return nv;
    }
    // DirectionalSamplingStrategy::pdf (directional.rs:258-304): Option<f32>, None => has=false
    bool directional_pdf(int vid, int eid, float* out) const {
        const Edge& e = path.edges[eid];
        if (!next_on_light_source(e)) return false;
        const Vertex& v = path.vertices[vid];
        if (v.kind == Vertex::Surface) {
            const BSDF& bsdf = scene.meshes[v.its.mesh].bsdf;
            if (bsdf.is_smooth()) return false;
            PDF p = bsdf.pdf(v.its.has_uv, v.its.uv, v.its.wi, v.its.frame.to_local(e.d), DomSolidAngle);
            *out = p.value();
            return true;
        }
        if (v.kind == Vertex::Volume) { *out = scene.volume.phase_pdf(v.d_in, e.d); return true; }
        if (v.kind == Vertex::Sensor) { *out = 1.0f; return true; }
        return false;
    }

    // LightSamplingStrategy::sample (strategies/emitters.rs:95-248)
    void light_sample(int vid, Sampler& sampler) {
        Vertex::Kind kind = path.vertices[vid].kind;
        if (kind != Vertex::Surface && kind != Vertex::Volume) return;
        V3 p;
        if (kind == Vertex::Surface) {
            const Intersection& its = path.vertices[vid].its;
            if (scene.meshes[its.mesh].bsdf.is_smooth()) return;
            p = its.p;
        } else p = path.vertices[vid].pos;
        float a = sampler.next();
        float b = sampler.next();
        V2 c = sampler.next2d();
        V3 n_s = kind == Vertex::Surface ? path.vertices[vid].its.n_s : V3{0, 0, 0};
        LightSampling lr = scene.sample_light(p, kind == Vertex::Surface ? &n_s : nullptr, a, b, c);   // Some(&its.n_s) | None (emitters.rs:118-124, 186-192)
        cnt.shadow_rays++;
        bool vis = scene.visible(p, lr.p);
        if (!(lr.is_valid() && vis)) return;
        Color weight;
        if (kind == Vertex::Surface) {
            const Intersection& its = path.vertices[vid].its;
            weight = scene.meshes[its.mesh].bsdf.eval(its.has_uv, its.uv, its.wi, its.frame.to_local(lr.d), DomSolidAngle);
        } else weight = scene.volume.phase_eval(path.vertices[vid].d_in, lr.d);
        if (scene.has_volume) {
            V3 dd = lr.p - p;
            float tfar = dot(dd, lr.d);
            mul_assign(weight, scene.volume.transmittance(tfar));
        }
        Vertex lv; lv.kind = Vertex::Light; lv.pos = lr.p; lv.n = lr.n; lv.emitter = lr.emitter; lv.its.has_uv = lr.has_uv; lv.its.uv = lr.uv;
        int lvid = (int)path.vertices.size();
        path.vertices.push_back(lv);
        // Edge::from_vertex (edge.rs:27-63)
        Edge e;
        V3 d = path.vertices[lvid].position() - path.vertices[vid].position();
        float dist = magnitude(d);
        d = d / dist;
        e.has_dist = true; e.dist = dist; e.d = d; e.v0 = vid; e.v1 = lvid;
        e.pdf_direction = lr.pdf; e.weight = weight; e.has_contrib = true; e.contrib = lr.weight; e.rr_weight = 1.0f; e.id_sampling = 1;
        int eid = (int)path.edges.size();
        path.edges.push_back(e);
        Vertex& v = path.vertices[vid];
        v.edge_out[v.n_out++] = eid;
    }
    // LightSamplingStrategy::{pdf,pdf_emitter} (emitters.rs:10-92, 250-282)
    bool light_pdf(int vid, int eid, float* out) const {
        const Edge& e = path.edges[eid];
        if (!next_on_light_source(e)) return false;
        const Vertex& v = path.vertices[vid];
        if (v.kind == Vertex::Surface) { if (scene.meshes[v.its.mesh].bsdf.is_smooth()) return false; }
        else if (v.kind != Vertex::Volume) return false;
        V3 o = v.position();
        if (e.v1 < 0) {   // pdf_emitter with no next vertex: the environment (emitters.rs:18-46)
            if (!scene.has_env) return false;
            *out = scene.direct_pdf_env(e.d).value();   // LightSamplingPDF.dir = edge.d
            return true;
        }
        const Vertex& nx = path.vertices[e.v1];
        if (nx.kind == Vertex::Surface) {
            PDF p = scene.direct_pdf(nx.its.mesh, {o, nx.its.p, nx.its.n_g, e.d}, nullptr, nx.its.primitive_id);   // n = None (emitters.rs:52-57)
            *out = p.value();
            return true;
        }
        if (nx.kind == Vertex::Light) {   // (not reached by the path integrator: NEE edges never ask for the light pdf)
            const EmitterRec& em = scene.emitters[nx.emitter];
            if (em.kind != EM_MESH) return false;
            PDF p = scene.direct_pdf(em.mesh, {o, nx.pos, nx.n, e.d});
            *out = p.value();
            return true;
        }
        return false;
    }

    bool use_light_strategy() const { return prm.strategy == RL_STRATEGY_ALL || prm.strategy == RL_STRATEGY_EMITTER; }

    // generate (strategies/mod.rs:35-80) specialised to the chain TechniquePathTracing produces
    void generate(int root, Sampler& sampler) {
        int curr = root;
        Color thr = Color::one();
        uint32_t depth = 1;
        while (curr >= 0) {
            int next = -1;
            bool expand = prm.has_max ? depth < prm.max_depth : true;  // path.rs:28-30
            if (depth >= ORC_DEPTH_CAP) expand = false;
            if (expand) {
                if (path.vertices[curr].kind != Vertex::Sensor) cnt.vertices++;
                Color t = thr;
                int nv = directional_sample(curr, t, sampler, depth);
                if (nv >= 0) { next = nv; thr = t; }
                if (use_light_strategy()) light_sample(curr, sampler);
            }
            curr = next;
            depth++;
        }
    }

    // TechniquePathTracing::evalute_edge (explicit/path.rs:37-111)
    Color evaluate_edge(uint32_t curr_depth, int vid, int eid) const {
        const Edge& e = path.edges[eid];
        Color contrib = edge_contribution(e);
        if (prm.strategy == RL_STRATEGY_BSDF && e.id_sampling != 0) contrib = Color::zero();
        if (prm.strategy == RL_STRATEGY_EMITTER && e.id_sampling != 1) contrib = Color::zero();
        bool add_contrib = prm.has_min ? curr_depth >= prm.min_depth : true;
        if (!contrib.is_zero() && add_contrib) {
            float weight = 1.0f;
            if (prm.strategy == RL_STRATEGY_ALL && e.pdf_direction.kind == PDF::SolidAngle) {
                float v = e.pdf_direction.v;
                float total = 0.0f;
                for (int id = 0; id < 2; id++) {
                    float pdf;
                    if (id == e.id_sampling) pdf = v;
                    else {
                        bool has = id == 0 ? directional_pdf(vid, eid, &pdf) : light_pdf(vid, eid, &pdf);
                        if (!has) pdf = 0.0f;
                    }
                    total = total + pdf;
                }
                weight = v / total;
            }
            return contrib * weight;
        }
        return Color::zero();
    }
    // TechniquePathTracing::evaluate (explicit/path.rs:113-184), inner-first recursion
    Color evaluate(uint32_t curr_depth, int vid) const {
        const Vertex& v = path.vertices[vid];
        if (prm.single_scattering && v.on_surface()) return Color::zero();
        Color l_i = Color::zero();
        if (v.kind == Vertex::Surface || v.kind == Vertex::Volume) {
            for (int k = 0; k < v.n_out; k++) {
                int eid = v.edge_out[k];
                add_assign(l_i, evaluate_edge(curr_depth, vid, eid));
                const Edge& e = path.edges[eid];
                if (e.v1 >= 0) add_assign(l_i, e.weight * e.rr_weight * evaluate(curr_depth + 1, e.v1));
            }
        } else if (v.kind == Vertex::Sensor) {
            const Edge& e = path.edges[v.edge_out[0]];
            bool add_contrib = prm.has_min ? curr_depth >= prm.min_depth : true;
            Color contrib = edge_contribution(e);
            if (!contrib.is_zero() && add_contrib) add_assign(l_i, contrib);
            if (e.v1 >= 0) add_assign(l_i, e.weight * e.rr_weight * evaluate(curr_depth + 1, e.v1));
        }
        return l_i;
    }
    // Front-to-back evaluation in the order the GPU wavefront accumulates (DESIGN.md §Radiance order):
    // at vertex k the NEE term is added before the emission term of the directional edge.
    Color evaluate_forward(int root) const {
        Color L = Color::zero();
        Color beta = Color::one();
        int vid = root; uint32_t curr_depth = 0;
        while (vid >= 0) {
            const Vertex& v = path.vertices[vid];
            bool zeroed = prm.single_scattering && v.on_surface();
            int next = -1;
            if (v.kind == Vertex::Sensor) {
                const Edge& e = path.edges[v.edge_out[0]];
                bool add_contrib = prm.has_min ? curr_depth >= prm.min_depth : true;
                Color contrib = edge_contribution(e);
                if (!zeroed && !contrib.is_zero() && add_contrib) add_assign(L, contrib);
                if (e.v1 >= 0) { beta = e.weight * e.rr_weight; next = e.v1; }
            } else if (v.kind == Vertex::Surface || v.kind == Vertex::Volume) {
                int e_d = -1, e_l = -1;
                for (int k = 0; k < v.n_out; k++) { if (path.edges[v.edge_out[k]].id_sampling == 0) e_d = v.edge_out[k]; else e_l = v.edge_out[k]; }
                if (!zeroed) {
                    if (e_l >= 0) add_assign(L, beta * evaluate_edge(curr_depth, vid, e_l));
                    if (e_d >= 0) add_assign(L, beta * evaluate_edge(curr_depth, vid, e_d));
                }
                if (zeroed) break;  // the whole subtree contributes 0 (path.rs:122-124)
                if (e_d >= 0 && path.edges[e_d].v1 >= 0) { const Edge& e = path.edges[e_d]; beta = beta * (e.weight * e.rr_weight); next = e.v1; }
            }
            vid = next; curr_depth++;
        }
        return L;
    }

    // IntegratorPathTracing::compute_pixel (explicit/path.rs:198-237)
    Color compute_pixel(uint32_t ix, uint32_t iy, Sampler& sampler) {
        path.clear();
        Vertex root; root.kind = Vertex::Sensor;
        float u = (float)ix + sampler.next();
        float v = (float)iy + sampler.next();
        root.uv_px = {u, v};
        root.pos = scene.camera.position();
        path.vertices.push_back(root);
        generate(0, sampler);
        cnt.samples++;
        if (path.vertices[0].n_out == 0) return Color::zero();  // sensor not expanded (max_depth <= 1): edge_out.unwrap() would panic in the reference
        return prm.eval_order == 0 ? evaluate(0, 0) : evaluate_forward(0);
    }
};

// ------------------------------------------------------------------------------------------
// mis_weight: power heuristic with the reference's guards (src/integrators/mod.rs:462-478)
static inline float mis_weight(float pdf_a, float pdf_b) {
    if (pdf_a == 0.0f) return 0.0f;
    if (!is_finite(pdf_a) || !is_finite(pdf_b)) return 0.0f;
    float w = (pdf_a * pdf_a) / (pdf_a * pdf_a + pdf_b * pdf_b);
    return is_finite(w) ? w : 0.0f;
}

// IntegratorAO::compute_pixel (src/integrators/ao.rs:20-70)
struct AOParams { bool has_max_distance = true; float max_distance = 1.0f; bool normal_correction = false; };
static Color ao_compute_pixel(const Scene& scene, const AOParams& ap, uint32_t ix, uint32_t iy, Sampler& sampler, Counters& cnt) {
    float u = (float)ix + sampler.next();
    float v = (float)iy + sampler.next();
    Ray ray = scene.camera.generate({u, v});
    cnt.samples++;
    Intersection its;
    cnt.extension_rays++;
    if (!scene.trace(ray, &its)) return Color::zero();
    if (!ap.normal_correction && its.wi.z <= 0.0f) return Color::zero();
    bool flipped = ap.normal_correction && its.wi.z <= 0.0f;
    V3 d_local = cosine_sample_hemisphere(sampler.next2d());
    V3 d_world = flipped ? its.frame.to_world(-d_local) : its.frame.to_world(d_local);
    Ray r2 = {its.p, d_world, EPSILON, F32_MAX};
    Intersection n2;
    cnt.extension_rays++;
    if (!scene.trace(r2, &n2)) return Color::one();
    if (!ap.has_max_distance) return Color::zero();
    return n2.dist > ap.max_distance ? Color::one() : Color::zero();
}

// IntegratorDirect::compute_pixel (src/integrators/direct.rs:21-233)
struct DirectParams { uint32_t nb_bsdf_samples = 1, nb_light_samples = 1; };
static Color direct_compute_pixel(const Scene& scene, const DirectParams& dp, uint32_t ix, uint32_t iy, Sampler& sampler, Counters& cnt) {
    float u = (float)ix + sampler.next();
    float v = (float)iy + sampler.next();
    Ray ray = scene.camera.generate({u, v});
    cnt.samples++;
    Color l_i = Color::zero();
    Intersection its;
    cnt.extension_rays++;
    if (!scene.trace(ray, &its)) return scene.environment_luminance(ray.d);
    if (its.wi.z <= 0.0f) return l_i;
    const Mesh& mesh = scene.meshes[its.mesh];
    add_assign(l_i, mesh.emit(its.has_uv, its.uv));
    float w_nb_bsdf = dp.nb_bsdf_samples == 0 ? 0.0f : 1.0f / (float)dp.nb_bsdf_samples;
    float w_nb_light = dp.nb_light_samples == 0 ? 0.0f : 1.0f / (float)dp.nb_light_samples;
    cnt.vertices++;
    for (uint32_t k = 0; k < dp.nb_light_samples; k++) {
        float a = sampler.next();
        float b = sampler.next();
        V2 c = sampler.next2d();
        LightSampling lr = scene.sample_light(its.p, &its.n_s, a, b, c);   // Some(&its.n_s) (direct.rs:64-70)
        V3 d_out_local = its.frame.to_local(lr.d);
        if (!lr.is_valid()) continue;
        cnt.shadow_rays++;
        if (!scene.visible(its.p, lr.p)) continue;
        if (mesh.bsdf.is_smooth()) continue;
        PDF pdf_bsdf = mesh.bsdf.pdf(its.has_uv, its.uv, its.wi, d_out_local, DomSolidAngle);
        float weight_light;
        if (lr.pdf.kind == PDF::SolidAngle) weight_light = mis_weight(lr.pdf.v * w_nb_light, pdf_bsdf.v * w_nb_bsdf);
        else weight_light = 1.0f;   // (Discrete, _): the light is discrete, MIS does not apply
        add_assign(l_i, weight_light * mesh.bsdf.eval(its.has_uv, its.uv, its.wi, d_out_local, DomSolidAngle) * w_nb_light * lr.weight);
    }
    for (uint32_t k = 0; k < dp.nb_bsdf_samples; k++) {
        SampledDirection sd;
        V2 s2 = sampler.next2d();
        if (!mesh.bsdf.sample(its.has_uv, its.uv, its.wi, s2, &sd)) continue;
        V3 d_out_world = its.frame.to_world(sd.d);
        Ray r2 = {its.p, d_out_world, EPSILON, F32_MAX};
        Intersection nx;
        cnt.extension_rays++;
        if (scene.trace(r2, &nx)) {
            const Mesh& nm = scene.meshes[nx.mesh];
            if (nm.is_light && dot(nx.n_g, -r2.d) > 0.0f) {
                float weight_bsdf;
                if (sd.pdf.kind == PDF::SolidAngle) {
                    float light_pdf = scene.direct_pdf(nx.mesh, {r2.o, nx.p, nx.n_g, r2.d}, &its.n_s, nx.primitive_id).value();   // direct.rs:156-164
                    weight_bsdf = mis_weight(sd.pdf.v * w_nb_bsdf, light_pdf * w_nb_light);
                } else weight_bsdf = 1.0f;
                add_assign(l_i, weight_bsdf * sd.weight * nm.emit(nx.has_uv, nx.uv) * w_nb_bsdf);
            }
        } else if (scene.has_env) {
            float weight_bsdf;
            if (sd.pdf.kind == PDF::SolidAngle) weight_bsdf = mis_weight(sd.pdf.v * w_nb_bsdf, scene.direct_pdf_env(r2.d).value() * w_nb_light);
            else weight_bsdf = 1.0f;
            add_assign(l_i, weight_bsdf * sd.weight * scene.environment_luminance(r2.d) * w_nb_bsdf);
        }
    }
    return l_i;
}

}  // namespace orc

// ==========================================================================================
// C API
// ==========================================================================================
using namespace orc;

struct orc_scene { Scene s; };

static BSDFColor conv_color(const rl_color_desc& d, const Scene& s) {
    BSDFColor c;
    c.type = d.type;
    c.c0 = {d.color0[0], d.color0[1], d.color0[2]};
    c.c1 = {d.color1[0], d.color1[1], d.color1[2]};
    c.offset = {d.offset[0], d.offset[1]};
    c.scale = {d.scale[0], d.scale[1]};
    c.line_width = d.line_width;
    if (d.type == RL_TEX_BITMAP && d.bitmap_id >= 0 && (size_t)d.bitmap_id < s.bitmaps.size()) c.img = s.bitmaps[d.bitmap_id].get();
    return c;
}

extern "C" {

orc_scene* orc_scene_create(void) { return new orc_scene(); }
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_scene_set_camera(orc_scene* sc, uint32_t w, uint32_t h, float fov, int fov_axis, const float* to_world, int flip) {
    M4 m;
    for (int c = 0; c < 4; c++) m.c[c] = {to_world[4 * c], to_world[4 * c + 1], to_world[4 * c + 2], to_world[4 * c + 3]};
    return sc->s.camera.init(w, h, fov, fov_axis, m, flip != 0) ? 0 : -1;
}

int orc_scene_add_bitmap(orc_scene* sc, uint32_t w, uint32_t h, const float* rgb) {
    auto b = std::make_unique<Bitmap>();
    b->w = w; b->h = h; b->colors.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) b->colors[i] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    sc->s.bitmaps.push_back(std::move(b));
    return (int)sc->s.bitmaps.size() - 1;
}

int orc_scene_add_mesh(orc_scene* sc, const float* vertices, size_t nv, const uint32_t* indices, size_t ntri,
                       const float* normals, const float* uv, const rl_bsdf_desc* bd, const float* emission) {
    Mesh m;
    m.vertices.resize(nv);
    for (size_t i = 0; i < nv; i++) m.vertices[i] = {vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]};
    m.indices.assign(indices, indices + 3 * ntri);
    if (normals) { m.has_normals = true; m.normals.resize(nv); for (size_t i = 0; i < nv; i++) m.normals[i] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]}; }
    if (uv) { m.has_uv = true; m.uv.resize(nv); for (size_t i = 0; i < nv; i++) m.uv[i] = {uv[2 * i], uv[2 * i + 1]}; }
    BSDF b;
    b.type = bd->type;
    b.diffuse = conv_color(bd->diffuse, sc->s);
    b.specular = conv_color(bd->specular, sc->s);
    b.transmittance = conv_color(bd->transmittance, sc->s);
    b.eta = conv_color(bd->eta, sc->s);
    b.k = conv_color(bd->k, sc->s);
    b.exponent = bd->exponent; b.weight_specular = bd->weight_specular;
    b.distribution = bd->distribution; b.alpha_u = bd->alpha_u; b.alpha_v = bd->alpha_v;
    b.g_eta = bd->glass_eta; b.g_inv_eta = 1.0f / bd->glass_eta;
    m.bsdf = b;
    if (emission) { m.is_light = true; m.emission = {emission[0], emission[1], emission[2]}; }
    if (!m.finish()) return -1;
    sc->s.meshes.push_back(std::move(m));
    return (int)sc->s.meshes.size() - 1;
}

// examples/cli.rs:410-429 (`-x hvs-light`, `-x texture-light`): the emission of a light mesh becomes EmissionType::HSV { scale } or Texture { scale, img }
int orc_scene_set_mesh_emission(orc_scene* sc, int mesh, int type, float scale, int bitmap_id) {
    if (mesh < 0 || (size_t)mesh >= sc->s.meshes.size() || type < 0 || type > 2) return -1;
    Mesh& m = sc->s.meshes[mesh];
    if (!m.is_light) return -1;
    if (type != 0 && !m.has_uv) return -1;                        // `uv.unwrap()` would panic
    if (type == 2 && (bitmap_id < 0 || (size_t)bitmap_id >= sc->s.bitmaps.size())) return -1;
    m.emission_type = type; m.emission_scale = scale;
    m.emission_img = type == 2 ? sc->s.bitmaps[bitmap_id].get() : nullptr;
    return 0;
}

int orc_scene_set_medium(orc_scene* sc, const float* sigma_a, const float* sigma_s, int phase, float g) {
    Volume v;
    v.sigma_a = {sigma_a[0], sigma_a[1], sigma_a[2]};
    v.sigma_s = {sigma_s[0], sigma_s[1], sigma_s[2]};
    v.sigma_t = (v.sigma_a + v.sigma_s) * 1.0f;  // cli.rs:383-385 (density_mult = 1.0)
    v.phase = phase; v.g = g;
    sc->s.volume = v; sc->s.has_volume = true;
    return 0;
}

int orc_scene_add_point_light(orc_scene* sc, const float* position, const float* intensity) {
    EmitterRec e; e.kind = EM_POINT; e.v = {position[0], position[1], position[2]}; e.c = {intensity[0], intensity[1], intensity[2]};
    sc->s.other_emitters.push_back(e);
    return 0;
}
int orc_scene_add_directional_light(orc_scene* sc, const float* direction, const float* intensity) {
    EmitterRec e; e.kind = EM_DIRECTIONAL; e.v = {direction[0], direction[1], direction[2]}; e.c = {intensity[0], intensity[1], intensity[2]};
    sc->s.other_emitters.push_back(e);
    return 0;
}
int orc_scene_set_environment(orc_scene* sc, const float* rgb) {
    sc->s.has_env = true; sc->s.env_color = {rgb[0], rgb[1], rgb[2]};
    return 0;
}

int orc_scene_set_environment_map(orc_scene* sc, uint32_t w, uint32_t h, const float* rgb) {   // EnvironmentLightColor::new_texture
    if (w == 0 || h == 0) return -1;
    sc->s.has_env = true;
    sc->s.env_tex.reset(new EnvTexture());
    sc->s.env_tex->image.w = w; sc->s.env_tex->image.h = h;
    sc->s.env_tex->image.colors.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) sc->s.env_tex->image.colors[i] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    sc->s.env_tex->build();
    return 0;
}

int orc_scene_set_ats(orc_scene* sc, int build_ats) { sc->s.want_ats = build_ats != 0; return 0; }   // Scene::build_emitters(build_ats)
// light tree dump, 16 words per node in node order: aabb min / max, cone axis, phi, cos_theta_o, cos_theta_e, then
// left / right / parent / light as int32 bit patterns (-1 = none); `leaf_of` = node of every (emitter, triangle) in proxy order
int orc_ats_dump(const orc_scene* sc, uint64_t* n_nodes, float* nodes16, uint64_t* n_lights, int32_t* light_emitter, int32_t* light_prim) {
    const LightSamplerATS* a = sc->s.ats.get();
    if (!a) { *n_nodes = 0; *n_lights = 0; return 0; }
    if (nodes16) for (size_t i = 0; i < a->nodes.size(); i++) {
        const LightBVHNode& nd = a->nodes[i];
        float* o = nodes16 + 16 * i;
        o[0] = nd.bounds.aabb.p_min.x; o[1] = nd.bounds.aabb.p_min.y; o[2] = nd.bounds.aabb.p_min.z;
        o[3] = nd.bounds.aabb.p_max.x; o[4] = nd.bounds.aabb.p_max.y; o[5] = nd.bounds.aabb.p_max.z;
        o[6] = nd.bounds.w.x; o[7] = nd.bounds.w.y; o[8] = nd.bounds.w.z;
        o[9] = nd.bounds.phi; o[10] = nd.bounds.cos_theta_o; o[11] = nd.bounds.cos_theta_e;
        int32_t link[4] = {(int32_t)nd.left, (int32_t)nd.right, (int32_t)nd.parent, (int32_t)nd.light};
        std::memcpy(o + 12, link, sizeof(link));
    }
    if (light_emitter) for (size_t i = 0; i < a->lights.size(); i++) { light_emitter[i] = (int32_t)a->lights[i].emitter_id; light_prim[i] = (int32_t)a->lights[i].primitive_idx; }
    *n_nodes = a->nodes.size(); *n_lights = a->lights.size();
    return 0;
}

// light-tree probes: kind 0 sample(r = in[0], p = in[1..3], n = in[4..6] if in[7] != 0) -> out = {emitter, prim, pdf_sel};
// kind 1 pdf(emitter = in[0], prim = in[1], p = in[2..4], n = in[5..7] if in[8] != 0) -> out[0]
int orc_ats_probe(const orc_scene* sc, int kind, const float* in, float* out) {
    const LightSamplerATS* a = sc->s.ats.get();
    if (!a) return -1;
    if (kind == 0) {
        V3 p{in[1], in[2], in[3]}, n{in[4], in[5], in[6]};
        const V3* np_ = in[7] != 0.0f ? &n : nullptr;
        float pdf_sel;
        const LightProxy& lp = a->sample(in[0], [&](const LightBounds& b) { return b.importance_point(p, np_); }, &pdf_sel);
        out[0] = (float)lp.emitter_id; out[1] = (float)lp.primitive_idx; out[2] = pdf_sel;
    } else {
        V3 p{in[2], in[3], in[4]}, n{in[5], in[6], in[7]};
        const V3* np_ = in[8] != 0.0f ? &n : nullptr;
        out[0] = a->pdf((size_t)in[0], (size_t)in[1], [&](const LightBounds& b) { return b.importance_point(p, np_); });
    }
    return 0;
}

int orc_scene_build(orc_scene* sc) {
    sc->s.build_emitters();
    sc->s.build_bvh();
    return 0;
}

// ---- RNG known-answer helpers
void orc_rng_seed(uint64_t seed, int variant, uint64_t* state_out) { Rng r = Rng::seed_from_u64(seed, variant); std::memcpy(state_out, r.s, 32); }
uint64_t orc_rng_next_u64(uint64_t* state) { Rng r; std::memcpy(r.s, state, 32); uint64_t v = r.next_u64(); std::memcpy(state, r.s, 32); return v; }
float orc_rng_next_f32(uint64_t* state) { Rng r; std::memcpy(r.s, state, 32); float v = r.next_f32(); std::memcpy(state, r.s, 32); return v; }

size_t orc_block_count(uint32_t w, uint32_t h) { return (size_t)((w + 15) / 16) * ((h + 15) / 16); }
// generate_img_blocks seeds (integrators/mod.rs:357-371): x-major creation order
void orc_generate_block_seeds(uint64_t* master_state, uint32_t w, uint32_t h, uint64_t* seeds) {
    Rng r; std::memcpy(r.s, master_state, 32);
    size_t k = 0;
    for (uint32_t ix = 0; ix < w; ix += 16) for (uint32_t iy = 0; iy < h; iy += 16) seeds[k++] = r.next_u64();
    std::memcpy(master_state, r.s, 32);
}

// ---- math KAT helpers
float orc_sinf(float x) { return detmath::sinf_det(x); }
float orc_cosf(float x) { return detmath::cosf_det(x); }
float orc_expf(float x) { return detmath::expf_det(x); }
float orc_logf(float x) { return detmath::logf_det(x); }
float orc_powf(float x, float y) { return detmath::powf_det(x, y); }
float orc_acosf(float x) { return detmath::acosf_det(x); }
float orc_atan2f(float y, float x) { return detmath::atan2f_det(y, x); }
void orc_math_batch(int fn, size_t n, const float* a, const float* b, float* out) {
    for (size_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = detmath::sinf_det(a[i]); break;
            case 1: out[i] = detmath::cosf_det(a[i]); break;
            case 2: out[i] = detmath::expf_det(a[i]); break;
            case 3: out[i] = detmath::logf_det(a[i]); break;
            case 4: out[i] = detmath::powf_det(a[i], b[i]); break;
            case 5: out[i] = detmath::acosf_det(a[i]); break;
            case 6: out[i] = detmath::atan2f_det(a[i], b[i]); break;
            case 7: out[i] = detmath::asinf_det(a[i]); break;
            default: out[i] = 0.0f;
        }
    }
}

// ---- textured-environment probe: kind 0 sample_direction(u) -> out[0..2] d, [3..5] value, [6] pdf;
//      kind 1 eval(d) -> out[0..2]; kind 2 pdf(d) -> out[0]; kind 3 -> marginal.func_int
int orc_env_probe(const orc_scene* sc, int kind, const float* in, float* out) {
    const EnvTexture* t = sc->s.env_tex.get();
    if (!t) return -1;
    if (kind == 0) { V3 d; Color v; float pdf; t->sample_direction({in[0], in[1]}, &d, &v, &pdf); out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = v.r; out[4] = v.g; out[5] = v.b; out[6] = pdf; }
    else if (kind == 1) { Color v = t->eval({in[0], in[1], in[2]}); out[0] = v.r; out[1] = v.g; out[2] = v.b; }
    else if (kind == 2) out[0] = t->pdf({in[0], in[1], in[2]});
    else out[0] = t->cdf.marginal.func_int;
    return 0;
}

// ---- camera / sampling probes (for unit tests)
void orc_camera_generate(const orc_scene* sc, float px, float py, float* o, float* d) {
    Ray r = sc->s.camera.generate({px, py});
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
void orc_scene_info(const orc_scene* sc, uint64_t* n_nodes, uint64_t* n_prims, uint64_t* n_emitters, float* bsphere) {
    *n_nodes = sc->s.nodes.size(); *n_prims = sc->s.primitives.size(); *n_emitters = sc->s.emitters.size();
    bsphere[0] = sc->s.bsphere.center.x; bsphere[1] = sc->s.bsphere.center.y; bsphere[2] = sc->s.bsphere.center.z; bsphere[3] = sc->s.bsphere.radius;
}
// BVH dump: per node {min xyz, max xyz} + info + count ; per primitive {mesh, tri}
void orc_bvh_dump(const orc_scene* sc, float* boxes, uint64_t* info, uint64_t* count, int32_t* prim_mesh, int32_t* prim_tri) {
    for (size_t i = 0; i < sc->s.nodes.size(); i++) {
        const auto& n = sc->s.nodes[i];
        boxes[6 * i] = n.aabb.p_min.x; boxes[6 * i + 1] = n.aabb.p_min.y; boxes[6 * i + 2] = n.aabb.p_min.z;
        boxes[6 * i + 3] = n.aabb.p_max.x; boxes[6 * i + 4] = n.aabb.p_max.y; boxes[6 * i + 5] = n.aabb.p_max.z;
        info[i] = n.info; count[i] = n.count;
    }
    for (size_t i = 0; i < sc->s.primitives.size(); i++) { prim_mesh[i] = sc->s.primitives[i].id_mesh; prim_tri[i] = (int32_t)sc->s.primitives[i].id_tri; }
}

// ---- batched Acceleration::{trace,visible}; brute = NaiveAcceleration (accel.rs:14-77)
int orc_trace_batch(const orc_scene* sc, size_t n, const float* o, const float* d, int brute,
                    float* t_out, float* u_out, float* v_out, int32_t* mesh_out, int32_t* tri_out) {
    const Scene& s = sc->s;
    for (size_t i = 0; i < n; i++) {
        Ray ray = Ray::make({o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]});
        IntersectionUV its = {F32_MAX, {0, 0, 0}, {0, 0, 0}, 0, 0};
        Scene::TriRef res{-1, 0};
        bool hit;
        if (brute) {
            for (size_t m = 0; m < s.meshes.size(); m++)
                for (size_t k = 0; k < s.meshes[m].n_tris(); k++)
                    if (s.meshes[m].intersection_tri(k, ray.o, ray.d, &its)) res = {(int)m, k};
            hit = its.t != F32_MAX;
        } else hit = s.trace_uv(ray, &its, &res);
        if (hit) { t_out[i] = its.t; u_out[i] = its.u; v_out[i] = its.v; mesh_out[i] = res.id_mesh; tri_out[i] = (int32_t)res.id_tri; }
        else { t_out[i] = F32_MAX; u_out[i] = 0; v_out[i] = 0; mesh_out[i] = -1; tri_out[i] = -1; }
    }
    return 0;
}
int orc_visible_batch(const orc_scene* sc, size_t n, const float* p0, const float* p1, uint8_t* out) {
    for (size_t i = 0; i < n; i++) out[i] = sc->s.visible({p0[3 * i], p0[3 * i + 1], p0[3 * i + 2]}, {p1[3 * i], p1[3 * i + 1], p1[3 * i + 2]}) ? 1 : 0;
    return 0;
}

// full surface interaction record for one ray (tests of fill_intersection / two-sided rule)
int orc_trace_full(const orc_scene* sc, const float* o, const float* d, float* out /*[t, p3, ng3, ns3, uv2, wi3, mesh, tri]*/) {
    Ray ray = Ray::make({o[0], o[1], o[2]}, {d[0], d[1], d[2]});
    Intersection its;
    if (!sc->s.trace(ray, &its)) return 0;
    float v[] = {its.dist, its.p.x, its.p.y, its.p.z, its.n_g.x, its.n_g.y, its.n_g.z, its.n_s.x, its.n_s.y, its.n_s.z,
                 its.uv.x, its.uv.y, its.wi.x, its.wi.y, its.wi.z, (float)its.mesh, (float)its.primitive_id};
    std::memcpy(out, v, sizeof(v));
    return 1;
}

// BSDF probes: op 0 = sample (out: ok, weight3, d3, pdf, pdf_kind), 1 = eval (out: rgb), 2 = pdf (out: value, kind)
int orc_bsdf_probe(const orc_scene* sc, int mesh, int op, const float* wi, const float* wo_or_sample, float* out) {
    const BSDF& b = sc->s.meshes[mesh].bsdf;
    V3 w = {wi[0], wi[1], wi[2]};
    if (op == 0) {
        SampledDirection sd;
        bool ok = b.sample(false, {0, 0}, w, {wo_or_sample[0], wo_or_sample[1]}, &sd);
        out[0] = ok ? 1.0f : 0.0f;
        if (ok) { out[1] = sd.weight.r; out[2] = sd.weight.g; out[3] = sd.weight.b; out[4] = sd.d.x; out[5] = sd.d.y; out[6] = sd.d.z; out[7] = sd.pdf.v; out[8] = (float)sd.pdf.kind; }
        return 0;
    }
    V3 o = {wo_or_sample[0], wo_or_sample[1], wo_or_sample[2]};
    if (op == 1) { Color c = b.eval(false, {0, 0}, w, o, DomSolidAngle); out[0] = c.r; out[1] = c.g; out[2] = c.b; return 0; }
    PDF p = b.pdf(false, {0, 0}, w, o, DomSolidAngle); out[0] = p.v; out[1] = (float)p.kind;
    return 0;
}
// EmitterSampler::sample_light probe: out = [pdf, p3, n3, d3, weight3, mesh id (or -kind), pdf kind]
int orc_sample_light(const orc_scene* sc, const float* p, float r_sel, float r, float ux, float uy, float* out) {
    LightSampling l = sc->s.sample_light({p[0], p[1], p[2]}, nullptr, r_sel, r, {ux, uy});
    const EmitterRec& em = sc->s.emitters[l.emitter];
    float v[] = {l.pdf.v, l.p.x, l.p.y, l.p.z, l.n.x, l.n.y, l.n.z, l.d.x, l.d.y, l.d.z, l.weight.r, l.weight.g, l.weight.b,
                 (float)(em.kind == EM_MESH ? em.mesh : -em.kind), (float)l.pdf.kind};
    std::memcpy(out, v, sizeof(v));
    return 0;
}

// single camera sample with an explicit sampler state (draw-level tests). Returns draws consumed.
uint64_t orc_compute_pixel(const orc_scene* sc, const orc_path_params* pp, uint32_t ix, uint32_t iy, uint64_t* rng_state, float* rgb,
                           uint64_t* n_vertices, uint64_t* n_shadow) {
    PathParams p;
    p.has_min = pp->has_min_depth; p.min_depth = pp->min_depth; p.has_max = pp->has_max_depth; p.max_depth = pp->max_depth;
    p.has_rr = pp->has_rr_depth; p.rr_depth = pp->rr_depth; p.strategy = pp->strategy; p.single_scattering = pp->single_scattering; p.eval_order = pp->eval_order;
    PathTracer pt(sc->s, p);
    Sampler s; std::memcpy(s.rnd.s, rng_state, 32); s.variant = pp->seed_variant;
    Color c = pt.compute_pixel(ix, iy, s);
    std::memcpy(rng_state, s.rnd.s, 32);
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
    if (n_vertices) *n_vertices = pt.cnt.vertices;
    if (n_shadow) *n_shadow = pt.cnt.shadow_rays;
    return s.draws;
}

}  // extern "C" (reopened below)

// compute_mc (integrators/mod.rs:403-450): tiles, per-block sampler, accumulate, scale, merge.
// block_seeds: one u64 per block in creation order.  Renders blocks with b % shard_count == shard_index.
// `make_worker()` returns a per-thread object with `Color pixel(ix, iy, Sampler&)` and a `Counters cnt`.
template <class MakeWorker>
static int render_tiles(const Scene& scene, uint32_t spp, int stream_mode, int seed_variant, uint32_t shard_index, uint32_t shard_count_,
                        const uint64_t* block_seeds, size_t n_blocks, float* out_rgb, int n_threads, orc_stats* stats, MakeWorker make_worker) {
    if (!scene.bvh_built || !scene.emitters_built) return -4;
    uint32_t W = scene.camera.w, H = scene.camera.h;
    size_t nby = (H + 15) / 16, nbx = (W + 15) / 16;
    if (n_blocks != nbx * nby) return -1;
    if (spp == 0) return -1;
    std::memset(out_rgb, 0, sizeof(float) * 3 * (size_t)W * H);
    uint32_t shard_count = shard_count_ ? shard_count_ : 1;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    std::atomic<size_t> next_block{0};
    std::vector<Counters> counters(n_threads);
    auto worker = [&](int tid) {
        auto wk = make_worker();
        std::vector<Color> block(256);
        for (;;) {
            size_t b = next_block.fetch_add(1);
            if (b >= n_blocks) break;
            if (b % shard_count != shard_index) continue;
            uint32_t bx = (uint32_t)(b / nby) * 16, by = (uint32_t)(b % nby) * 16;
            uint32_t bw = std::min(16u, W - bx), bh = std::min(16u, H - by);
            Sampler block_sampler; block_sampler.rnd = Rng::seed_from_u64(block_seeds[b], seed_variant); block_sampler.variant = seed_variant;
            std::fill(block.begin(), block.end(), Color::zero());
            for (uint32_t iy = 0; iy < bh; iy++)
                for (uint32_t ix = 0; ix < bw; ix++) {
                    Sampler pixel_sampler;
                    if (stream_mode == RL_STREAM_PER_SAMPLE) pixel_sampler = block_sampler.clone_box();
                    for (uint32_t s = 0; s < spp; s++) {
                        Color c;
                        if (stream_mode == RL_STREAM_PER_SAMPLE) {
                            Sampler sample_sampler = pixel_sampler.clone_box();
                            c = wk.pixel(ix + bx, iy + by, sample_sampler);
                            wk.cnt.draws += sample_sampler.draws;
                        } else {
                            uint64_t d0 = block_sampler.draws;
                            c = wk.pixel(ix + bx, iy + by, block_sampler);
                            wk.cnt.draws += block_sampler.draws - d0;
                        }
                        add_assign(block[iy * 16 + ix], c);
                    }
                }
            float inv = 1.0f / (float)spp;
            for (uint32_t iy = 0; iy < bh; iy++)
                for (uint32_t ix = 0; ix < bw; ix++) {
                    Color c = block[iy * 16 + ix];
                    c.scale(inv);
                    float* o = out_rgb + 3 * ((size_t)(iy + by) * W + (ix + bx));
                    o[0] += c.r; o[1] += c.g; o[2] += c.b;  // accumulate_bitmap into a zeroed image
                }
        }
        counters[tid] = wk.cnt;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (auto& c : counters) { stats->camera_samples += c.samples; stats->vertices += c.vertices; stats->extension_rays += c.extension_rays; stats->shadow_rays += c.shadow_rays; stats->rng_draws += c.draws; }
        stats->threads = (uint32_t)n_threads;
    }
    return 0;
}

extern "C" {

int orc_render_path(const orc_scene* sc, const orc_path_params* pp, const uint64_t* block_seeds, size_t n_blocks,
                    float* out_rgb, int n_threads, orc_stats* stats) {
    const Scene& scene = sc->s;
    PathParams p;
    p.has_min = pp->has_min_depth; p.min_depth = pp->min_depth; p.has_max = pp->has_max_depth; p.max_depth = pp->max_depth;
    p.has_rr = pp->has_rr_depth; p.rr_depth = pp->rr_depth; p.strategy = pp->strategy; p.single_scattering = pp->single_scattering; p.eval_order = pp->eval_order;
    if ((p.strategy == RL_STRATEGY_ALL || p.strategy == RL_STRATEGY_EMITTER) && scene.emitters.empty()) return -8;
    struct W { PathTracer pt; Counters& cnt; W(const Scene& s, const PathParams& p) : pt(s, p), cnt(pt.cnt) {} Color pixel(uint32_t x, uint32_t y, Sampler& sm) { return pt.compute_pixel(x, y, sm); } };
    return render_tiles(scene, pp->spp, pp->stream_mode, pp->seed_variant, pp->shard_index, pp->shard_count, block_seeds, n_blocks, out_rgb, n_threads, stats,
                        [&]() { return W(scene, p); });
}

// ao / direct.  kind: 0 = ao, 1 = direct
int orc_render_mc(const orc_scene* sc, int kind, const orc_mc_params* mp, const uint64_t* block_seeds, size_t n_blocks,
                  float* out_rgb, int n_threads, orc_stats* stats) {
    const Scene& scene = sc->s;
    if (kind == 0) {
        AOParams ap; ap.has_max_distance = mp->has_max_distance != 0; ap.max_distance = mp->max_distance; ap.normal_correction = mp->normal_correction != 0;
        struct W { const Scene& s; AOParams ap; Counters cnt; Color pixel(uint32_t x, uint32_t y, Sampler& sm) { return ao_compute_pixel(s, ap, x, y, sm, cnt); } };
        return render_tiles(scene, mp->spp, mp->stream_mode, mp->seed_variant, mp->shard_index, mp->shard_count, block_seeds, n_blocks, out_rgb, n_threads, stats,
                            [&]() { return W{scene, ap, Counters()}; });
    }
    if (kind == 1) {
        if (mp->nb_light_samples > 0 && scene.emitters.empty()) return -8;
        DirectParams dp; dp.nb_bsdf_samples = mp->nb_bsdf_samples; dp.nb_light_samples = mp->nb_light_samples;
        struct W { const Scene& s; DirectParams dp; Counters cnt; Color pixel(uint32_t x, uint32_t y, Sampler& sm) { return direct_compute_pixel(s, dp, x, y, sm, cnt); } };
        return render_tiles(scene, mp->spp, mp->stream_mode, mp->seed_variant, mp->shard_index, mp->shard_count, block_seeds, n_blocks, out_rgb, n_threads, stats,
                            [&]() { return W{scene, dp, Counters()}; });
    }
    return -1;
}

}  // extern "C"
