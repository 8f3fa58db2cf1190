// trace.hip.h — BVH2 traversal for gfx950: `Acceleration::{trace,visible}` of rustlight's BVHAccel
// (src/accel.rs:243-343) restated as an iterative, stack-based, wave64-friendly loop.
//
//  * ordered traversal: both child boxes come with the parent's 64-byte record; the nearer child
//    (by slab entry distance, ties keep the left child first) is entered first and the farther
//    one is pushed with its entry distance, which is re-checked against the current closest hit
//    when popped — exactly the order and the pruning rule of the reference's recursion;
//  * the per-lane stack lives in LDS (layout [level][lane], conflict-free);
//  * small scenes (nodes + triangles fit the LDS budget) are staged into LDS once per workgroup,
//    so the inner loop never leaves the CU; otherwise records stream from L2/HBM as four
//    16-byte loads per lane;
//  * triangle test = Mesh::intersection_tri (src/geometry.rs:358-410) with the ray-independent
//    terms (e1, e2, n_geo, det) precomputed in the TriRecord.
#pragma once
#include "../device_types.h"
#include "devmath.hip.h"
#include <type_traits>

namespace rl {

struct Hit { float t, u, v; int prim; int steps = 0, tris = 0, fetches = 0; };   // steps / tris: dev-only traversal statistics (dead code unless read)

// Layout of a scene staged in LDS.  A 64-byte record stride puts the same field of every node on two LDS banks (bank = dword
// address mod 32) and the 16-byte quarters of every triangle on four bank groups, so lanes that read different records collide
// (SQ_LDS_BANK_CONFLICT was 25 % of the LDS cycles of k_path_fused on the Cornell box): nodes are staged 17 dwords apart, triangles
// 20 dwords (ds_read_b128 needs 16-byte alignment), which spreads distinct records over distinct banks.  Child references of
// inner nodes are rewritten to byte offsets (index x stride x 4) while staging, so a node's address is one add per plane pair.
#ifndef RL_LDS_NODE_STRIDE
#define RL_LDS_NODE_STRIDE 17
#endif
#ifndef RL_LDS_TRI_STRIDE4
#define RL_LDS_TRI_STRIDE4 5
#endif
static constexpr int kLdsNodeStride = RL_LDS_NODE_STRIDE;      // dwords
static constexpr int kLdsNode2Stride = 36;                     // dwords between two-level records staged in LDS (32 + 4: ds_read_b128 rows stay 16-byte aligned, consecutive records start 4 banks apart)
static constexpr int kLdsTriStride4 = RL_LDS_TRI_STRIDE4;      // float4s
RL_DEV __host__ unsigned lds_nodes_float4s(unsigned n_nodes) { return (n_nodes * (unsigned)kLdsNodeStride + 3u) / 4u; }
RL_DEV __host__ unsigned lds_scene_float4s(unsigned n_nodes, unsigned n_prims) { return lds_nodes_float4s(n_nodes) + n_prims * (unsigned)kLdsTriStride4; }
// the same scene with the nodes as two-level records (k_path_fused on LDS-staged scenes: traverse2)
RL_DEV __host__ unsigned lds_nodes2_float4s(unsigned n_nodes) { return n_nodes * (unsigned)(kLdsNode2Stride / 4); }
RL_DEV __host__ unsigned lds_scene2_float4s(unsigned n_nodes, unsigned n_prims) { return lds_nodes2_float4s(n_nodes) + n_prims * (unsigned)kLdsTriStride4; }

// AABB::intersect (src/structure.rs:849-869) with 1/d hoisted out of the loop, restated without its per-axis
// early exit and compare/select chains — same value, same verdict, a third fewer VALU instructions:
//  * `t_min = if a0 > t_min {a0} else {t_min}` never makes t_min a NaN (a NaN a0 = 0 * inf loses the compare), so it
//    is IEEE maxNum(a0, t_min) = v_max_f32, and likewise t_max = minNum(a1, t_max); maxNum/minNum skip NaNs in any
//    association, so the three axes fold into v_max3_f32 / v_min3_f32.  (t_min >= tnear > 0: no signed-zero case.)
//  * t_min only grows and t_max only shrinks, so the reference's exit `t_max <= t_min` after axis k implies the
//    same relation after axis 3: one final test decides, and on success t_min is the returned entry distance.
// The near/far plane choice stays a select on the sign of 1/d (min/max of t0, t1 would resurrect a NaN plane).
#ifdef RL_SLAB_REFERENCE_FORM
RL_DEV bool slab(V3 lo, V3 hi, V3 o, V3 inv_d, float tnear, float tfar, float* t_entry) {
    float t_min = tnear, t_max = tfar;
    float t0 = (lo.x - o.x) * inv_d.x, t1 = (hi.x - o.x) * inv_d.x;
    float a0 = inv_d.x < 0.0f ? t1 : t0, a1 = inv_d.x < 0.0f ? t0 : t1;
    t_min = a0 > t_min ? a0 : t_min;
    t_max = a1 < t_max ? a1 : t_max;
    bool ok = !(t_max <= t_min);
    t0 = (lo.y - o.y) * inv_d.y; t1 = (hi.y - o.y) * inv_d.y;
    a0 = inv_d.y < 0.0f ? t1 : t0; a1 = inv_d.y < 0.0f ? t0 : t1;
    t_min = a0 > t_min ? a0 : t_min;
    t_max = a1 < t_max ? a1 : t_max;
    ok = ok && !(t_max <= t_min);
    t0 = (lo.z - o.z) * inv_d.z; t1 = (hi.z - o.z) * inv_d.z;
    a0 = inv_d.z < 0.0f ? t1 : t0; a1 = inv_d.z < 0.0f ? t0 : t1;
    t_min = a0 > t_min ? a0 : t_min;
    t_max = a1 < t_max ? a1 : t_max;
    ok = ok && !(t_max <= t_min);
    *t_entry = t_min;
    return ok;
}
#else
RL_DEV bool slab(V3 lo, V3 hi, V3 o, V3 inv_d, float tnear, float tfar, float* t_entry) {
    const float x0 = (lo.x - o.x) * inv_d.x, x1 = (hi.x - o.x) * inv_d.x;
    const float y0 = (lo.y - o.y) * inv_d.y, y1 = (hi.y - o.y) * inv_d.y;
    const float z0 = (lo.z - o.z) * inv_d.z, z1 = (hi.z - o.z) * inv_d.z;
    const bool sx = inv_d.x < 0.0f, sy = inv_d.y < 0.0f, sz = inv_d.z < 0.0f;
    const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(sx ? x1 : x0, sy ? y1 : y0), sz ? z1 : z0), tnear);
    const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fminf(sx ? x0 : x1, sy ? y0 : y1), sz ? z0 : z1), tfar);
    *t_entry = t_min;
    return !(t_max <= t_min);
}
#endif

// Mesh::intersection_tri; returns true and updates `hit` (t, prim) if the triangle is the new closest hit.
// The reference evaluates u, v (two sqrt + two divides) before it looks at `t < its.t && t > 1e-5`
// (geometry.rs:391-398).  All conditions are pure and must hold together, so testing the cheap
// distance window first accepts exactly the same set of hits with the same (t, u, v) bits.
//
// The barycentric verdict `!(u < 0 || v < 0 || u > 1 || v > 1) && u + v <= 1` with v = |u0| / det, u = |w0| / det (both
// correctly rounded sqrt and divide, ~90 issue slots a wave pays whenever one of its lanes gets this far) is decided from
// s = sqrt~(|u0|^2) + sqrt~(|w0|^2) (1-ulp v_sqrt_f32) against det: the f32 value of u + v is within 4e-7 relative of
// s / det, so s <= det (1 - 1e-5) accepts and s >= det (1 + 1e-5) rejects exactly as the reference does; only the band in
// between (and denormal-range or non-finite inputs) takes the reference's own arithmetic.  u, v themselves are needed for
// the final closest hit only and are recomputed there from (t, prim) by `tri_uv` with the reference's operations (same
// inputs, same bits).
RL_DEV void tri_uv(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o, V3 d, float t, float* u, float* v) {
    V3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    V3 pv = (o + t * d) - v0;
    *v = div_rn(length(cross(e1, pv)), q3.x);
    *u = div_rn(length(cross(pv, e2)), q3.x);
}
RL_DEV bool tri_test(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o, V3 d, Hit& hit, int prim) {
    const V3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    const V3 n = mk3(q0.w, q1.w, q2.w);
    const float det = q3.x;
    const float denom = dot(d, n);
    const float t = div_rn(-dot(o - v0, n), denom);
    // The reference rejects on `denom == 0`, then on `t < 0`, and after the barycentrics on `!(t < its.t && t > 1e-5)`.  All of them are pure
    // comparisons, and the last one implies the first two (denom == 0 makes t +-inf or NaN, which fails it; t > 1e-5 excludes t < 0), so one
    // compare pair decides — evaluated without short-circuit: a wave runs every level of an early-out ladder anyway as long as one of its 64
    // lanes gets through, and each level costs half a dozen scalar mask instructions (the ladder was ~35 of the ~130 instructions of a test).
    const bool window = (t < hit.t) & (t > 0.00001f);
    bool accept = false;
    if (window) {
        const V3 pv = (o + t * d) - v0;
        const V3 u0 = cross(e1, pv);
        const V3 w0 = cross(pv, e2);
        const bool facing = !(dot(u0, n) < 0.0f) & !(dot(w0, n) < 0.0f);
        const float uu = dot(u0, u0), ww = dot(w0, w0);
#if defined(RL_FAST_MATH)
        accept = facing & (det > 0.0f) & (__builtin_amdgcn_sqrtf(uu) + __builtin_amdgcn_sqrtf(ww) <= det);     // tolerance build: the 1-ulp estimate decides everywhere (det > 0: a degenerate triangle is rejected as the exact build's NaN barycentrics reject it)
#else
#if defined(RL_TRI_REFERENCE_FORM)
        const bool in_range = false; const float s = 0.0f;
#else
        const float s = __builtin_amdgcn_sqrtf(uu) + __builtin_amdgcn_sqrtf(ww);
        const bool in_range = __builtin_fminf(__builtin_fminf(uu, ww), det) >= 0x1p-100f;   // false for NaNs too
#endif
        const bool sure_in = in_range & (s <= det * 0.99999f), sure_out = in_range & (s >= det * 1.00001f);
        accept = facing & sure_in;
        if (facing & !sure_in & !sure_out) {      // inside the band (or denormal-range / non-finite inputs): the reference's own arithmetic
            const float v = div_rn(sqrt_rn(uu), det);
            const float u = div_rn(sqrt_rn(ww), det);
            accept = !(u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) && (u + v <= 1.0f);
        }
#endif
    }
    hit.t = accept ? t : hit.t;
    hit.prim = accept ? prim : hit.prim;
    return accept;
}

// Scene records either in LDS (staged) or in global memory.
struct SceneRecs {
    const float4* nodes;   // 4 float4 per BvhNode
    const float4* tris;    // 4 float4 per TriRecord
};

// Traverse.  ANY_HIT: return as soon as one triangle is accepted (Acceleration::visible only asks
// whether `intersect` found something; its.t starts at the segment length, accel.rs:316-343).
// `stack` points at this lane's column of the LDS stack, entries are `stride` ints apart;
// two ints per level: child code and the bits of its entry distance.
//
// Control flow: per-lane visit order and pruning are those of the reference's recursion; only the interleaving between lanes
// changes — "while-while" phases on LDS-staged scenes, wave-voted node / leaf trips on streaming scenes (see traverse).
// Per-lane traversal stack: the first `lds_levels` entries live in LDS (layout [level][lane] of 8-byte pairs,
// conflict-free, one ds_read_b64 / ds_write_b64 per pop / push), deeper levels spill to a global overflow buffer
// with the same coalesced layout.  LDS_ONLY (BVH depth <= the LDS levels, always the case for LDS-staged scenes)
// compiles the overflow path out, which keeps every stack access a plain ds_* instruction instead of a flat one.  Keeping only ~12 levels
// in LDS (96 B/lane) lets 8 waves/SIMD stay resident on scenes whose BVH is 20-40 levels deep.
struct NodeFetchLds;
struct NodeFetchGlobal;
#ifndef RL_TWO_LEVEL
// 1: scenes that stream their BVH traverse the two-level records (traverse2) in the exact build.  OFF: measured slower — 508 k triangles, 1080p x 128 spp: node trips per
// ray 19.1 -> 10.5, k_path_fused 305 -> 374 ms, same CRC (profiles/NEGATIVES.md, round 5): the kernel pays per ISSUED instruction (64-lane instructions at 22 % live lanes),
// not per dependent round trip, and a two-level trip issues twice the loads and slab arithmetic of a one-level trip.  The construction stays tested through
// TravStack2 / rl_debug_trace_batch_two_level.
#define RL_TWO_LEVEL 0
#endif
template <bool LDS_ONLY>
struct TravStackT {
    using NodeFetch = typename std::conditional<LDS_ONLY, NodeFetchLds, NodeFetchGlobal>::type;   // LDS-only stacks go with LDS-staged scenes
    static constexpr bool kLdsOnly = LDS_ONLY;
#ifdef RL_FAST_MATH
    static constexpr bool kBvh4 = !LDS_ONLY;       // tolerance build, streaming scenes: quantised BVH4 nodes (traverse4)
#else
    static constexpr bool kBvh4 = false;
#endif
#if RL_TWO_LEVEL && !defined(RL_FAST_MATH)
    static constexpr bool kTwoLevel = !LDS_ONLY;   // exact build, streaming scenes: two-level records (device_types.h: BvhNode2; traverse2)
#else
    static constexpr bool kTwoLevel = false;
#endif
    static constexpr int kTriStride4 = LDS_ONLY ? kLdsTriStride4 : 4;                              // float4s between triangle records
    static constexpr int kNodeRefScale = LDS_ONLY ? 4 * kLdsNodeStride : 1;                        // inner-node reference = index x this (LDS: byte offset)
    static constexpr int lds_stride = 256;       // every traversal kernel runs 256-lane workgroups
    // The two halves of the stack are addressed through pointers of their own address space.  With generic pointers the compiler merged the LDS and the overflow
    // access of a pop into flat loads (and, given the chance, of a push into a flat store): every pop of a streaming kernel went through the vector-memory path,
    // two instructions each.  Typed: ds_read_b64 / one global_load_dwordx2, and the overflow store only when the entry counts — 508 k triangles, 1080p x 128 spp:
    // vector-memory reads 7.94 G -> 6.73 G, writes 1.00 G -> 0.56 G wave-instructions per render (what is left are scratch spills); the time did not move (305 ms):
    // this kernel is not bound by the count of vector-memory instructions after all (profiles/NEGATIVES.md, round 4).  The stack itself rarely passes six pending
    // entries: a ring that keeps the newest six in LDS changed neither count.
    typedef int i2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) i2v LdsPair;
    typedef __attribute__((address_space(1))) i2v GlobPair;
    // (the tolerance build's BVH4 loop keeps generic pointers and the [2 x level][thread] dword layout of the overflow levels: see push)
    using LdsPtr = typename std::conditional<kBvh4, int2*, LdsPair*>::type;
    using GlobPtr = typename std::conditional<kBvh4, int*, GlobPair*>::type;
    LdsPtr lds; int lds_levels;     // lds already offset to this lane; one (code, distance bits) pair per level
    GlobPtr glob; size_t glob_stride;            // overflow levels, [level][thread] pairs; glob already offset to this lane
    // `counts`: the entry will be popped (the BVH2 loops store the far child on every node trip and only advance sp when both children were hit: a free ds_write, not a free global store)
    RL_DEV void push(int sp, int code, float dist, bool counts = true) const { push_to(lds, glob, sp, code, dist, counts); }
    RL_DEV void get(int sp, int* code, float* dist) const { get_from(lds, glob, sp, code, dist); }
    // typed halves
    RL_DEV void push_to(LdsPair* l, GlobPair* g, int sp, int code, float dist, bool counts) const {
        i2v e; e.x = code; e.y = __float_as_int(dist);
        if (LDS_ONLY || sp < lds_levels) l[sp * lds_stride] = e;
        else if (counts) g[(size_t)(sp - lds_levels) * glob_stride] = e;
    }
    RL_DEV void get_from(const LdsPair* l, const GlobPair* g, int sp, int* code, float* dist) const {
        i2v e;
        if (LDS_ONLY || sp < lds_levels) e = l[sp * lds_stride];
        else e = g[(size_t)(sp - lds_levels) * glob_stride];
        *code = e.x; *dist = __int_as_float(e.y);
    }
    // generic halves: the tolerance build's BVH4 loop pushes up to three entries per node, and there the branch-free form the compiler makes of generic pointers (pops as
    // flat loads) is the faster one — typed, 508 k triangles 259.7 -> 268.7 ms, 4.01 M triangles 319.6 -> 333.0 ms on one box (profiles/NEGATIVES.md, round 4)
    RL_DEV void push_to(int2* l, int* g, int sp, int code, float dist, bool) const {
        if (sp < lds_levels) l[sp * lds_stride] = make_int2(code, __float_as_int(dist));
        else { size_t k = (size_t)(2 * (sp - lds_levels)); g[k * glob_stride] = code; g[(k + 1) * glob_stride] = __float_as_int(dist); }
    }
    RL_DEV void get_from(const int2* l, const int* g, int sp, int* code, float* dist) const {
        if (sp < lds_levels) { const int2 e = l[sp * lds_stride]; *code = e.x; *dist = __int_as_float(e.y); }
        else { size_t k = (size_t)(2 * (sp - lds_levels)); *code = g[k * glob_stride]; *dist = __int_as_float(g[(k + 1) * glob_stride]); }
    }
};
using TravStack = TravStackT<false>;
#ifndef RL_LDS_TWO_LEVEL
// 1: k_path_fused on LDS-staged scenes traverses two-level records (one LDS round trip decides two levels; the near / far rows come pre-swizzled as six ds_read_b128,
// no selects).  OFF: measured slower on the same box — cbox 1080p x 128 spp 50.7 -> 59.0 ms, square frame 51.6 -> 59.8, cbox + medium (32 spp) 103.1 -> 115.4, same CRCs
// (profiles/NEGATIVES.md round 5): half the children of a 19-node tree are leaves, so half of the second-level slabs are wasted, and the kernel is bound by VALU issue.
#define RL_LDS_TWO_LEVEL 0
#endif
struct TravStackLds2 : TravStackT<true> {        // LDS-staged scene, two-level records (stage_scene_lds2)
    static constexpr bool kTwoLevel = true;
    RL_DEV explicit TravStackLds2(const TravStackT<true>& s) : TravStackT<true>(s) {}
};
struct TravStack2 : TravStackT<false> {          // the same stack, traversed through the two-level records whatever RL_TWO_LEVEL says (test hook)
    static constexpr bool kTwoLevel = true;
    static constexpr bool kBvh4 = false;
    RL_DEV explicit TravStack2(const TravStackT<false>& s) : TravStackT<false>(s) {}
};

// One inner node: both child boxes against the ray, verdicts already folded with the current closest hit.
// `AABB::intersect` clips the far plane with the ray's tfar and the caller then asks `d < its.t` (accel.rs:262-284); its.t never
// exceeds tfar, so clipping with its.t instead gives `!(min(far planes, its.t) <= t_min)` = "box hit AND t_min < its.t" in one
// compare (min is exact and its.t is never a NaN).  NodeFetchLds reads the six near and six far planes through per-ray
// pre-swizzled LDS addresses (the sign of 1/d picks lo or hi once per ray instead of once per plane per node: 12 selects -> 6
// address adds, and the paired left/right planes come from one ds_read2_b32); NodeFetchGlobal keeps the four 16-byte loads.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4))) F4c;      // constant address space: uniform addresses load through the scalar cache
struct NodePlanes { float lnx, lny, lnz, lfx, lfy, lfz, rnx, rny, rnz, rfx, rfy, rfz; int id1, id2; };

struct NodeFetchLds {
    static constexpr bool kHasUniform = false;
    RL_DEV NodePlanes uniform(int) const { return NodePlanes(); }
    const float *nx, *ny, *nz, *fx, *fy, *fz; const int* ids;
    RL_DEV NodeFetchLds(const SceneRecs& recs, V3 inv_d) {
        const float* b = reinterpret_cast<const float*>(recs.nodes);
        const bool sx = inv_d.x < 0.0f, sy = inv_d.y < 0.0f, sz = inv_d.z < 0.0f;
        nx = b + (sx ? 3 : 0); fx = b + (sx ? 0 : 3);
        ny = b + (sy ? 4 : 1); fy = b + (sy ? 1 : 4);
        nz = b + (sz ? 5 : 2); fz = b + (sz ? 2 : 5);
        ids = reinterpret_cast<const int*>(b) + 12;     // two dword reads (an odd record stride is only 4-byte aligned)
    }
    RL_DEV NodePlanes operator()(int cur) const {
        // `cur` is the node's byte offset (child references are pre-multiplied while staging): one add per plane pair
        NodePlanes p;
#define RL_AT(ptr, off) (*reinterpret_cast<const float*>(reinterpret_cast<const char*>(ptr) + cur + 4 * (off)))
        p.lnx = RL_AT(nx, 0); p.rnx = RL_AT(nx, 6); p.lny = RL_AT(ny, 0); p.rny = RL_AT(ny, 6); p.lnz = RL_AT(nz, 0); p.rnz = RL_AT(nz, 6);
        p.lfx = RL_AT(fx, 0); p.rfx = RL_AT(fx, 6); p.lfy = RL_AT(fy, 0); p.rfy = RL_AT(fy, 6); p.lfz = RL_AT(fz, 0); p.rfz = RL_AT(fz, 6);
        p.id1 = __float_as_int(RL_AT(ids, 0)); p.id2 = __float_as_int(RL_AT(ids, 1));
#undef RL_AT
        return p;
    }
};
struct NodeFetchGlobal {
    const float4* nodes; bool sx, sy, sz;
    RL_DEV NodeFetchGlobal(const SceneRecs& recs, V3 inv_d) : nodes(recs.nodes), sx(inv_d.x < 0.0f), sy(inv_d.y < 0.0f), sz(inv_d.z < 0.0f) {}
    RL_DEV NodePlanes operator()(int cur) const {
        const float4* q = nodes + 4 * cur;
        const float4 a = q[0], b = q[1], c = q[2];
        const float4 e = q[3];      // (the compiler narrows this to the two child references, and the triangle quarters to what tri_test reads)
        NodePlanes p;
        p.lnx = sx ? a.w : a.x; p.lfx = sx ? a.x : a.w; p.lny = sy ? b.x : a.y; p.lfy = sy ? a.y : b.x; p.lnz = sz ? b.y : a.z; p.lfz = sz ? a.z : b.y;
        p.rnx = sx ? c.y : b.z; p.rfx = sx ? b.z : c.y; p.rny = sy ? c.z : b.w; p.rfy = sy ? b.w : c.z; p.rnz = sz ? c.w : c.x; p.rfz = sz ? c.x : c.w;
        p.id1 = __float_as_int(e.x); p.id2 = __float_as_int(e.y);
        return p;
    }
    // every calling lane wants the SAME node (`cur` is wave-uniform): the record comes through the scalar data cache (one s_load_dwordx16) instead
    // of four vector loads — it stays off the CU's vector address path, which is what bounds the streaming kernel, and arrives ~3x sooner
    // (profiles/r02_vmem_calibration.jsonl: 21 vs 68 cycles of the CU per record, 0.2 vs 0.7 us per dependent fetch under load)
    static constexpr bool kHasUniform = true;
    RL_DEV NodePlanes uniform(int cur) const {
        const F4c* q = (const F4c*)(nodes) + 4 * cur;
        const f4v a = q[0], b = q[1], c = q[2], e = q[3];
        NodePlanes p;
        p.lnx = sx ? a.w : a.x; p.lfx = sx ? a.x : a.w; p.lny = sy ? b.x : a.y; p.lfy = sy ? a.y : b.x; p.lnz = sz ? b.y : a.z; p.lfz = sz ? a.z : b.y;
        p.rnx = sx ? c.y : b.z; p.rfx = sx ? b.z : c.y; p.rny = sy ? c.z : b.w; p.rfy = sy ? b.w : c.z; p.rnz = sz ? c.w : c.x; p.rfz = sz ? c.x : c.w;
        p.id1 = __float_as_int(e.x); p.id2 = __float_as_int(e.y);
        return p;
    }
};

#ifndef RL_UNIFORM_TRIPS
// streaming scenes: trips on which all lanes hold the same node / leaf fetch it through the scalar cache: 81.2 -> 78.7 ms on the 508 k-triangle scene at 32 spp,
// 75.5 -> 74.4 ms in the tolerance build
#define RL_UNIFORM_TRIPS 1
#endif
#ifndef RL_VOTE_NUM
#define RL_VOTE_NUM 3      // streaming scenes: a node trip while (lanes with node / stack work) * DEN >= NUM * (lanes holding leaves)
#define RL_VOTE_DEN 2
#endif
#ifndef RL_VOTE4_NUM
#define RL_VOTE4_NUM 3     // BVH4 form of the vote (traverse4)
#define RL_VOTE4_DEN 2
#endif
template <bool ANY_HIT, class Stack>
RL_DEV bool traverse4(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar, Hit& hit, const Stack& st);
template <bool ANY_HIT, class Stack>
RL_DEV bool traverse2(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar, Hit& hit, const Stack& st);
// the node array `traverse` reads through a stack of type Stack on a scene that streams its BVH (root: sc.root4 for the BVH4, else sc.root)
template <class Stack>
RL_DEV const float4* streamed_nodes(const DeviceScene& sc) {
    return Stack::kBvh4 ? reinterpret_cast<const float4*>(sc.nodes4) : (Stack::kTwoLevel ? reinterpret_cast<const float4*>(sc.nodes2) : reinterpret_cast<const float4*>(sc.nodes));
}

template <bool ANY_HIT, class Stack>
RL_DEV bool traverse(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar,
                     Hit& hit, const Stack& st) {
    if constexpr (Stack::kBvh4) return traverse4<ANY_HIT>(recs, root, root_lo, root_hi, o, d, tnear, tfar, hit, st);
    if constexpr (Stack::kTwoLevel) return traverse2<ANY_HIT>(recs, root, root_lo, root_hi, o, d, tnear, tfar, hit, st);
    V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float dummy;
    int cur = root >= 0 ? root * Stack::kNodeRefScale : root;
    if (!slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = RL_CHILD_NONE;   // accel.rs:293-295 / 338-340
    int sp = 0;
    bool found = false;
    typename Stack::NodeFetch fetch(recs, inv_d);
    // Same visits, same order as the reference's recursion, one loop level less than a literal transcription: the far child is stored
    // unconditionally (the slot only counts when both children were hit) and a lane that needs the next stack entry takes ONE per trip of the
    // node loop — a stale entry (its subtree lies behind the current hit) simply keeps the lane in the popping state — instead of spinning in a
    // nested loop the rest of the wave waits for.
    constexpr int kPop = -1;            // = ~0: a leaf code with zero triangles, which the builder never emits
    // one trip over inner nodes / stack entries for a lane in that state
    auto node_trip = [&]() {
        if (cur >= 0) {
            hit.steps++;
            NodePlanes p;
            const int cur0 = __builtin_amdgcn_readfirstlane(cur);
#if RL_UNIFORM_TRIPS >= 2
            // experiment: trips on which the live lanes hold exactly TWO nodes fetch both through the scalar cache and select per lane
            const unsigned long long other = Stack::NodeFetch::kHasUniform ? __ballot(cur != cur0) : ~0ull;
            if (Stack::NodeFetch::kHasUniform && other == 0ull) p = fetch.uniform(cur0);
            else if (Stack::NodeFetch::kHasUniform) {
                const int cur1 = __builtin_amdgcn_readlane(cur, (int)__builtin_ctzll(other));
                if (__ballot((cur != cur0) & (cur != cur1)) == 0ull) {
                    const NodePlanes a = fetch.uniform(cur0), b = fetch.uniform(cur1);
                    const bool f = cur == cur0;
                    p.lnx = f ? a.lnx : b.lnx; p.lny = f ? a.lny : b.lny; p.lnz = f ? a.lnz : b.lnz; p.lfx = f ? a.lfx : b.lfx; p.lfy = f ? a.lfy : b.lfy; p.lfz = f ? a.lfz : b.lfz;
                    p.rnx = f ? a.rnx : b.rnx; p.rny = f ? a.rny : b.rny; p.rnz = f ? a.rnz : b.rnz; p.rfx = f ? a.rfx : b.rfx; p.rfy = f ? a.rfy : b.rfy; p.rfz = f ? a.rfz : b.rfz;
                    p.id1 = f ? a.id1 : b.id1; p.id2 = f ? a.id2 : b.id2;
                } else p = fetch(cur);
            } else p = fetch(cur);
#else
            if (Stack::NodeFetch::kHasUniform && RL_UNIFORM_TRIPS && __ballot(cur != cur0) == 0ull) p = fetch.uniform(cur0);
            else p = fetch(cur);
#endif
            const float d1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((p.lnx - o.x) * inv_d.x, (p.lny - o.y) * inv_d.y), (p.lnz - o.z) * inv_d.z), tnear);
            const float f1 = __builtin_fminf(__builtin_fminf(__builtin_fminf((p.lfx - o.x) * inv_d.x, (p.lfy - o.y) * inv_d.y), (p.lfz - o.z) * inv_d.z), hit.t);
            const float d2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((p.rnx - o.x) * inv_d.x, (p.rny - o.y) * inv_d.y), (p.rnz - o.z) * inv_d.z), tnear);
            const float f2 = __builtin_fminf(__builtin_fminf(__builtin_fminf((p.rfx - o.x) * inv_d.x, (p.rfy - o.y) * inv_d.y), (p.rfz - o.z) * inv_d.z), hit.t);
            const bool v1 = !(f1 <= d1), v2 = !(f2 <= d2);                 // box hit and entry distance < its.t
            // the reference orders by distance with a missed box at +inf and keeps the left child first on ties
            const bool right_first = v2 & (!v1 | (d1 > d2));
            st.push(sp, right_first ? p.id1 : p.id2, right_first ? d1 : d2, v1 && v2);
            sp += (v1 && v2) ? 1 : 0;
            cur = (v1 || v2) ? (right_first ? p.id2 : p.id1) : kPop;
        }
        if (cur == kPop) {
            cur = RL_CHILD_NONE;
            if (sp > 0) {
                sp--;
                int code; float dist;
                st.get(sp, &code, &dist);
                cur = dist < hit.t ? code : kPop;      // `if d2 < its.t` evaluated after the near subtree (accel.rs:279-284)
            }
        }
    };
    // a leaf (<= 2 triangles, tested in order: accel.rs:245-254); returns true when an any-hit query is decided
    auto leaf_visit = [&]() -> bool {
        const unsigned int code = (unsigned int)(~cur);
        const int first = (int)(code >> 2), count = (int)(code & 3u);
        // every calling lane holds the SAME leaf (wave-uniform code): its triangle records come through the scalar cache as well; one test site
        // for both forms (a second inlined copy of tri_test upset the register allocation of the whole loop: 75 -> 93 ms in the tolerance build)
        bool uni = false;
        if (Stack::NodeFetch::kHasUniform && RL_UNIFORM_TRIPS) uni = __ballot(cur != __builtin_amdgcn_readfirstlane(cur)) == 0ull;
        for (int k = 0; k < count; k++) {
            hit.tris++;
            float4 q0, q1, q2, q3;
            if (uni) {
                const F4c* q = (const F4c*)(recs.tris) + 4 * (__builtin_amdgcn_readfirstlane(first) + k);
                const f4v a = q[0], b = q[1], c = q[2], e = q[3];
                q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w); q2 = make_float4(c.x, c.y, c.z, c.w); q3 = make_float4(e.x, e.y, e.z, e.w);
            } else {
                const float4* q = recs.tris + Stack::kTriStride4 * (first + k);
                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
            }
            if (tri_test(q0, q1, q2, q3, o, d, hit, first + k)) {
                found = true;
                if (ANY_HIT) return true;
            }
        }
        cur = kPop;
        return false;
    };
    if (Stack::kLdsOnly) {
        // scenes staged in LDS (instruction-issue bound): "while-while" — every lane first descends inner nodes (or takes stack entries) until it
        // holds a leaf (or is done), then the wave tests leaves together
        while (cur != RL_CHILD_NONE) {
            while (cur >= 0 || cur == kPop) node_trip();
            if (cur != RL_CHILD_NONE && leaf_visit()) return true;
        }
    } else {
        // scenes that stream their BVH (bound by vector-memory instructions: a trip costs the same address-path cycles for 3 live lanes as for
        // 60): which kind of trip runs next is VOTED by the wave — a node trip while the lanes holding inner nodes / stack work are at least 1.5 x
        // the lanes holding leaves, a leaf trip otherwise; the minority waits instead of dragging the wave through a sparsely filled trip
        // (508 k triangles, 32 spp: node : leaf >= 0 (while-while) / 1/4 / 1/2 / 1 / 3/2 / 2 / 4 / 8 / leaf-first = 97.0 / 91.7 / 87.1 / 83.6 / 81.8 /
        // 82.5 / 84.8 / 89.3 / 98.7 ms, both kinds in every trip: 98.1 ms; a different ratio for shadow rays — 1/2, 1, 2, 3 with 3/2 for extension rays:
        // 84.3 / 82.7 / 81.7 / 82.2 vs 81.8 ms.  On LDS-staged scenes the ballots cost more than they save: 62 vs 50 ms; so does the lighter form that
        // only leaves the nested node loop early, when the lanes still in it are fewer than 1/8 ... 1 x the lanes waiting: 56.1 ... 60.1 vs 49.5 ms)
#if defined(RL_TRAVERSE_SPARSE)
        // k_stream_chain (chain_stream.hip defines this): one or two live lanes per wave, bound by the latency of the dependent fetches — a vote would only make
        // one chain wait for the other's trip, so every lane takes the trip it needs
        while (cur != RL_CHILD_NONE) {
            if (cur >= 0 || cur == kPop) node_trip();
            else if (leaf_visit()) return true;
        }
#else
        for (;;) {
            const bool in_node = cur >= 0 || cur == kPop;
            const bool in_leaf = !in_node && cur != RL_CHILD_NONE;
            const int n_node = __popcll(__ballot(in_node)), n_leaf = __popcll(__ballot(in_leaf));
            if (n_node + n_leaf == 0) break;
            if (n_node > 0 && n_node * RL_VOTE_DEN >= RL_VOTE_NUM * n_leaf) { if (in_node) node_trip(); }
            else if (in_leaf && leaf_visit()) return true;
        }
#endif
    }
    if (!ANY_HIT && found) {   // barycentrics of the closest hit (see tri_test)
        const float4* q = recs.tris + Stack::kTriStride4 * hit.prim;
        tri_uv(q[0], q[1], q[2], q[3], o, d, hit.t, &hit.u, &hit.v);
    }
    return found;
}

// ------------------------------------------------------------------------------------------------------------------------------
// traverse2 — the exact build's traversal of scenes that stream their BVH (round 5): two levels of the reference's recursion per fetched record
// (device_types.h: BvhNode2; host: two_level_nodes).  The streaming kernels are bound by the latency of a ray's DEPENDENT record fetches (~19 per ray on the
// 508 k-triangle scene; profiles/NEGATIVES.md round 4); a record that carries the grandchildren's exact boxes halves the chain at 2 x the bytes per fetch.
//
// Same visits, same order, same bits as `traverse` — by construction:
//  * AABB::intersect (src/structure.rs:849-869) is a pure function of (ray, tnear, tfar, box); its.t enters only through `if d < its.t` (accel.rs:277-284), folded
//    into the far clip exactly as in `traverse`.  Between a node's trip and its near child's trip no triangle is tested, so its.t — hence every verdict and
//    distance of the child's trip — is already decided when the node's record arrives.
//  * the child boxes are not stored: child c is the min / max union of slots 2c, 2c + 1 (checked per node on the host).  The per-axis plane -> distance map
//    g(p) = (p - o) * (1 / d) is monotone under round-to-nearest (non-decreasing for 1 / d > 0, non-increasing for 1 / d < 0) whenever 1 / d is finite and non-zero
//    and o is finite, and no NaN arises then (box planes are never NaN; an empty slot's +-inf planes map to +-inf), so g(min(p, q)) = min(g(p), g(q)) exactly:
//    the child's near distances are the minima of its slots' near distances, its far distances the maxima.  A ray with a zero / infinite / NaN direction component
//    or a non-finite origin (0 * inf = NaN planes the reference's compare chain skips) takes the union of the PLANES and then the one-level arithmetic (`unsafe`).
//  * second level: the entered child's two slots with the same tnear and the same its.t — the one-level trip at that child, operation for operation.
template <bool ANY_HIT, class Stack>
RL_DEV bool traverse2(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar, Hit& hit, const Stack& st) {
    constexpr bool LDS = Stack::kLdsOnly;       // the records are staged in LDS (stage_scene_lds<true>: 36 dwords apart, inner references = byte offsets)
    const V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float dummy;
    int cur = (LDS && root >= 0) ? root * (4 * kLdsNode2Stride) : root;
    if (!slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = RL_CHILD_NONE;   // accel.rs:293-295 / 338-340
    int sp = 0;
    bool found = false;
    constexpr int kPop = -1;
    const bool sx = inv_d.x < 0.0f, sy = inv_d.y < 0.0f, sz = inv_d.z < 0.0f;
    const float ax = __builtin_fabsf(inv_d.x), ay = __builtin_fabsf(inv_d.y), az = __builtin_fabsf(inv_d.z);
    const float inf = f32_inf();
    const bool unsafe = !((ax > 0.0f) & (ax < inf) & (ay > 0.0f) & (ay < inf) & (az > 0.0f) & (az < inf) &
                          (__builtin_fabsf(o.x) < inf) & (__builtin_fabsf(o.y) < inf) & (__builtin_fabsf(o.z) < inf));
    // LDS: the sign of 1 / d picks the near / far plane row once per ray (byte offsets into a record: lox 0, loy 16, loz 32, hix 48, hiy 64, hiz 80, slots 96, children 112)
    const int onx = sx ? 48 : 0, ofx = sx ? 0 : 48, ony = sy ? 64 : 16, ofy = sy ? 16 : 64, onz = sz ? 80 : 32, ofz = sz ? 32 : 80;
    auto node_trip = [&]() {
        if (cur >= 0) {
            hit.steps++;
            f4v npx, npy, npz, fpx, fpy, fpz, sl, ch;       // near / far PLANES of the four slots per axis, slot references, child references
            if constexpr (LDS) {
                typedef __attribute__((address_space(3))) const char LdsBytes;
                LdsBytes* b = (LdsBytes*)(recs.nodes) + cur;
#define RL_ROW(off) (*reinterpret_cast<__attribute__((address_space(3))) const f4v*>(b + (off)))
                npx = RL_ROW(onx); fpx = RL_ROW(ofx); npy = RL_ROW(ony); fpy = RL_ROW(ofy); npz = RL_ROW(onz); fpz = RL_ROW(ofz); sl = RL_ROW(96); ch = RL_ROW(112);
#undef RL_ROW
            } else {
                f4v lox, loy, loz, hix, hiy, hiz;
                const int cur0 = __builtin_amdgcn_readfirstlane(cur);
                if (RL_UNIFORM_TRIPS && __ballot(cur != cur0) == 0ull) {      // every lane holds the same node: through the scalar cache
                    const F4c* q = (const F4c*)(recs.nodes) + 8 * cur0;
                    lox = q[0]; loy = q[1]; loz = q[2]; hix = q[3]; hiy = q[4]; hiz = q[5]; sl = q[6]; ch = q[7];
                } else {
                    const f4v* q = reinterpret_cast<const f4v*>(recs.nodes) + 8 * cur;
                    lox = q[0]; loy = q[1]; loz = q[2]; hix = q[3]; hiy = q[4]; hiz = q[5]; sl = q[6]; ch = q[7];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    npx[k] = sx ? hix[k] : lox[k]; fpx[k] = sx ? lox[k] : hix[k];
                    npy[k] = sy ? hiy[k] : loy[k]; fpy[k] = sy ? loy[k] : hiy[k];
                    npz[k] = sz ? hiz[k] : loz[k]; fpz[k] = sz ? loz[k] : hiz[k];
                }
            }
            float nx[4], ny[4], nz[4], fx[4], fy[4], fz[4], tn[4], tf[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                nx[k] = (npx[k] - o.x) * inv_d.x; fx[k] = (fpx[k] - o.x) * inv_d.x;
                ny[k] = (npy[k] - o.y) * inv_d.y; fy[k] = (fpy[k] - o.y) * inv_d.y;
                nz[k] = (npz[k] - o.z) * inv_d.z; fz[k] = (fpz[k] - o.z) * inv_d.z;
                tn[k] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx[k], ny[k]), nz[k]), tnear);
                tf[k] = __builtin_fminf(__builtin_fminf(fx[k], fy[k]), fz[k]);       // (unclipped: the clip with its.t is applied where the level is decided)
            }
            // first level: the two children, each the union of its two slots
            float d1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(nx[0], nx[1]), __builtin_fminf(ny[0], ny[1])), __builtin_fminf(nz[0], nz[1])), tnear);
            float f1 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(fx[0], fx[1]), __builtin_fmaxf(fy[0], fy[1])), __builtin_fmaxf(fz[0], fz[1])), hit.t);
            float d2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(nx[2], nx[3]), __builtin_fminf(ny[2], ny[3])), __builtin_fminf(nz[2], nz[3])), tnear);
            float f2 = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fmaxf(fx[2], fx[3]), __builtin_fmaxf(fy[2], fy[3])), __builtin_fmaxf(fz[2], fz[3])), hit.t);
            if (__ballot(unsafe) != 0ull) {       // (wave-uniform; a ray along an axis: ~2^-23 of the cosine-sampled directions)
                if (unsafe) {
                    // the union of the planes (AABB::union_aabb: the near plane of the union is the max of the hi planes when 1 / d < 0, else the min of the lo planes),
                    // then the one-level trip's arithmetic on the child boxes
#define RL_UN(S, P, A, B) (S ? __builtin_fmaxf(P[A], P[B]) : __builtin_fminf(P[A], P[B]))
#define RL_UF(S, P, A, B) (S ? __builtin_fminf(P[A], P[B]) : __builtin_fmaxf(P[A], P[B]))
                    d1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((RL_UN(sx, npx, 0, 1) - o.x) * inv_d.x, (RL_UN(sy, npy, 0, 1) - o.y) * inv_d.y), (RL_UN(sz, npz, 0, 1) - o.z) * inv_d.z), tnear);
                    f1 = __builtin_fminf(__builtin_fminf(__builtin_fminf((RL_UF(sx, fpx, 0, 1) - o.x) * inv_d.x, (RL_UF(sy, fpy, 0, 1) - o.y) * inv_d.y), (RL_UF(sz, fpz, 0, 1) - o.z) * inv_d.z), hit.t);
                    d2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((RL_UN(sx, npx, 2, 3) - o.x) * inv_d.x, (RL_UN(sy, npy, 2, 3) - o.y) * inv_d.y), (RL_UN(sz, npz, 2, 3) - o.z) * inv_d.z), tnear);
                    f2 = __builtin_fminf(__builtin_fminf(__builtin_fminf((RL_UF(sx, fpx, 2, 3) - o.x) * inv_d.x, (RL_UF(sy, fpy, 2, 3) - o.y) * inv_d.y), (RL_UF(sz, fpz, 2, 3) - o.z) * inv_d.z), hit.t);
#undef RL_UN
#undef RL_UF
                }
            }
            const bool v1 = !(f1 <= d1), v2 = !(f2 <= d2);                 // box hit and entry distance < its.t
            const bool right_first = v2 & (!v1 | (d1 > d2));               // missed box = +inf, ties keep the left child first
            const bool both = v1 & v2, any = v1 | v2;
            const int c0 = __float_as_int(ch[0]), c1 = __float_as_int(ch[1]);
            st.push(sp, right_first ? c0 : c1, right_first ? d1 : d2, both);
            sp += both ? 1 : 0;
            // second level: the entered child's own trip (accel.rs:256-287 once more), decided now — its.t cannot change before it
            const int s0 = __float_as_int(sl[0]), s1 = __float_as_int(sl[1]), s2 = __float_as_int(sl[2]), s3 = __float_as_int(sl[3]);
            const int sa = right_first ? s2 : s0, sb = right_first ? s3 : s1;
            const float da = right_first ? tn[2] : tn[0], db = right_first ? tn[3] : tn[1];
            const float fa = __builtin_fminf(right_first ? tf[2] : tf[0], hit.t), fb = __builtin_fminf(right_first ? tf[3] : tf[1], hit.t);
            const bool expanded = sb != RL_CHILD_NONE;                     // else slot a is the child itself (a leaf, or an inner node the host left whole)
            const bool va = !(fa <= da), vb = !(fb <= db);
            const bool rf2 = vb & (!va | (da > db));
            const bool both2 = expanded & any & va & vb;
            st.push(sp, rf2 ? sa : sb, rf2 ? da : db, both2);
            sp += both2 ? 1 : 0;
            cur = !any ? kPop : (!expanded ? sa : ((va | vb) ? (rf2 ? sb : sa) : kPop));
        }
        if (cur == kPop) {
            cur = RL_CHILD_NONE;
            if (sp > 0) {
                sp--;
                int code; float dist;
                st.get(sp, &code, &dist);
                cur = dist < hit.t ? code : kPop;      // `if d2 < its.t` evaluated after the near subtree (accel.rs:279-284)
            }
        }
    };
    auto leaf_visit = [&]() -> bool {
        const unsigned int code = (unsigned int)(~cur);
        const int first = (int)(code >> 2), count = (int)(code & 3u);
        bool uni = false;
        if (!LDS && RL_UNIFORM_TRIPS) uni = __ballot(cur != __builtin_amdgcn_readfirstlane(cur)) == 0ull;
        for (int k = 0; k < count; k++) {
            hit.tris++;
            float4 q0, q1, q2, q3;
            if (uni) {
                const F4c* q = (const F4c*)(recs.tris) + 4 * (__builtin_amdgcn_readfirstlane(first) + k);
                const f4v a = q[0], b = q[1], c = q[2], e = q[3];
                q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w); q2 = make_float4(c.x, c.y, c.z, c.w); q3 = make_float4(e.x, e.y, e.z, e.w);
            } else {
                const float4* q = recs.tris + Stack::kTriStride4 * (first + k);
                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
            }
            if (tri_test(q0, q1, q2, q3, o, d, hit, first + k)) {
                found = true;
                if (ANY_HIT) return true;
            }
        }
        cur = kPop;
        return false;
    };
    if constexpr (LDS) {
        while (cur != RL_CHILD_NONE) {          // "while-while", as in traverse
            while (cur >= 0 || cur == kPop) node_trip();
            if (cur != RL_CHILD_NONE && leaf_visit()) return true;
        }
    } else {
#if defined(RL_TRAVERSE_SPARSE)
        while (cur != RL_CHILD_NONE) {          // k_stream_chain: no vote (see traverse)
            if (cur >= 0 || cur == kPop) node_trip();
            else if (leaf_visit()) return true;
        }
#else
        for (;;) {
            const bool in_node = cur >= 0 || cur == kPop;
            const bool in_leaf = !in_node && cur != RL_CHILD_NONE;
            const int n_node = __popcll(__ballot(in_node)), n_leaf = __popcll(__ballot(in_leaf));
            if (n_node + n_leaf == 0) break;
            if (n_node > 0 && n_node * RL_VOTE_DEN >= RL_VOTE_NUM * n_leaf) { if (in_node) node_trip(); }
            else if (in_leaf && leaf_visit()) return true;
        }
#endif
    }
    if (!ANY_HIT && found) {
        const float4* q = recs.tris + Stack::kTriStride4 * hit.prim;
        tri_uv(q[0], q[1], q[2], q[3], o, d, hit.t, &hit.u, &hit.v);
    }
    return found;
}

// ------------------------------------------------------------------------------------------------------------------------------
// traverse4 — the tolerance build's traversal of scenes that stream their BVH: quantised BVH4 nodes (device_types.h: Bvh4Node; host: build_bvh4).
// The streaming kernel is bound by vector-memory INSTRUCTIONS (profiles/r02_vmem_calibration.jsonl: >= 17 cycles of the CU's address path per
// wave64 load however few lanes are live), four per 64-byte record; a BVH4 node decides four children with one record, so a ray makes about half
// the node trips.  The boxes are conservative (they contain the BVH2 boxes), so the leaves visited are a superset of the exact build's and the
// closest hit is the same except where two candidates tie or a hit grazes a box face — differences `numerics = fast` is allowed.
// Order: the (up to four) entered children are sorted by entry distance, the nearest is descended, the others are pushed farthest first with
// their entry distances and re-checked against the closest hit when popped.  Trips are voted by the wave like in `traverse`.
RL_DEV float ubyte_f(unsigned v, int k) { return (float)((v >> (8 * k)) & 0xffu); }     // v_cvt_f32_ubyteK
template <bool ANY_HIT, class Stack>
RL_DEV bool traverse4(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar, Hit& hit, const Stack& st) {
    const V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float dummy;
    int cur = root;
    if (!slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = RL_CHILD_NONE;
    int sp = 0;
    bool found = false;
    constexpr int kPop = -1;
    const bool sx = inv_d.x < 0.0f, sy = inv_d.y < 0.0f, sz = inv_d.z < 0.0f;
    auto node_trip = [&]() {
        if (cur >= 0) {
            hit.steps++;
            uint4 q0, q1, q2; uint2 q3;
            const int cur0 = __builtin_amdgcn_readfirstlane(cur);
            if (RL_UNIFORM_TRIPS && __ballot(cur != cur0) == 0ull) {      // every lane holds the same node: through the scalar cache
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                typedef const u4v __attribute__((address_space(4))) U4c;
                const U4c* q = (const U4c*)(recs.nodes) + 4 * cur0;
                const u4v a = q[0], b = q[1], c = q[2], e = q[3];
                q0 = make_uint4(a.x, a.y, a.z, a.w); q1 = make_uint4(b.x, b.y, b.z, b.w); q2 = make_uint4(c.x, c.y, c.z, c.w); q3 = make_uint2(e.x, e.y);
            } else {
                const uint4* q = reinterpret_cast<const uint4*>(recs.nodes) + 4 * cur;
                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = *reinterpret_cast<const uint2*>(q + 3);
            }
            // plane distance t = (org + q * step - o) * inv_d = A + q * B per axis
            const float stx = __uint_as_float((q0.w & 0xffu) << 23), sty = __uint_as_float(((q0.w >> 8) & 0xffu) << 23), stz = __uint_as_float(((q0.w >> 16) & 0xffu) << 23);
            const float Ax = (__uint_as_float(q0.x) - o.x) * inv_d.x, Ay = (__uint_as_float(q0.y) - o.y) * inv_d.y, Az = (__uint_as_float(q0.z) - o.z) * inv_d.z;
            const float Bx = stx * inv_d.x, By = sty * inv_d.y, Bz = stz * inv_d.z;
            // qlo = (q1.x, q1.y, q1.z), qhi = (q1.w, q2.x, q2.y); the sign of 1/d picks the near / far plane set once per axis
            const unsigned nx = sx ? q1.w : q1.x, fx = sx ? q1.x : q1.w, ny = sy ? q2.x : q1.y, fy = sy ? q1.y : q2.x, nz = sz ? q2.y : q1.z, fz = sz ? q1.z : q2.y;
            const int ids[4] = {(int)q2.z, (int)q2.w, (int)q3.x, (int)q3.y};
            float dist[4]; int code[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(Ax + ubyte_f(nx, k) * Bx, Ay + ubyte_f(ny, k) * By), Az + ubyte_f(nz, k) * Bz), tnear);
                const float tf = __builtin_fminf(__builtin_fminf(__builtin_fminf(Ax + ubyte_f(fx, k) * Bx, Ay + ubyte_f(fy, k) * By), Az + ubyte_f(fz, k) * Bz), hit.t);
                const bool in = !(tf <= tn) & (ids[k] != RL_CHILD_NONE);
                dist[k] = in ? tn : f32_inf();
                code[k] = ids[k];
            }
            // sort the four (distance, child) pairs, nearest first (5 compare-exchanges; missed children carry +inf)
#define RL_CE(i, j) { const bool sw = dist[j] < dist[i]; const float td = sw ? dist[j] : dist[i]; dist[j] = sw ? dist[i] : dist[j]; dist[i] = td; \
                      const int tc = sw ? code[j] : code[i]; code[j] = sw ? code[i] : code[j]; code[i] = tc; }
#if defined(RL_BVH4_NOSORT)
            RL_CE(0, 1) RL_CE(0, 2) RL_CE(0, 3)      // experiment: only the nearest child is found, the others are pushed in slot order
#else
            RL_CE(0, 1) RL_CE(2, 3) RL_CE(0, 2) RL_CE(1, 3) RL_CE(1, 2)
#endif
#undef RL_CE
            // farthest first onto the stack, so that the nearest pending child is popped first
            if (dist[3] < f32_inf()) { st.push(sp, code[3], dist[3]); sp++; }
            if (dist[2] < f32_inf()) { st.push(sp, code[2], dist[2]); sp++; }
            if (dist[1] < f32_inf()) { st.push(sp, code[1], dist[1]); sp++; }
            cur = dist[0] < f32_inf() ? code[0] : kPop;
        }
        if (cur == kPop) {
            cur = RL_CHILD_NONE;
            if (sp > 0) {
                sp--;
                int c; float dd;
                st.get(sp, &c, &dd);
                cur = dd < hit.t ? c : kPop;
            }
        }
    };
    auto leaf_visit = [&]() -> bool {
        const unsigned int code = (unsigned int)(~cur);
        const int first = (int)(code >> 2), count = (int)(code & 3u);
        const bool uni = RL_UNIFORM_TRIPS && __ballot(cur != __builtin_amdgcn_readfirstlane(cur)) == 0ull;
        for (int k = 0; k < count; k++) {
            hit.tris++;
            float4 q0, q1, q2, q3;
            if (uni) {
                const F4c* q = (const F4c*)(recs.tris) + 4 * (__builtin_amdgcn_readfirstlane(first) + k);
                const f4v a = q[0], b = q[1], c = q[2], e = q[3];
                q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w); q2 = make_float4(c.x, c.y, c.z, c.w); q3 = make_float4(e.x, e.y, e.z, e.w);
            } else {
                const float4* q = recs.tris + 4 * (first + k);
                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3];
            }
            if (tri_test(q0, q1, q2, q3, o, d, hit, first + k)) {
                found = true;
                if (ANY_HIT) return true;
            }
        }
        cur = kPop;
        return false;
    };
#if defined(RL_TRAVERSE_SPARSE)
    while (cur != RL_CHILD_NONE) {          // k_stream_chain: no vote (see traverse)
        if (cur >= 0 || cur == kPop) node_trip();
        else if (leaf_visit()) return true;
    }
#else
    for (;;) {
        const bool in_node = cur >= 0 || cur == kPop;
        const bool in_leaf = !in_node && cur != RL_CHILD_NONE;
        const int n_node = __popcll(__ballot(in_node)), n_leaf = __popcll(__ballot(in_leaf));
        if (n_node + n_leaf == 0) break;
        if (n_node > 0 && n_node * RL_VOTE4_DEN >= RL_VOTE4_NUM * n_leaf) { if (in_node) node_trip(); }
        else if (in_leaf && leaf_visit()) return true;
    }
#endif
    if (!ANY_HIT && found) {
        const float4* q = recs.tris + 4 * hit.prim;
        tri_uv(q[0], q[1], q[2], q[3], o, d, hit.t, &hit.u, &hit.v);
    }
    return found;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Wave-cooperative traversal for scenes that stream their BVH from L2 / HBM (the persistent kernel on the 508 k-triangle scene).
//
// EXPERIMENT, OFF BY DEFAULT (RL_COOP_FETCH = 0; bit-identical images, but 144 ms (LDS staging) / 145 ms (registers + ds_bpermute) vs 97-110 ms at
// 1080p x 32 spp: DESIGN.md §4).
// The per-lane form is bound by the instruction throughput of the CU's vector-memory address path (TA busy 81-91 % of the kernel,
// profiles/r02_ta_living_room.json; calibrated in profiles/r02_vmem_calibration.jsonl: a wave64 `global_load_dwordx4` holds that path for >= 17
// cycles however few lanes are live, + 0.45 / 2.3 cycles per live lane from L2 / the Infinity Cache), and a node costs four such loads.  This form
// replaces them by one or two full ones and still loses: finding, ranking and redistributing the records costs two LDS round trips (or 14-17
// ds_bpermute) per trip and ballot-driven loops every lane stays in, more than the saved load instructions give back.
// Here the live lanes' 64-byte records are fetched by the WHOLE wave: the wanting lanes are ranked (ballot + mbcnt), their record indices travel to the rank slots with one ds_permute,
// lane L then loads quarter L % 4 of the record of rank L / 4 — a quad reads 64 contiguous bytes, one request — into a per-wave LDS
// staging area (32 records = 2 KB), and every wanting lane reads its own record back from LDS, the planes through the same
// sign-swizzled addresses the LDS-staged scenes use.  One or two coalesced load instructions per trip instead of four scattered
// ones; lanes ranked 32 or higher simply take their turn on the next trip (the wave is well filled then).
// Every lane of the wave must call this in step (wave-uniform control flow): the loops below are driven by ballots and a lane
// without a ray (or whose ray is done) stays as a loader.  Visits, order and arithmetic per ray are those of `traverse`.
static constexpr int kCoopRecords = 32;                         // records staged per wave and trip
static constexpr int kCoopStageFloat4s = 4 * kCoopRecords;      // per wave
RL_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
RL_DEV const float* coop_fetch64(const float4* base, int idx, bool want, float4* stage) {
    const unsigned long long mask = __ballot(want);
    const unsigned lane = lane_id();
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));   // wanting lanes below this one
    const unsigned n = (unsigned)__popcll(mask);
    // record indices by rank: wanting lanes send theirs to slots 0 .. n-1, the others fill n .. 63 (a full permutation)
    const unsigned slot = want ? below : n + (lane - below);
    const int by_rank = __builtin_amdgcn_ds_permute((int)(slot << 2), idx);
    const unsigned k0 = lane >> 2, q = lane & 3u;
    const int id0 = __builtin_amdgcn_ds_bpermute((int)(k0 << 2), by_rank);
    int id1 = 0;
    if (n > 16u) id1 = __builtin_amdgcn_ds_bpermute((int)((k0 + 16u) << 2), by_rank);     // wave-uniform branch
    const bool l0 = k0 < n, l1 = k0 + 16u < n;
    float4 v0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v1 = v0;
    if (l0) v0 = base[4 * (size_t)id0 + q];          // both loads are in flight before either is stored
    if (l1) v1 = base[4 * (size_t)id1 + q];
    if (l0) stage[lane] = v0;
    if (l1) stage[64u + lane] = v1;
    // the records are read back by other lanes of this wave: LDS operations of one wave execute in order, the compiler only has to keep them so
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return (want && below < (unsigned)kCoopRecords) ? reinterpret_cast<const float*>(stage + 4u * below) : nullptr;
}

// The same fetch without LDS memory (RL_COOP_FETCH = 2): the loaded quarters stay in the loader lanes' registers and every owner pulls its NDW
// dwords with ds_bpermute (the LDS crossbar, no storage), so the kernel's LDS budget — and with it 6 waves/SIMD — is untouched.
template <int NDW>
RL_DEV bool coop_fetch_regs(const float4* base, int idx, bool want, float (&out)[NDW]) {
    const unsigned long long mask = __ballot(want);
    const unsigned lane = lane_id();
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const unsigned n = (unsigned)__popcll(mask);
    const unsigned slot = want ? below : n + (lane - below);
    const int by_rank = __builtin_amdgcn_ds_permute((int)(slot << 2), idx);
    const unsigned k0 = lane >> 2, q = lane & 3u;
    const int id0 = __builtin_amdgcn_ds_bpermute((int)(k0 << 2), by_rank);
    int id1 = 0;
    if (n > 16u) id1 = __builtin_amdgcn_ds_bpermute((int)((k0 + 16u) << 2), by_rank);
    float4 v0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v1 = v0;
    if (k0 < n) v0 = base[4 * (size_t)id0 + q];
    if (k0 + 16u < n) v1 = base[4 * (size_t)id1 + q];
    const unsigned src = (below & 15u) << 4;                     // byte address of lane 4 * (rank % 16) in the bpermute address space
#pragma unroll
    for (int f = 0; f < NDW; f++) {
        const int addr = (int)(src + (unsigned)((f >> 2) << 2));
        const float a0 = (f & 3) == 0 ? v0.x : (f & 3) == 1 ? v0.y : (f & 3) == 2 ? v0.z : v0.w;
        float x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a0)));
        if (n > 16u) {                                           // wave-uniform
            const float a1 = (f & 3) == 0 ? v1.x : (f & 3) == 1 ? v1.y : (f & 3) == 2 ? v1.z : v1.w;
            const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a1)));
            x = below >= 16u ? y : x;
        }
        out[f] = x;
    }
    return want && below < 32u;
}

template <bool ANY_HIT, class Stack>
RL_DEV bool traverse_coop(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar,
                          Hit& hit, const Stack& st, bool valid, float4* stage) {
    const V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float dummy;
    int cur = RL_CHILD_NONE;
    if (valid && slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = root;   // accel.rs:293-295 / 338-340
    int sp = 0, leaf_k = 0;
    bool found = false;
    constexpr int kPop = -1;
    const bool sx = inv_d.x < 0.0f, sy = inv_d.y < 0.0f, sz = inv_d.z < 0.0f;
    // dword offsets of the near / far planes of the left child inside a node record (the right child's are 6 further)
    const int onx = sx ? 3 : 0, ofx = sx ? 0 : 3, ony = sy ? 4 : 1, ofy = sy ? 1 : 4, onz = sz ? 5 : 2, ofz = sz ? 2 : 5;
    while (__ballot(cur != RL_CHILD_NONE) != 0ull) {
        // ---- inner nodes and stack entries
        while (__ballot(cur >= 0 || cur == kPop) != 0ull) {
#if RL_COOP_FETCH == 2
            float rec[14];
            const bool got = coop_fetch_regs<14>(recs.nodes, cur, cur >= 0, rec);
            if (got) {
                hit.steps++;
                const float lnx = sx ? rec[3] : rec[0], lfx = sx ? rec[0] : rec[3], lny = sy ? rec[4] : rec[1], lfy = sy ? rec[1] : rec[4], lnz = sz ? rec[5] : rec[2], lfz = sz ? rec[2] : rec[5];
                const float rnx = sx ? rec[9] : rec[6], rfx = sx ? rec[6] : rec[9], rny = sy ? rec[10] : rec[7], rfy = sy ? rec[7] : rec[10], rnz = sz ? rec[11] : rec[8], rfz = sz ? rec[8] : rec[11];
                const int id1 = __float_as_int(rec[12]), id2 = __float_as_int(rec[13]);
#else
            const float* rec = coop_fetch64(recs.nodes, cur, cur >= 0, stage);
            if (rec) {
                hit.steps++;
                const float lnx = rec[onx], rnx = rec[onx + 6], lny = rec[ony], rny = rec[ony + 6], lnz = rec[onz], rnz = rec[onz + 6];
                const float lfx = rec[ofx], rfx = rec[ofx + 6], lfy = rec[ofy], rfy = rec[ofy + 6], lfz = rec[ofz], rfz = rec[ofz + 6];
                const int id1 = __float_as_int(rec[12]), id2 = __float_as_int(rec[13]);
#endif
                const float d1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((lnx - o.x) * inv_d.x, (lny - o.y) * inv_d.y), (lnz - o.z) * inv_d.z), tnear);
                const float f1 = __builtin_fminf(__builtin_fminf(__builtin_fminf((lfx - o.x) * inv_d.x, (lfy - o.y) * inv_d.y), (lfz - o.z) * inv_d.z), hit.t);
                const float d2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((rnx - o.x) * inv_d.x, (rny - o.y) * inv_d.y), (rnz - o.z) * inv_d.z), tnear);
                const float f2 = __builtin_fminf(__builtin_fminf(__builtin_fminf((rfx - o.x) * inv_d.x, (rfy - o.y) * inv_d.y), (rfz - o.z) * inv_d.z), hit.t);
                const bool v1 = !(f1 <= d1), v2 = !(f2 <= d2);
                const bool right_first = v2 & (!v1 | (d1 > d2));
                st.push(sp, right_first ? id1 : id2, right_first ? d1 : d2, v1 && v2);
                sp += (v1 && v2) ? 1 : 0;
                cur = (v1 || v2) ? (right_first ? id2 : id1) : kPop;
            }
            if (cur == kPop) {
                cur = RL_CHILD_NONE;
                if (sp > 0) {
                    sp--;
                    int code; float dist;
                    st.get(sp, &code, &dist);
                    cur = dist < hit.t ? code : kPop;
                }
            }
        }
        // ---- leaves (<= 2 triangles, tested in order: accel.rs:245-254); every lane that still has a ray holds one now
        while (__ballot(cur != RL_CHILD_NONE && cur != kPop) != 0ull) {
            const bool is_leaf = cur != RL_CHILD_NONE && cur != kPop;
            const unsigned int code = (unsigned int)(~cur);
            const int first = (int)(code >> 2), count = (int)(code & 3u);
#if RL_COOP_FETCH == 2
            float rec[13];
            const bool got = coop_fetch_regs<13>(recs.tris, first + leaf_k, is_leaf, rec);
            if (got) {
                hit.tris++;
                const float4 q[4] = {make_float4(rec[0], rec[1], rec[2], rec[3]), make_float4(rec[4], rec[5], rec[6], rec[7]), make_float4(rec[8], rec[9], rec[10], rec[11]), make_float4(rec[12], 0.0f, 0.0f, 0.0f)};
#else
            const float* rec = coop_fetch64(recs.tris, first + leaf_k, is_leaf, stage);
            if (rec) {
                hit.tris++;
                const float4* q = reinterpret_cast<const float4*>(rec);
#endif
                const bool accepted = tri_test(q[0], q[1], q[2], q[3], o, d, hit, first + leaf_k);
                found = found || accepted;
                leaf_k++;
                if (ANY_HIT && accepted) { cur = RL_CHILD_NONE; leaf_k = 0; }
                else if (leaf_k == count) { cur = kPop; leaf_k = 0; }
            }
        }
    }
    if (!ANY_HIT) {   // barycentrics of the closest hit (see tri_test)
        bool need = found;
        while (__ballot(need) != 0ull) {
#if RL_COOP_FETCH == 2
            float rec[13];
            const bool got = coop_fetch_regs<13>(recs.tris, hit.prim, need, rec);
            if (got) {
                const float4 q[4] = {make_float4(rec[0], rec[1], rec[2], rec[3]), make_float4(rec[4], rec[5], rec[6], rec[7]), make_float4(rec[8], rec[9], rec[10], rec[11]), make_float4(rec[12], 0.0f, 0.0f, 0.0f)};
#else
            const float* rec = coop_fetch64(recs.tris, hit.prim, need, stage);
            if (rec) {
                const float4* q = reinterpret_cast<const float4*>(rec);
#endif
                tri_uv(q[0], q[1], q[2], q[3], o, d, hit.t, &hit.u, &hit.v);
                need = false;
            }
        }
    }
    return found;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Lane-parallel records + serial walk (k_stream_chain on tiny LDS-staged scenes, round 3).
// A chain of the draw-count pass traces ONE ray at a time while the other lanes of its group (32 or 64) idle, and 2/3 of the pass is that ray's
// traversal at one instruction per ~5 cycles.  Here the group first evaluates everything about the ray that does not depend on the traversal
// state — for every node the entry distance and the unclipped exit distance of both child boxes, for every triangle the plane distance t and
// the barycentric verdict — one record per lane, into LDS; the chain lane then walks the tree in the reference's order reading those records:
// per node two compares against the current closest hit instead of two slab tests, per triangle one window compare instead of a triangle test.
// Same visits, same order, same arithmetic on the same operands (the clip with hit.t is the outermost min of `traverse`'s expression, the
// distance window the first conjunct of tri_test's) — same bits.
RL_DEV bool tri_inside(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o, V3 d, float t) {
    // the body of tri_test's `if (window)` — keep the two in step (tests: two-pass == single-pass == oracle)
    const V3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    const V3 n = mk3(q0.w, q1.w, q2.w);
    const float det = q3.x;
    const V3 pv = (o + t * d) - v0;
    const V3 u0 = cross(e1, pv);
    const V3 w0 = cross(pv, e2);
    const bool facing = !(dot(u0, n) < 0.0f) & !(dot(w0, n) < 0.0f);
    const float uu = dot(u0, u0), ww = dot(w0, w0);
#if defined(RL_FAST_MATH)
    return facing & (det > 0.0f) & (__builtin_amdgcn_sqrtf(uu) + __builtin_amdgcn_sqrtf(ww) <= det);
#else
#if defined(RL_TRI_REFERENCE_FORM)
    const bool in_range = false; const float s = 0.0f;
#else
    const float s = __builtin_amdgcn_sqrtf(uu) + __builtin_amdgcn_sqrtf(ww);
    const bool in_range = __builtin_fminf(__builtin_fminf(uu, ww), det) >= 0x1p-100f;
#endif
    const bool sure_in = in_range & (s <= det * 0.99999f), sure_out = in_range & (s >= det * 1.00001f);
    bool accept = facing & sure_in;
    if (facing & !sure_in & !sure_out) {
        const float v = div_rn(sqrt_rn(uu), det);
        const float u = div_rn(sqrt_rn(ww), det);
        accept = !(u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) && (u + v <= 1.0f);
    }
    return accept;
#endif
}
// `sub` = this lane's index inside its chain's group of `group` lanes; `active`: the group's chain carries a ray this iteration.
// pre_nodes: 2 float4 per node (d1, f1, d2, f2 | id1, id2 as node indices / leaf codes); pre_tris: (t, inside ? 1 : 0) per triangle.
RL_DEV void precompute_records(const SceneRecs& recs, unsigned n_nodes, unsigned n_tris, V3 o, V3 d, V3 inv_d, float tnear, bool active, unsigned sub, unsigned group,
                               float4* pre_nodes, float2* pre_tris) {
    for (unsigned base = 0; base < n_tris; base += group) {
        const unsigned idx = base + sub;
        if (active && idx < n_tris) {
            const float4* q = recs.tris + kLdsTriStride4 * idx;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const V3 v0 = mk3(q0.x, q0.y, q0.z), n = mk3(q0.w, q1.w, q2.w);
            const float denom = dot(d, n);
            const float t = div_rn(-dot(o - v0, n), denom);
            const bool in = tri_inside(q0, q1, q2, q3, o, d, t);
            pre_tris[idx] = make_float2(t, in ? 1.0f : 0.0f);
        }
    }
    const NodeFetchLds fetch(recs, inv_d);
    for (unsigned base = 0; base < n_nodes; base += group) {
        const unsigned idx = base + sub;
        if (active && idx < n_nodes) {
            const NodePlanes p = fetch((int)(idx * 4u * (unsigned)kLdsNodeStride));
            const float d1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((p.lnx - o.x) * inv_d.x, (p.lny - o.y) * inv_d.y), (p.lnz - o.z) * inv_d.z), tnear);
            const float f1 = __builtin_fminf(__builtin_fminf((p.lfx - o.x) * inv_d.x, (p.lfy - o.y) * inv_d.y), (p.lfz - o.z) * inv_d.z);
            const float d2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((p.rnx - o.x) * inv_d.x, (p.rny - o.y) * inv_d.y), (p.rnz - o.z) * inv_d.z), tnear);
            const float f2 = __builtin_fminf(__builtin_fminf((p.rfx - o.x) * inv_d.x, (p.rfy - o.y) * inv_d.y), (p.rfz - o.z) * inv_d.z);
            const int id1 = p.id1 >= 0 ? p.id1 / (4 * kLdsNodeStride) : p.id1, id2 = p.id2 >= 0 ? p.id2 / (4 * kLdsNodeStride) : p.id2;     // staged references are byte offsets
            pre_nodes[2u * idx] = make_float4(d1, f1, d2, f2);
            pre_nodes[2u * idx + 1u] = make_float4(__int_as_float(id1), __int_as_float(id2), 0.0f, 0.0f);
        }
    }
    // the records are read back by the chain lane of this wave: LDS operations of one wave execute in order, the compiler only has to keep them so
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Acceleration::trace over the precomputed records: `traverse<false>`'s LDS-scene loop with the fetch + slab / triangle arithmetic replaced by reads.
template <class Stack>
RL_DEV bool traverse_pre(const float4* pre_nodes, const float2* pre_tris, const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, V3 inv_d, float tnear, float tfar,
                         Hit& hit, const Stack& st) {
    float dummy;
    int cur = root;
    if (!slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = RL_CHILD_NONE;
    int sp = 0;
    bool found = false;
    constexpr int kPop = -1;
    while (cur != RL_CHILD_NONE) {
        while (cur >= 0 || cur == kPop) {
            if (cur >= 0) {
                hit.steps++;
                const float4 a = pre_nodes[2 * cur], b = pre_nodes[2 * cur + 1];
                const float d1 = a.x, f1 = __builtin_fminf(a.y, hit.t), d2 = a.z, f2 = __builtin_fminf(a.w, hit.t);
                const int id1 = __float_as_int(b.x), id2 = __float_as_int(b.y);
                const bool v1 = !(f1 <= d1), v2 = !(f2 <= d2);
                const bool right_first = v2 & (!v1 | (d1 > d2));
                st.push(sp, right_first ? id1 : id2, right_first ? d1 : d2, v1 && v2);
                sp += (v1 && v2) ? 1 : 0;
                cur = (v1 || v2) ? (right_first ? id2 : id1) : kPop;
            }
            if (cur == kPop) {
                cur = RL_CHILD_NONE;
                if (sp > 0) {
                    sp--;
                    int code; float dist;
                    st.get(sp, &code, &dist);
                    cur = dist < hit.t ? code : kPop;
                }
            }
        }
        if (cur != RL_CHILD_NONE) {
            const unsigned int code = (unsigned int)(~cur);
            const int first = (int)(code >> 2), count = (int)(code & 3u);
            for (int k = 0; k < count; k++) {
                hit.tris++;
                const float2 r = pre_tris[first + k];
                const bool accept = (r.x < hit.t) & (r.x > 0.00001f) & (r.y != 0.0f);
                hit.t = accept ? r.x : hit.t;
                hit.prim = accept ? first + k : hit.prim;
                found = found | accept;
            }
            cur = kPop;
        }
    }
    if (found) {
        const float4* q = recs.tris + kLdsTriStride4 * hit.prim;
        tri_uv(q[0], q[1], q[2], q[3], o, d, hit.t, &hit.u, &hit.v);
    }
    return found;
}

// ------------------------------------------------------------------------------------------------------------------------------
// traverse_treelet — k_stream_chain on scenes that stream their BVH (round 3).  A chain traces one ray at a time and each node / leaf it visits is a
// dependent fetch from L2 / the Infinity Cache (~25 per ray on the 508 k-triangle scene: four fifths of the pass).  The node array this function reads
// is laid out in 1 KB blocks of 16 nodes that hold connected pieces of the tree (host: treelet_blocks): when the chain lane needs a node outside the
// block it has, the `group` lanes of its chain fetch the WHOLE block — 64 x 16 bytes, one or two loads per lane, one round trip — into LDS, and the walk
// then descends up to four levels out of LDS; a leaf's one or two triangle records are fetched the same way.  Visits, order and arithmetic per ray are those
// of `traverse`.  Called by EVERY lane of the wave in step (`has_ray`: this lane is a chain lane with a ray; the others only fetch).
// a chain's whole traversal stack in LDS, one contiguous column per chain (the [level][lane] stacks of the other kernels keep only a few levels there and spill
// the rest to global memory — for a chain every pop from that part is one more dependent round trip)
struct ChainStack {
    int2* base;
    RL_DEV void push(int sp, int code, float dist, bool = true) const { base[sp] = make_int2(code, __float_as_int(dist)); }
    RL_DEV void get(int sp, int* code, float* dist) const { const int2 e = base[sp]; *code = e.x; *dist = __int_as_float(e.y); }
};
template <class Stack>
RL_DEV bool traverse_treelet(const float4* nodes_t, const float4* tris, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar, bool has_ray,
                             unsigned lead, unsigned sub, unsigned group, float4* cache_nodes, float4* cache_tris, Hit& hit, const Stack& st) {
    const V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float dummy;
    int cur = RL_CHILD_NONE;
    if (has_ray && slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = root;
    int sp = 0, cached = -1;
    bool found = false;
    constexpr int kPop = -1;
    const bool sx = inv_d.x < 0.0f, sy = inv_d.y < 0.0f, sz = inv_d.z < 0.0f;
    while (__ballot(cur != RL_CHILD_NONE) != 0ull) {
        // ---- inner node (or stack entry)
        const bool need = cur >= 0 && (cur >> 4) != cached;
        if (__ballot(need) != 0ull) {         // (no cross-lane traffic on the common path: a node of the block the chain already has)
            const int g_blk = __shfl(need ? (cur >> 4) : -1, (int)lead, 64);
            const bool g_need = g_blk >= 0;
            if (g_need) {
                const float4* src = nodes_t + (size_t)g_blk * 64;
                for (unsigned i = sub; i < 64u; i += group) cache_nodes[i] = src[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (need) { cached = cur >> 4; hit.fetches++; }
        }
        if (cur >= 0) {
            hit.steps++;
            const float4* q = cache_nodes + 4 * (cur & 15);
            const float4 a = q[0], b = q[1], c = q[2], e = q[3];
            NodePlanes p;
            p.lnx = sx ? a.w : a.x; p.lfx = sx ? a.x : a.w; p.lny = sy ? b.x : a.y; p.lfy = sy ? a.y : b.x; p.lnz = sz ? b.y : a.z; p.lfz = sz ? a.z : b.y;
            p.rnx = sx ? c.y : b.z; p.rfx = sx ? b.z : c.y; p.rny = sy ? c.z : b.w; p.rfy = sy ? b.w : c.z; p.rnz = sz ? c.w : c.x; p.rfz = sz ? c.x : c.w;
            p.id1 = __float_as_int(e.x); p.id2 = __float_as_int(e.y);
            const float d1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((p.lnx - o.x) * inv_d.x, (p.lny - o.y) * inv_d.y), (p.lnz - o.z) * inv_d.z), tnear);
            const float f1 = __builtin_fminf(__builtin_fminf(__builtin_fminf((p.lfx - o.x) * inv_d.x, (p.lfy - o.y) * inv_d.y), (p.lfz - o.z) * inv_d.z), hit.t);
            const float d2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf((p.rnx - o.x) * inv_d.x, (p.rny - o.y) * inv_d.y), (p.rnz - o.z) * inv_d.z), tnear);
            const float f2 = __builtin_fminf(__builtin_fminf(__builtin_fminf((p.rfx - o.x) * inv_d.x, (p.rfy - o.y) * inv_d.y), (p.rfz - o.z) * inv_d.z), hit.t);
            const bool v1 = !(f1 <= d1), v2 = !(f2 <= d2);
            const bool right_first = v2 & (!v1 | (d1 > d2));
            st.push(sp, right_first ? p.id1 : p.id2, right_first ? d1 : d2, v1 && v2);
            sp += (v1 && v2) ? 1 : 0;
            cur = (v1 || v2) ? (right_first ? p.id2 : p.id1) : kPop;
        }
        if (cur == kPop) {
            cur = RL_CHILD_NONE;
            if (sp > 0) {
                sp--;
                int code; float dist;
                st.get(sp, &code, &dist);
                cur = dist < hit.t ? code : kPop;
            }
        }
        // ---- leaf (<= 2 triangles, tested in order)
        const bool in_leaf = cur != RL_CHILD_NONE && cur != kPop && cur < 0;
        if (__ballot(in_leaf) != 0ull) {
            const unsigned code = (unsigned)(~cur);
            const unsigned g_code = (unsigned)__shfl(in_leaf ? (int)code : 0, (int)lead, 64);      // 0: this group's chain holds no leaf
            const int g_first = (int)(g_code >> 2), g_count = (int)(g_code & 3u);
            if ((int)sub < 4 * g_count) cache_tris[sub] = tris[4 * (size_t)g_first + sub];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (in_leaf) {
                const int first = (int)(code >> 2), count = (int)(code & 3u);
                for (int k = 0; k < count; k++) {
                    hit.tris++;
                    const float4* q = cache_tris + 4 * k;
                    found = found | tri_test(q[0], q[1], q[2], q[3], o, d, hit, first + k);
                }
                cur = kPop;
            }
        }
    }
    if (found) {
        const float4* q = tris + 4 * (size_t)hit.prim;
        tri_uv(q[0], q[1], q[2], q[3], o, d, hit.t, &hit.u, &hit.v);
    }
    return found;
}

// Stage the node / triangle records into LDS (cooperatively) in the padded layout above; inner-child references become dword offsets.
RL_DEV void stage_scene_lds(const DeviceScene& sc, float4* lds_nodes, float4* lds_tris) {
    const float* gn = reinterpret_cast<const float*>(sc.nodes);
    const float4* gt = reinterpret_cast<const float4*>(sc.tris);
    float* ln = reinterpret_cast<float*>(lds_nodes);
    for (unsigned int i = threadIdx.x; i < 16u * sc.n_nodes; i += blockDim.x) {
        const unsigned int node = i >> 4, f = i & 15u;
        if (f >= 14u) continue;                                  // padding words of the 64-byte record
        float v = gn[i];
        if (f >= 12u) { const int id = __float_as_int(v); if (id >= 0) v = __int_as_float(id * 4 * kLdsNodeStride); }   // byte offset
        ln[node * (unsigned)kLdsNodeStride + f] = v;
    }
    for (unsigned int i = threadIdx.x; i < 4u * sc.n_prims; i += blockDim.x) lds_tris[(i >> 2) * (unsigned)kLdsTriStride4 + (i & 3u)] = gt[i];
    __syncthreads();
}

// The same with the nodes as two-level records (device_types.h: BvhNode2) kLdsNode2Stride dwords apart; inner references (slots and children) become byte offsets.
RL_DEV void stage_scene_lds2(const DeviceScene& sc, float4* lds_nodes, float4* lds_tris) {
    const float* gn = reinterpret_cast<const float*>(sc.nodes2);
    const float4* gt = reinterpret_cast<const float4*>(sc.tris);
    float* ln = reinterpret_cast<float*>(lds_nodes);
    for (unsigned int i = threadIdx.x; i < 32u * sc.n_nodes; i += blockDim.x) {
        const unsigned int node = i >> 5, f = i & 31u;
        if (f >= 30u) continue;                                  // padding words of the 128-byte record
        float v = gn[i];
        if (f >= 24u) { const int id = __float_as_int(v); if (id >= 0) v = __int_as_float(id * 4 * kLdsNode2Stride); }   // byte offset
        ln[node * (unsigned)kLdsNode2Stride + f] = v;
    }
    for (unsigned int i = threadIdx.x; i < 4u * sc.n_prims; i += blockDim.x) lds_tris[(i >> 2) * (unsigned)kLdsTriStride4 + (i & 3u)] = gt[i];
    __syncthreads();
}

}  // namespace rl
