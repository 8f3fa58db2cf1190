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

namespace rl {

struct Hit { float t, u, v; int prim; int steps = 0, tris = 0; };   // steps / tris: dev-only traversal statistics (dead code unless read)

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

// Mesh::intersection_tri; returns true and updates `hit` if the triangle is the new closest hit.
// The reference evaluates u, v (two sqrt + two divides) before it looks at `t < its.t && t > 1e-5`
// (geometry.rs:391-398).  All conditions are pure and must hold together, so testing the cheap
// distance window first accepts exactly the same set of hits with the same (t, u, v) bits.
RL_DEV bool tri_test(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 o, V3 d, Hit& hit, int prim) {
    V3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    V3 n = mk3(q0.w, q1.w, q2.w);
    float det = q3.x;
    float denom = dot(d, n);
    if (denom == 0.0f) return false;
    float t = div_rn(-dot(o - v0, n), denom);
    if (t < 0.0f) return false;
    if (!(t < hit.t && t > 0.00001f)) return false;
    V3 p = o + t * d;
    V3 pv = p - v0;
    V3 u0 = cross(e1, pv);
    V3 w0 = cross(pv, e2);
    if (dot(u0, n) < 0.0f || dot(w0, n) < 0.0f) return false;
    float v = div_rn(length(u0), det);
    float u = div_rn(length(w0), det);
    if (u < 0.0f || v < 0.0f || u > 1.0f || v > 1.0f) return false;
    if (u + v <= 1.0f) { hit.t = t; hit.u = u; hit.v = v; hit.prim = prim; return true; }
    return false;
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
// Control flow is "while-while": every lane first descends inner nodes until it holds a leaf (or is
// done), then the wave tests leaves together.  Visit order and pruning are those of the reference's
// recursion; only the interleaving between lanes changes (wave64 lane utilisation 32 % -> see profiles/).
// Per-lane traversal stack: the first `lds_levels` entries live in LDS (layout [level][lane] of 8-byte pairs,
// conflict-free, one ds_read_b64 / ds_write_b64 per pop / push), deeper levels spill to a global overflow buffer
// with the same coalesced layout.  LDS_ONLY (BVH depth <= the LDS levels, always the case for LDS-staged scenes)
// compiles the overflow path out, which keeps every stack access a plain ds_* instruction instead of a flat one.  Keeping only ~12 levels
// in LDS (96 B/lane) lets 8 waves/SIMD stay resident on scenes whose BVH is 20-40 levels deep.
template <bool LDS_ONLY>
struct TravStackT {
    int2* lds; int lds_stride; int lds_levels;   // lds already offset to this lane; one (code, distance bits) pair per level
    int* glob; size_t glob_stride;               // glob already offset to this lane
    RL_DEV void push(int sp, int code, float dist) const {
        if (LDS_ONLY || sp < lds_levels) lds[sp * lds_stride] = make_int2(code, __float_as_int(dist));
        else { size_t k = (size_t)(2 * (sp - lds_levels)); glob[k * glob_stride] = code; glob[(k + 1) * glob_stride] = __float_as_int(dist); }
    }
    RL_DEV void get(int sp, int* code, float* dist) const {
        if (LDS_ONLY || sp < lds_levels) { const int2 e = lds[sp * lds_stride]; *code = e.x; *dist = __int_as_float(e.y); }
        else { size_t k = (size_t)(2 * (sp - lds_levels)); *code = glob[k * glob_stride]; *dist = __int_as_float(glob[(k + 1) * glob_stride]); }
    }
};
using TravStack = TravStackT<false>;

template <class Stack>
RL_DEV int stack_pop(const Stack& st, int& sp, float t_best) {
    while (sp > 0) {
        sp--;
        int code; float dist;
        st.get(sp, &code, &dist);
        if (dist < t_best) return code;    // `if d2 < its.t` evaluated after the near subtree (accel.rs:279-284)
    }
    return RL_CHILD_NONE;
}

template <bool ANY_HIT, class Stack>
RL_DEV bool traverse(const SceneRecs& recs, int root, V3 root_lo, V3 root_hi, V3 o, V3 d, float tnear, float tfar,
                     Hit& hit, const Stack& st) {
    V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float dummy;
    int cur = root;
    if (!slab(root_lo, root_hi, o, inv_d, tnear, tfar, &dummy)) cur = RL_CHILD_NONE;   // accel.rs:293-295 / 338-340
    int sp = 0;
    bool found = false;
    while (cur != RL_CHILD_NONE) {
        // ---- phase 1: inner nodes
        while (cur >= 0) {
            hit.steps++;
            const float4* q = recs.nodes + 4 * cur;
            float4 a = q[0], b = q[1], c = q[2], e = q[3];
            V3 llo = mk3(a.x, a.y, a.z), lhi = mk3(a.w, b.x, b.y);
            V3 rlo = mk3(b.z, b.w, c.x), rhi = mk3(c.y, c.z, c.w);
            int id1 = __float_as_int(e.x), id2 = __float_as_int(e.y);
            float d1, d2;
            if (!slab(llo, lhi, o, inv_d, tnear, tfar, &d1)) d1 = f32_inf();
            if (!slab(rlo, rhi, o, inv_d, tnear, tfar, &d2)) d2 = f32_inf();
            if (d1 > d2) { float s = d1; d1 = d2; d2 = s; int si = id1; id1 = id2; id2 = si; }
            if (d1 < hit.t) {
                if (d2 < hit.t) { st.push(sp, id2, d2); sp++; }   // may still be pruned by a closer hit: re-checked at pop time
                cur = id1;
            } else cur = stack_pop(st, sp, hit.t);
        }
        // ---- phase 2: a leaf (<= 2 triangles, tested in order: accel.rs:245-254) or nothing left
        if (cur != RL_CHILD_NONE) {
            unsigned int code = (unsigned int)(~cur);
            int first = (int)(code >> 2), count = (int)(code & 3u);
            for (int k = 0; k < count; k++) {
                hit.tris++;
                const float4* q = recs.tris + 4 * (first + k);
                if (tri_test(q[0], q[1], q[2], q[3], o, d, hit, first + k)) {
                    found = true;
                    if (ANY_HIT) return true;
                }
            }
            cur = stack_pop(st, sp, hit.t);
        }
    }
    return found;
}

// Stage the node / triangle records into LDS (cooperatively, 16 bytes per lane per step).
RL_DEV void stage_scene_lds(const DeviceScene& sc, float4* lds_nodes, float4* lds_tris) {
    const float4* gn = reinterpret_cast<const float4*>(sc.nodes);
    const float4* gt = reinterpret_cast<const float4*>(sc.tris);
    for (unsigned int i = threadIdx.x; i < 4u * sc.n_nodes; i += blockDim.x) lds_nodes[i] = gn[i];
    for (unsigned int i = threadIdx.x; i < 4u * sc.n_prims; i += blockDim.x) lds_tris[i] = gt[i];
    __syncthreads();
}

}  // namespace rl
