// fused.hip.h — k_path_fused, the persistent form of the pipeline, and its launcher; instantiated by fused_lds.hip (scene staged in LDS)
// and fused_stream.hip (BVH streamed from L2 / HBM) so that the two families compile in parallel, and once more each by fused_*_fast.hip with
// RL_FAST_MATH (NUM = 1: the opt-in tolerance build; the template parameter only keeps the kernel symbols of the two builds apart).
#pragma once
#ifdef RL_STAGE_TIMERS
#include <algorithm>
#include <cstdlib>
#include <vector>
#endif

#ifndef RL_RELOAD_SCENE
#define RL_RELOAD_SCENE 1    // persistent loop re-reads scene / render constants from the kernarg segment per iteration (see k_path_fused)
#endif
#ifndef RL_FUSED_QUEUE
#define RL_FUSED_QUEUE 0    // 1 (fusedq_lds.hip / fusedq_stream.hip): this translation unit instantiates the queue-fed form of the kernel
#endif
#ifndef RL_FUSED_COLD_SCRATCH
#define RL_FUSED_COLD_SCRATCH 0   // 1 (experiment, round 6, measured: profiles/NEGATIVES.md): scenes that stream their BVH keep the per-sample-cold path state in registers / scratch instead of LDS (the freed LDS takes more stack levels: -DRL_LDS_LEVELS_STREAMING)
#endif
#ifndef RL_SHADE_NOINLINE
#define RL_SHADE_NOINLINE 0       // 1 (experiment, round 6, measured): the run-time-switch shading of scenes that stream their BVH as an out-of-line function, its live-in set passed by value
#endif
#ifndef RL_COOP_FETCH
#define RL_COOP_FETCH 0     // 1 / 2: streaming scenes fetch BVH records wave-cooperatively (trace.hip.h: traverse_coop; 1 = LDS staging, 2 = registers + ds_bpermute) — both measured slower, kept for the record
#endif

namespace rl {

// ------------------------------------------------------------------------------------------
// k_path_fused<MAT, MEDIUM, LDS> — the persistent form of the pipeline for scenes with one BSDF type: one
// launch, one lane per pixel item, the four stage functions above run back-to-back per iteration
// (raygen -> extend -> shade -> shadow) with the whole path state in registers (RegState) and the scene +
// traversal stacks in LDS.  Same functions, same order of operations, same results as the wavefront kernels;
// what disappears is ~1.4 KB/sample of state traffic through HBM and ~2000 kernel boundaries per render.
#ifdef RL_STAGE_TIMERS
__device__ unsigned long long g_stage_timers[32];      // [0..3] cycles per stage, [4..7] live lanes per stage, [8] lane slots, [16..] shadow-stage occupancy (dump_stage_timers_impl)
#endif
// RL_SHADE_NOINLINE: shade_slot out of line — the path state and the counters go in and come back by value, so that the traversal loops' live set need not include the material switch's
template <class PS> struct ShadeIO { PS ps; unsigned nv, nd, ns, ne; };
template <int MAT, bool MEDIUM, int LIGHTS, class PS>
__device__ __attribute__((noinline)) ShadeIO<PS> shade_outlined(const RenderConst* rc, const DeviceScene* sc, ShadeIO<PS> io) {
    shade_slot<MAT, MEDIUM, LIGHTS>(*rc, *sc, io.ps, io.ps.u(U_FLAGS), io.nv, io.nd, io.ns, io.ne);
    return io;
}
// QUEUE: the form that takes its work from the chain pass's completion queue (the evaluation pass of reference-order streams, launched beside the chain pass): an
// instantiation of its own (fusedq_lds.hip / fusedq_stream.hip), so that the per-sample kernel's code is exactly what it is without it
template <int MAT, bool MEDIUM, bool LDS_SCENE, int LIGHTS, int NUM, bool QUEUE = false>
__global__ void __launch_bounds__(256, LDS_SCENE ? RL_FUSED_WAVES : RL_FUSED_WAVES_STREAMING) k_path_fused(RenderConst rc_arg, DeviceScene sc_arg, StackConf stc) {
    const RenderConst& rc = rc_arg;
    const DeviceScene& sc = sc_arg;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    float4* after_scene = smem;
    constexpr bool LDS2 = LDS_SCENE && RL_LDS_TWO_LEVEL;      // LDS-staged scenes: two-level node records (trace.hip.h: traverse2, stage_scene_lds2)
    if (LDS2) {
        stage_scene_lds2(sc, smem, smem + lds_nodes2_float4s(sc.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes2_float4s(sc.n_nodes);
        after_scene = smem + lds_scene2_float4s(sc.n_nodes, sc.n_prims);
    } else if (LDS_SCENE) {
        stage_scene_lds(sc, smem, smem + lds_nodes_float4s(sc.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes_float4s(sc.n_nodes);
        after_scene = smem + lds_scene_float4s(sc.n_nodes, sc.n_prims);
    } else {
        recs.nodes = streamed_nodes<TravStackT<false>>(sc);   // exact build: two-level records; tolerance build: quantised BVH4 nodes
        recs.tris = reinterpret_cast<const float4*>(sc.tris);
    }
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    // LDS: [scene][cold path state (u64 | f32 | u32 planes)][per-lane stacks]
    unsigned long long* cold_q = reinterpret_cast<unsigned long long*>(after_scene);
    float* cold_f = reinterpret_cast<float*>(cold_q + 256 * FusedState::kColdQ);
    unsigned* cold_u = reinterpret_cast<unsigned*>(cold_f + 256 * FusedState::kColdF);
    using StackT = typename std::conditional<LDS2, TravStackLds2, TravStackT<LDS_SCENE>>::type;
    constexpr bool COLD_LDS = LDS_SCENE || !RL_FUSED_COLD_SCRATCH;
    const StackT stack(make_stack<LDS_SCENE>(stc, COLD_LDS ? cold_u + 256 * FusedState::kColdU : reinterpret_cast<unsigned*>(after_scene), tid));
    // streaming scenes: per-wave staging area of the cooperative record fetch, after the stacks
    constexpr bool COOP = !LDS_SCENE && RL_COOP_FETCH;
    float4* stage = reinterpret_cast<float4*>(cold_u + 256 * FusedState::kColdU + 2 * 256 * stc.lds_levels) + (threadIdx.x >> 6) * kCoopStageFloat4s;
    typename std::conditional<COLD_LDS, FusedState, RegState>::type ps;
    if constexpr (COLD_LDS) { ps.cold_q = cold_q + threadIdx.x; ps.cold_f = cold_f + threadIdx.x; ps.cold_u = cold_u + threadIdx.x; }
#pragma unroll
    for (int i = 0; i < F_COUNT; i++) ps.fv[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < U_COUNT; i++) ps.uv[i] = 0u;
#pragma unroll
    for (int i = 0; i < Q_COUNT; i++) ps.qv[i] = 0ull;
    storec(ps, F_AR, czero());
    PU(U_CURSOR) = 0u; PU(U_SAMPLE) = 0u;
    // few work items (reference-order streams: one per 16x16 block, 8160 at 1080p): they are dealt to every 2^item_shift-th lane, so that
    // they spread over as many waves as the chip holds — a wave's speed does not depend on how many of its lanes are live, and a SIMD
    // needs 2-3 ready waves to issue at its rate (profiles/r02_valu_calibration.json)
    const unsigned item0 = (tid & ((1u << rc.item_shift) - 1u)) == 0u ? (tid >> rc.item_shift) : 0xffffffffu;
    PU(U_ITEM) = item0;
    PU(U_PRIM) = 0xffffffffu;
    PU(U_FLAGS) = item0 < rc.n_items ? (ST_REGEN | ST_FRESH) : ST_FINISHED;
    unsigned n_samples = 0, n_draws = 0, n_vertices = 0, n_shadow = 0, n_ext = 0;
#ifdef RL_STAGE_TIMERS
    unsigned long long tm[4] = {0, 0, 0, 0}, ln[5] = {0, 0, 0, 0, 0};
    // the shadow stage where it runs: wave-iterations by the number of lanes that hold a shadow ray (0 | 1-16 | 17-32 | 33-48 | 49-64), and what packing the shadow rays of
    // TWO consecutive iterations into one traversal could save: pairs of iterations by (both empty | one empty | both hold rays and together <= 64 | together > 64)
    unsigned long long sh_hist[5] = {0, 0, 0, 0, 0}, sh_pair[4] = {0, 0, 0, 0}, sh_cyc_exec = 0; unsigned sh_prev = 0, sh_parity = 0;
#define RL_T0 { t0 = __builtin_readcyclecounter(); }
#define RL_T1(K, COND) { unsigned long long t1 = __builtin_readcyclecounter(); tm[K] += t1 - t0; ln[K] += __popcll(__ballot(COND)); t0 = t1; }
    unsigned long long t0;
#else
#define RL_T0
#define RL_T1(K, COND)
#endif
    if constexpr (COOP) {
        // every lane of the wave stays in the loop until the whole wave has no work left: idle lanes are the loaders of the cooperative fetch
        while (__ballot(!(PU(U_FLAGS) & ST_FINISHED)) != 0ull) {
            RL_T0
#ifdef RL_STAGE_TIMERS
            ln[4] += 64;
            const bool c0 = PU(U_FLAGS) & ST_REGEN;
#endif
            if (PU(U_FLAGS) & ST_REGEN) raygen_slot<true>(rc, sc, ps, n_samples, n_draws);
            RL_T1(0, c0)
            const bool has_ray = (PU(U_FLAGS) & ST_RAY) != 0u;
            extend_slot_coop(sc, recs, stack, ps, has_ray, stage);
            RL_T1(1, has_ray)
            if (has_ray) shade_slot<MAT, MEDIUM, LIGHTS>(rc, sc, ps, PU(U_FLAGS), n_vertices, n_draws, n_shadow, n_ext);
            RL_T1(2, has_ray)
            const bool has_shadow = (PU(U_FLAGS) & ST_SHADOW) != 0u;
            shadow_slot_coop(sc, recs, stack, ps, has_shadow, stage);
            RL_T1(3, has_shadow)
        }
    } else
    {
    // ---- QUEUE form (the evaluation pass of reference-order streams beside the chain pass): the lanes take their pixel items from this launch's block list — item
    // position c = block c / (256 split) of the list, item c % (256 split) of that block — through one claim counter, like the dispenser's lanes; every block of the list is
    // complete (the host only lists blocks the chain kernel has flagged), so nothing here waits.
    constexpr bool qmode = QUEUE;
    constexpr unsigned kQDone = 0xffffffffu;
    if constexpr (qmode) {
        PU(U_ITEM) = 0u;
        PU(U_FLAGS) = ST_FINISHED;           // "needs a claim"
    }
    for (;;) {
    if constexpr (qmode) {
        // (the render constants re-read from the kernarg segment, like the loop body does: kept in scalar registers across the loop they cost the medium kernel 82 spilled SGPRs)
        const char __attribute__((address_space(4)))* kq = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kq));
        const RenderConst& rc_arg = *(const RenderConst*)kq;
        const unsigned ipb = 256u * rc_arg.split, total = rc_arg.q_n * ipb;
        unsigned fl = PU(U_FLAGS);
        for (int tries = 0; tries < 4; tries++) {       // (a claim can land past the end of a ragged block: claim again, a few times per trip)
            if ((fl & ST_FINISHED) && PU(U_ITEM) != kQDone) {
                const unsigned c = atomicAdd(rc_arg.q_ctr, 1u);
                if (c >= total) PU(U_ITEM) = kQDone;
                else {
                    const unsigned j = rc_arg.q_list[c / ipb], idx = c - (c / ipb) * ipb;
                    unsigned bx_, by_, bw_, bh_;
                    block_geometry(rc_arg, rc_arg.owned_blocks[j], &bx_, &by_, &bw_, &bh_);
                    const unsigned npx = min(rc_arg.cursor_end, bw_ * bh_) - min(rc_arg.cursor_begin, bw_ * bh_);
                    if (idx < npx * rc_arg.split) {
                        PU(U_ITEM) = rc_arg.block_item_base[j] * rc_arg.split + idx;
                        PU(U_PRIM) = 0xffffffffu; PU(U_CURSOR) = 0u; PU(U_SAMPLE) = 0u;
                        storec(ps, F_AR, czero());
                        fl = ST_REGEN | ST_FRESH;
                    }                                   // else: past the block's last pixel, claim again
                }
            }
            if (__ballot((fl & ST_FINISHED) && PU(U_ITEM) != kQDone) == 0ull) break;
        }
        PU(U_FLAGS) = fl;
        if (__ballot(!(fl & ST_FINISHED)) == 0ull) {
            if (__ballot((fl & ST_FINISHED) && PU(U_ITEM) != kQDone) == 0ull) break;      // every lane of the wave is done
            continue;                                                                      // (only claims past ragged ends so far: claim on)
        }
    }
    // (queue mode: ONE trip of the loop body, then back to the claims — a lane whose pixel is done takes its next item while the others go on, like the dispenser's lanes)
    if (!(PU(U_FLAGS) & ST_FINISHED)) do {
#if RL_RELOAD_SCENE
        // The scene record (25 pointers, camera matrices, ...) and the render constants do not fit the scalar registers next to the saved
        // exec masks of the stage functions: kept live across the loop they are spilled to VGPR lanes (v_writelane / v_readlane were ~600 of
        // the kernel's ~4500 vector instructions, 126 spilled SGPRs).  Re-deriving their address from the kernarg segment once per iteration
        // lets the compiler s_load what each stage needs instead (0-22 spilled SGPRs, 2-6 spilled VGPRs instead of 21; cbox 55.8 -> 51.9 ms,
        // cbox + medium 136.0 -> 118.9 ms at 32 spp, same bits).  The render constants are re-read too where that paid (the medium kernels).
        const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        constexpr size_t sc_off = (sizeof(RenderConst) + alignof(DeviceScene) - 1) / alignof(DeviceScene) * alignof(DeviceScene);
        static_assert(sc_off == offsetof(PathKernargs, sc) && offsetof(PathKernargs, rc) == 0, "kernarg layout of k_path_fused(RenderConst, DeviceScene, StackConf)");
        const DeviceScene& sc = *(const DeviceScene*)(ka + sc_off);
        const RenderConst& rc = (MEDIUM || RL_RELOAD_SCENE > 1) ? *(const RenderConst*)ka : rc_arg;
#endif
        RL_T0
#ifdef RL_STAGE_TIMERS
        ln[4] += 64;
        const bool c0 = PU(U_FLAGS) & ST_REGEN;
#endif
        if (PU(U_FLAGS) & ST_REGEN) raygen_slot<true>(rc, sc, ps, n_samples, n_draws);   // work items from the global dispenser
        RL_T1(0, c0)
#ifdef RL_STAGE_TIMERS
        const bool c1 = PU(U_FLAGS) & ST_RAY;
#endif
        if (PU(U_FLAGS) & ST_RAY) {
            extend_slot(sc, recs, stack, ps);
            RL_T1(1, c1)
            if constexpr (RL_SHADE_NOINLINE && MAT == -1 && !LDS_SCENE) {
                ShadeIO<decltype(ps)> io{ps, n_vertices, n_draws, n_shadow, n_ext};
                io = shade_outlined<MAT, MEDIUM, LIGHTS>(&rc, &sc, io);
                ps = io.ps; n_vertices = io.nv; n_draws = io.nd; n_shadow = io.ns; n_ext = io.ne;
            } else
            shade_slot<MAT, MEDIUM, LIGHTS>(rc, sc, ps, PU(U_FLAGS), n_vertices, n_draws, n_shadow, n_ext);
        }
        RL_T1(2, c1)
#ifdef RL_STAGE_TIMERS
        const bool c3 = PU(U_FLAGS) & ST_SHADOW;
#endif
        if (PU(U_FLAGS) & ST_SHADOW) shadow_slot(sc, recs, stack, ps);
#ifdef RL_STAGE_TIMERS
        { const unsigned nsh = (unsigned)__popcll(__ballot(c3)); sh_hist[nsh == 0u ? 0 : 1 + (nsh - 1u) / 16u]++;
          if (nsh) sh_cyc_exec += __builtin_readcyclecounter() - t0;
          if (sh_parity) { const unsigned a = sh_prev, b = nsh; sh_pair[(a == 0u && b == 0u) ? 0 : ((a == 0u || b == 0u) ? 1 : (a + b <= 64u ? 2 : 3))]++; }
          sh_prev = nsh; sh_parity ^= 1u; }
#endif
        RL_T1(3, c3)
    } while (!qmode && !(PU(U_FLAGS) & ST_FINISHED));
    if (!qmode) break;
    }
    }
#ifdef RL_STAGE_TIMERS
    if ((threadIdx.x & 63u) == 0u) { for (int k = 0; k < 4; k++) { atomicAdd(&g_stage_timers[k], tm[k]); atomicAdd(&g_stage_timers[4 + k], ln[k]); } atomicAdd(&g_stage_timers[8], ln[4]);
        for (int k = 0; k < 5; k++) atomicAdd(&g_stage_timers[16 + k], sh_hist[k]); for (int k = 0; k < 4; k++) atomicAdd(&g_stage_timers[21 + k], sh_pair[k]); atomicAdd(&g_stage_timers[25], sh_cyc_exec); }
#endif
    {
        const int which[5] = {STAT_SAMPLES, STAT_VERTICES, STAT_DRAWS, STAT_SHADOW_RAYS, STAT_EXT_RAYS};
        const unsigned vals[5] = {n_samples, n_vertices, n_draws, n_shadow, n_ext};
        block_stats<5>(rc.partials, which, vals);
    }
}


template <bool LDS_SCENE, int MAT>
static void launch_fused_mat(bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    if (medium) { if (area_only) hipLaunchKernelGGL((k_path_fused<MAT, true, LDS_SCENE, LIGHTS_AREA_ONLY, RL_NUMERICS_ID, RL_FUSED_QUEUE != 0>), grid, block, lds_bytes, st, rc, ds, stc);
                  else hipLaunchKernelGGL((k_path_fused<MAT, true, LDS_SCENE, LIGHTS_ANY, RL_NUMERICS_ID, RL_FUSED_QUEUE != 0>), grid, block, lds_bytes, st, rc, ds, stc); }
    else { if (area_only) hipLaunchKernelGGL((k_path_fused<MAT, false, LDS_SCENE, LIGHTS_AREA_ONLY, RL_NUMERICS_ID, RL_FUSED_QUEUE != 0>), grid, block, lds_bytes, st, rc, ds, stc);
           else hipLaunchKernelGGL((k_path_fused<MAT, false, LDS_SCENE, LIGHTS_ANY, RL_NUMERICS_ID, RL_FUSED_QUEUE != 0>), grid, block, lds_bytes, st, rc, ds, stc); }
}
template <bool LDS_SCENE>
static void launch_fused_impl(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    switch (mat) {
        case BSDF_DIFFUSE: launch_fused_mat<LDS_SCENE, BSDF_DIFFUSE>(medium, area_only, grid, block, lds_bytes, st, rc, ds, stc); break;
        case BSDF_PHONG: launch_fused_mat<LDS_SCENE, BSDF_PHONG>(medium, area_only, grid, block, lds_bytes, st, rc, ds, stc); break;
        case BSDF_METAL: launch_fused_mat<LDS_SCENE, BSDF_METAL>(medium, area_only, grid, block, lds_bytes, st, rc, ds, stc); break;
        case BSDF_GLASS: launch_fused_mat<LDS_SCENE, BSDF_GLASS>(medium, area_only, grid, block, lds_bytes, st, rc, ds, stc); break;
        case -1: launch_fused_mat<LDS_SCENE, -1>(medium, area_only, grid, block, lds_bytes, st, rc, ds, stc); break;      // several BSDF types: run-time switch per vertex
        default: launch_fused_mat<LDS_SCENE, BSDF_SUBSTRATE>(medium, area_only, grid, block, lds_bytes, st, rc, ds, stc); break;
    }
}
template <bool LDS_SCENE>
static void dump_stage_timers_impl() {
#ifdef RL_STAGE_TIMERS
    // dev-only build: per-stage cycle shares and active-lane fractions of the fused loop
    unsigned long long h[32];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage_timers), sizeof(h));
    const double tot = (double)(h[0] + h[1] + h[2] + h[3]);
    const char* names[4] = {"raygen", "extend", "shade", "shadow"};
    for (int k = 0; k < 4; k++) std::fprintf(stderr, "[stage] %-7s cycles %5.1f %%  lanes %5.1f %%\n", names[k], 100.0 * h[k] / tot, 100.0 * h[4 + k] / (double)h[8]);
    {
        const double it = (double)(h[16] + h[17] + h[18] + h[19] + h[20]), ex = it - (double)h[16], pr = (double)(h[21] + h[22] + h[23] + h[24]);
        if (it > 0) std::fprintf(stderr, "[stage] shadow stage: runs in %.1f %% of the wave-iterations; where it runs the wave holds 1-16 / 17-32 / 33-48 / 49-64 shadow rays in %.1f / %.1f / %.1f / %.1f %% (%.1f lanes on average)\n",
                                 100.0 * ex / it, 100.0 * h[17] / std::max(1.0, ex), 100.0 * h[18] / std::max(1.0, ex), 100.0 * h[19] / std::max(1.0, ex), 100.0 * h[20] / std::max(1.0, ex), (double)h[7] / std::max(1.0, ex));
        if (pr > 0) std::fprintf(stderr, "[stage] pairs of consecutive iterations: neither holds shadow rays %.1f %%, one does %.1f %%, both and <= 64 together %.1f %% (one traversal could serve both), both and > 64 %.1f %%\n",
                                 100.0 * h[21] / pr, 100.0 * h[22] / pr, 100.0 * h[23] / pr, 100.0 * h[24] / pr);
    }
    std::memset(h, 0, sizeof(h)); hipMemcpyToSymbol(HIP_SYMBOL(g_stage_timers), h, sizeof(h));
#endif
}

}  // namespace rl
