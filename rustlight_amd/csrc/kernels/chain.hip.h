// chain.hip.h — k_stream_chain: the first pass of RL_STREAM_REFERENCE_ORDER in the persistent pipeline, and its launcher; instantiated by
// chain_lds.hip / chain_stream.hip (and once more each with RL_FAST_MATH by chain_*_fast.hip, so that the tolerance build decides path lengths
// with the arithmetic its second pass uses).
//
// rustlight's compute_mc (src/integrators/mod.rs:420-435) hands ONE sampler to a 16x16 block and consumes it over (iy, ix, sample): where the
// stream stands when sample k+1 begins depends on how many numbers sample k took, so a block is a serial chain of 256 x spp samples and an
// image has only ceil(W/16) x ceil(H/16) of them (8160 at 1080p) for a chip that wants 262 144 lanes busy.  What is serial, though, is only the
// COUNT of draws — and that is decided by the extension rays alone (hit or miss, medium distance, BSDF / phase sample, Russian roulette; never
// by a shadow ray, a light sample's value, an emission or a MIS weight).  So the mode runs in two passes:
//   1. k_stream_chain (here): one lane per block walks the chain with the radiance half of the integrator left out (shade_slot<.., DRAWS_ONLY>,
//      no NEE evaluation, no shadow traversal) and records the sampler state at the start of every camera sample — 32 B per sample in HBM;
//   2. k_path_fused with stream_mode = kStreamGivenStates: the per-sample form of the kernel, every lane of the chip busy, each camera sample
//      started from its recorded state and folded into its pixel in sample order.
// The image and the counters are those of the single-pass walk, bit for bit (tests/test_gpu_parity.py: reference-order cases run both forms).
#pragma once
#ifdef RL_STAGE_TIMERS
#include <algorithm>
#include <cstdlib>
#include <vector>
#endif

#ifndef RL_CHAIN_WAVES
#define RL_CHAIN_WAVES 6             // waves per SIMD asked for on LDS-staged scenes (73 VGPRs; 4 / 6 / 8: 912 / 867 / 907 ms for the chains of cbox 1080p x 128 spp)
#endif
#ifndef RL_CHAIN_WAVES_STREAMING
#define RL_CHAIN_WAVES_STREAMING 6
#endif

namespace rl {

#ifdef RL_STAGE_TIMERS
static __device__ unsigned long long g_chain_timers[8];      // dev-only (one copy per translation unit: read by that unit's own dump function): cycles in raygen / extend / shade, wave-iterations, lane-iterations of each
static constexpr unsigned kChainWaveSlots = 1u << 16;
static __device__ unsigned long long g_chain_waves[4 * kChainWaveSlots];      // dev-only: per wave (start, end) on the 100 MHz wall clock and its iteration count — the lifetime histogram of dump_chain_timers
#define RL_CT0 { ct0 = __builtin_readcyclecounter(); }
#define RL_CT1(K, COND) { const unsigned long long t1 = __builtin_readcyclecounter(); ctm[K] += t1 - ct0; cln[K] += __popcll(__ballot(COND)); ct0 = t1; }
#else
#define RL_CT0
#define RL_CT1(K, COND)
#endif

template <int MAT, bool MEDIUM, bool LDS_SCENE, int NUM>
__global__ void __launch_bounds__(256, LDS_SCENE ? RL_CHAIN_WAVES : RL_CHAIN_WAVES_STREAMING) k_stream_chain(RenderConst rc_arg, DeviceScene sc_arg, StackConf stc) {
    const DeviceScene& sc0 = sc_arg;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    float4* after_scene = smem;
    if (LDS_SCENE) {
        stage_scene_lds(sc0, smem, smem + lds_nodes_float4s(sc0.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes_float4s(sc0.n_nodes);
        after_scene = smem + lds_scene_float4s(sc0.n_nodes, sc0.n_prims);
    } else {
        recs.nodes = streamed_nodes<TravStackT<false>>(sc0);   // exact build: two-level records; tolerance build: quantised BVH4 nodes
        recs.tris = reinterpret_cast<const float4*>(sc0.tris);
    }
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (rc_arg.queue && threadIdx.x == 0u) queue_workgroup_started(rc_arg);      // (the host launches the evaluation pass beside this kernel once EVERY workgroup of it runs)
    // LDS: [scene][per-lane stacks]; the whole chain state lives in registers (no per-sample-cold radiance state to park)
    const TravStackT<LDS_SCENE> stack = make_stack<LDS_SCENE>(stc, reinterpret_cast<unsigned*>(after_scene), tid);
    RegState ps;
#pragma unroll
    for (int i = 0; i < F_COUNT; i++) ps.fv[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < U_COUNT; i++) ps.uv[i] = 0u;
#pragma unroll
    for (int i = 0; i < Q_COUNT; i++) ps.qv[i] = 0ull;
    // the chains are dealt to every 2^item_shift-th lane, so that they spread over as many waves as the chip holds (see k_path_fused)
    const unsigned item0 = (tid & ((1u << rc_arg.item_shift) - 1u)) == 0u ? (tid >> rc_arg.item_shift) : 0xffffffffu;
    PU(U_ITEM) = item0;
    PU(U_PRIM) = 0xffffffffu;
    PU(U_FLAGS) = item0 < rc_arg.n_items ? (ST_REGEN | ST_FRESH) : ST_FINISHED;
    unsigned dummy = 0;
#ifdef RL_STAGE_TIMERS
    unsigned long long ctm[3] = {0, 0, 0}, cln[3] = {0, 0, 0}, ct0, n_it = 0;
    const unsigned long long wave_t0 = wall_clock64();
#endif
    // scene record and render constants re-read from the kernarg segment once per iteration (see k_path_fused: SGPR pressure)
#define RL_CHAIN_KERNARGS \
        const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr(); \
        asm volatile("" : "+s"(ka)); \
        constexpr size_t sc_off = (sizeof(RenderConst) + alignof(DeviceScene) - 1) / alignof(DeviceScene) * alignof(DeviceScene); \
        static_assert(sc_off == offsetof(PathKernargs, sc), "kernarg layout of k_stream_chain"); \
        const DeviceScene& sc = *(const DeviceScene*)(ka + sc_off); \
        const RenderConst& rc = *(const RenderConst*)ka;
    if (LDS_SCENE && stc.pre_group > 0) {
        // Tiny scenes: the idle lanes of a chain's group evaluate its ray against every node and triangle at once, the chain lane walks the records
        // (trace.hip.h: precompute_records / traverse_pre).  Every lane of the wave stays in the loop until the wave's chains are done.
        const unsigned group = (unsigned)stc.pre_group, lane = threadIdx.x & 63u, sub = lane & (group - 1u), lead = lane & ~(group - 1u);
        const unsigned slot = threadIdx.x / group;
        float4* pre_nodes = reinterpret_cast<float4*>(reinterpret_cast<unsigned*>(after_scene) + 2 * 256 * stc.lds_levels) + (size_t)slot * (2u * sc0.n_nodes);
        float2* pre_tris = reinterpret_cast<float2*>(reinterpret_cast<float4*>(reinterpret_cast<unsigned*>(after_scene) + 2 * 256 * stc.lds_levels) + (size_t)(256u / group) * (2u * sc0.n_nodes)) + (size_t)slot * sc0.n_prims;
        while (__ballot(!(PU(U_FLAGS) & ST_FINISHED)) != 0ull) {
            RL_CHAIN_KERNARGS
#ifdef RL_STAGE_TIMERS
            n_it++;
            const bool c0 = PU(U_FLAGS) & ST_REGEN;
#endif
            RL_CT0
            if (PU(U_FLAGS) & ST_REGEN) raygen_chain_slot(rc, sc, ps);
            RL_CT1(0, c0)
            const unsigned flags = PU(U_FLAGS);
            const bool has_ray = (flags & ST_RAY) != 0u;
            if (__ballot(has_ray) != 0ull) {
                const bool primary = ((flags >> ST_PREV_SHIFT) & 3u) == PREV_SENSOR;
                const V3 o = primary ? mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]) : load3(ps, F_OX);
                const V3 d = load3(ps, F_DX);
                const V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
                // the group's ray, as its chain lane holds it
                const bool g_ray = __shfl((int)has_ray, (int)lead, 64) != 0;
                const V3 go = mk3(__shfl(o.x, (int)lead, 64), __shfl(o.y, (int)lead, 64), __shfl(o.z, (int)lead, 64));
                const V3 gd = mk3(__shfl(d.x, (int)lead, 64), __shfl(d.y, (int)lead, 64), __shfl(d.z, (int)lead, 64));
                const V3 gi = mk3(__shfl(inv_d.x, (int)lead, 64), __shfl(inv_d.y, (int)lead, 64), __shfl(inv_d.z, (int)lead, 64));
                precompute_records(recs, sc0.n_nodes, sc0.n_prims, go, gd, gi, kEps, g_ray, sub, group, pre_nodes, pre_tris);
                if (has_ray) {
                    Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
                    traverse_pre(pre_nodes, pre_tris, recs, sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                                 o, d, inv_d, kEps, kF32Max, hit, stack);
                    PF(F_T) = hit.t; PF(F_U) = hit.u; PF(F_V) = hit.v;
                    PU(U_PRIM) = (unsigned)hit.prim;
                }
                RL_CT1(1, has_ray)
                if (has_ray) shade_slot<MAT, MEDIUM, LIGHTS_AREA_ONLY, true>(rc, sc, ps, flags, dummy, dummy, dummy, dummy);
                RL_CT1(2, has_ray)
            }
        }
    } else if (!LDS_SCENE && !TravStackT<false>::kBvh4 && stc.pre_group > 0) {
        // Scenes that stream their BVH (exact build): the lanes of a chain's group fetch whole 16-node blocks of the treelet-ordered node array for it
        // (trace.hip.h: traverse_treelet).  Every lane of the wave stays in the loop until the wave's chains are done.
        const unsigned group = (unsigned)stc.pre_group, lane = threadIdx.x & 63u, sub = lane & (group - 1u), lead = lane & ~(group - 1u);
        const unsigned slot = threadIdx.x / group;
        // LDS per chain: 64 float4 of nodes + 8 of triangles + the whole traversal stack (stack_depth entries of 8 bytes, rounded to float4s)
        const unsigned per_chain = 72u + (sc0.stack_depth + 1u) / 2u;
        float4* cache_nodes = reinterpret_cast<float4*>(reinterpret_cast<unsigned*>(after_scene) + 2 * 256 * stc.lds_levels) + (size_t)slot * per_chain;
        float4* cache_tris = cache_nodes + 64;
        ChainStack cstack; cstack.base = reinterpret_cast<int2*>(cache_nodes + 72);
        while (__ballot(!(PU(U_FLAGS) & ST_FINISHED)) != 0ull) {
            RL_CHAIN_KERNARGS
            if (PU(U_FLAGS) & ST_REGEN) raygen_chain_slot(rc, sc, ps);
            const unsigned flags = PU(U_FLAGS);
            const bool has_ray = (flags & ST_RAY) != 0u;
            if (__ballot(has_ray) != 0ull) {
                const bool primary = ((flags >> ST_PREV_SHIFT) & 3u) == PREV_SENSOR;
                const V3 o = primary ? mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]) : load3(ps, F_OX);
                const V3 d = load3(ps, F_DX);
                Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
                traverse_treelet(reinterpret_cast<const float4*>(sc0.nodes_t), recs.tris, sc0.root_t, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                                 o, d, kEps, kF32Max, has_ray, lead, sub, group, cache_nodes, cache_tris, hit, cstack);
                if (has_ray) {
                    PF(F_T) = hit.t; PF(F_U) = hit.u; PF(F_V) = hit.v;
                    PU(U_PRIM) = (unsigned)hit.prim;
#ifdef RL_STAGE_TIMERS
                    atomicAdd(&g_chain_timers[7], ((unsigned long long)hit.fetches << 32) | (unsigned long long)hit.steps);      // dev-only: block fetches | node steps
#endif
                    shade_slot<MAT, MEDIUM, LIGHTS_AREA_ONLY, true>(rc, sc, ps, flags, dummy, dummy, dummy, dummy);
                }
            }
        }
    } else
    while (!(PU(U_FLAGS) & ST_FINISHED)) {
        RL_CHAIN_KERNARGS
#ifdef RL_STAGE_TIMERS
        n_it++;
        const bool c0 = PU(U_FLAGS) & ST_REGEN;
#endif
        RL_CT0
        if (PU(U_FLAGS) & ST_REGEN) raygen_chain_slot(rc, sc, ps);
        RL_CT1(0, c0)
#ifdef RL_STAGE_TIMERS
        const bool c1 = PU(U_FLAGS) & ST_RAY;
#endif
        if (PU(U_FLAGS) & ST_RAY) {
            extend_slot(sc, recs, stack, ps);
            RL_CT1(1, c1)
            shade_slot<MAT, MEDIUM, LIGHTS_AREA_ONLY, true>(rc, sc, ps, PU(U_FLAGS), dummy, dummy, dummy, dummy);
            RL_CT1(2, c1)
        }
    }
#undef RL_CHAIN_KERNARGS
#ifdef RL_STAGE_TIMERS
    if ((threadIdx.x & 63u) == 0u) { for (int k = 0; k < 3; k++) { atomicAdd(&g_chain_timers[k], ctm[k]); atomicAdd(&g_chain_timers[3 + k], cln[k]); } atomicAdd(&g_chain_timers[6], n_it);
        // (+ where the wave ran: HW_ID (SIMD, CU, shader array / engine) and XCC_ID — is a grid of a few hundred workgroups spread over the chip's 256 CUs?)
        const unsigned w = tid >> 6; if (w < kChainWaveSlots) { g_chain_waves[4 * w] = wave_t0; g_chain_waves[4 * w + 1] = wall_clock64(); g_chain_waves[4 * w + 2] = n_it;
            g_chain_waves[4 * w + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4); } }
#endif
}

template <bool LDS_SCENE>
static void dump_chain_timers_impl() {
#ifdef RL_STAGE_TIMERS
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_chain_timers), sizeof(h));
    const double tot = (double)(h[0] + h[1] + h[2]);
    const char* names[3] = {"raygen", "extend", "shade"};
    if (tot > 0) for (int k = 0; k < 3; k++) std::fprintf(stderr, "[chain] %-7s cycles %5.1f %%  (%.0f cycles per wave-iteration, %.2f lanes)\n", names[k], 100.0 * h[k] / tot, (double)h[k] / (double)h[6], (double)h[3 + k] / (double)h[6]);
    if (h[7]) std::fprintf(stderr, "[chain] treelet walk: %llu node steps, %llu block fetches (%.2f steps per fetch)\n", h[7] & 0xffffffffull, h[7] >> 32, (double)(h[7] & 0xffffffffull) / (double)(h[7] >> 32));
    std::memset(h, 0, sizeof(h)); hipMemcpyToSymbol(HIP_SYMBOL(g_chain_timers), h, sizeof(h));
    {   // wave lifetimes: when the waves started (generations), how long they lived, percentiles over the waves that did work
        std::vector<unsigned long long> w(4 * (size_t)kChainWaveSlots);
        hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_chain_waves), w.size() * 8);
        unsigned long long t0 = ~0ull, t1 = 0; size_t n = 0;
        for (size_t i = 0; i < kChainWaveSlots; i++) if (w[4 * i]) { t0 = std::min(t0, w[4 * i]); t1 = std::max(t1, w[4 * i + 1]); n++; }
        if (n) {
            std::vector<double> life, start; size_t heavy = 0;
            for (size_t i = 0; i < kChainWaveSlots; i++) if (w[4 * i]) { const double l = (w[4 * i + 1] - w[4 * i]) * 1e-5; if (w[4 * i + 2] > 1000) { life.push_back(l); start.push_back((w[4 * i] - t0) * 1e-5); heavy++; } }
            std::sort(life.begin(), life.end()); std::sort(start.begin(), start.end());
            auto pc = [](const std::vector<double>& v, double q) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(q * v.size()))]; };
            std::fprintf(stderr, "[chain] %zu waves, %zu with > 1000 iterations; kernel span %.1f ms; heavy waves' lifetime ms: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; their start ms: p50 %.2f p90 %.2f p99 %.2f max %.2f\n",
                         n, heavy, (t1 - t0) * 1e-5, pc(life, 0.1), pc(life, 0.5), pc(life, 0.9), pc(life, 0.99), life.empty() ? 0.0 : life.back(), pc(start, 0.5), pc(start, 0.9), pc(start, 0.99), start.empty() ? 0.0 : start.back());
            if (getenv("RL_CHAIN_WAVE_TIMES")) { FILE* f = std::fopen(getenv("RL_CHAIN_WAVE_TIMES"), "w"); if (f) { for (size_t i = 0; i < kChainWaveSlots; i++) if (w[4 * i]) std::fprintf(f, "%zu %.3f %.3f %llu %llu %llu\n", i, (w[4 * i] - t0) * 1e-5, (w[4 * i + 1] - t0) * 1e-5, w[4 * i + 2], w[4 * i + 3] & 0xffffffffull, w[4 * i + 3] >> 32); std::fclose(f); } }
        }
        std::fill(w.begin(), w.end(), 0ull); hipMemcpyToSymbol(HIP_SYMBOL(g_chain_waves), w.data(), w.size() * 8);
    }
#endif
}

template <bool LDS_SCENE, int MAT>
static void launch_chain_mat(bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    if (medium) hipLaunchKernelGGL((k_stream_chain<MAT, true, LDS_SCENE, RL_NUMERICS_ID>), grid, block, lds_bytes, st, rc, ds, stc);
    else hipLaunchKernelGGL((k_stream_chain<MAT, false, LDS_SCENE, RL_NUMERICS_ID>), grid, block, lds_bytes, st, rc, ds, stc);
}
template <bool LDS_SCENE>
static void launch_chain_impl(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    switch (mat) {
        case BSDF_DIFFUSE: launch_chain_mat<LDS_SCENE, BSDF_DIFFUSE>(medium, grid, block, lds_bytes, st, rc, ds, stc); break;
        case BSDF_PHONG: launch_chain_mat<LDS_SCENE, BSDF_PHONG>(medium, grid, block, lds_bytes, st, rc, ds, stc); break;
        case BSDF_METAL: launch_chain_mat<LDS_SCENE, BSDF_METAL>(medium, grid, block, lds_bytes, st, rc, ds, stc); break;
        case BSDF_GLASS: launch_chain_mat<LDS_SCENE, BSDF_GLASS>(medium, grid, block, lds_bytes, st, rc, ds, stc); break;
        case -1: launch_chain_mat<LDS_SCENE, -1>(medium, grid, block, lds_bytes, st, rc, ds, stc); break;
        default: launch_chain_mat<LDS_SCENE, BSDF_SUBSTRATE>(medium, grid, block, lds_bytes, st, rc, ds, stc); break;
    }
}

}  // namespace rl
