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

#ifndef RL_CHAIN_WAVES
#define RL_CHAIN_WAVES 6             // waves per SIMD asked for on LDS-staged scenes (73 VGPRs; 4 / 6 / 8: 912 / 867 / 907 ms for the chains of cbox 1080p x 128 spp)
#endif
#ifndef RL_CHAIN_WAVES_STREAMING
#define RL_CHAIN_WAVES_STREAMING 6
#endif

namespace rl {

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
        recs.nodes = TravStackT<false>::kBvh4 ? reinterpret_cast<const float4*>(sc0.nodes4) : reinterpret_cast<const float4*>(sc0.nodes);   // tolerance build: quantised BVH4 nodes
        recs.tris = reinterpret_cast<const float4*>(sc0.tris);
    }
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
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
    while (!(PU(U_FLAGS) & ST_FINISHED)) {
        // scene record and render constants re-read from the kernarg segment once per iteration (see k_path_fused: SGPR pressure)
        const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        constexpr size_t sc_off = (sizeof(RenderConst) + alignof(DeviceScene) - 1) / alignof(DeviceScene) * alignof(DeviceScene);
        static_assert(sc_off == offsetof(PathKernargs, sc), "kernarg layout of k_stream_chain");
        const DeviceScene& sc = *(const DeviceScene*)(ka + sc_off);
        const RenderConst& rc = *(const RenderConst*)ka;
        if (PU(U_FLAGS) & ST_REGEN) raygen_chain_slot(rc, sc, ps);
        if (PU(U_FLAGS) & ST_RAY) {
            extend_slot(sc, recs, stack, ps);
            shade_slot<MAT, MEDIUM, LIGHTS_AREA_ONLY, true>(rc, sc, ps, PU(U_FLAGS), dummy, dummy, dummy, dummy);
        }
    }
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
