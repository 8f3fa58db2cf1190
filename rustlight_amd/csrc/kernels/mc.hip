// mc.hip — k_pixel_mc: the `ao` and `direct` integrators (device functions in mc.hip.h)
#include "common.hip.h"
#include "mc.hip.h"

namespace rl {

// same occupancy targets as the fused path kernel (`direct`, 1080p x 16 spp, unconstrained 191 VGPRs = 2 waves/SIMD vs 4 / 6: cbox 6.7 vs 5.2 / 5.7 ms,
// 508 k triangles 36.1 vs 22.2 / 21.4 ms)
template <int KIND, bool LDS_SCENE>
__global__ void __launch_bounds__(256, LDS_SCENE ? RL_FUSED_WAVES : RL_FUSED_WAVES_STREAMING) k_pixel_mc(RenderConst rc, DeviceScene sc, StackConf stc, McConst mp) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    float4* after_scene = smem;
    if (LDS_SCENE) {
        stage_scene_lds(sc, smem, smem + lds_nodes_float4s(sc.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes_float4s(sc.n_nodes);
        after_scene = smem + lds_scene_float4s(sc.n_nodes, sc.n_prims);
    } else {
        recs.nodes = streamed_nodes<TravStackT<false>>(sc);
        recs.tris = reinterpret_cast<const float4*>(sc.tris);
    }
    const unsigned item = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStackT<LDS_SCENE> stack = make_stack<LDS_SCENE>(stc, reinterpret_cast<unsigned*>(after_scene), item);
    unsigned n_samples = 0, n_draws = 0, n_ext = 0, n_shadow = 0, n_vertices = 0;
    if (item < rc.n_items) {
        const float inv = rc.inv_spp;
        if (rc.stream_mode != RL_STREAM_REFERENCE_ORDER) {
            // per-pixel work items: RL_STREAM_PER_SAMPLE (sampler forked per pixel and per sample), or the second pass of reference-order streams
            // (kStreamGivenStates: every sample starts from the block sampler's state k_mc_chain recorded for it)
            const bool given = rc.stream_mode == kStreamGivenStates;
            Rng pixel_rng = given ? Rng{0ull, 0ull, 0ull, 0ull} : rng_seed(rc.item_seed[item], rc.seed_variant);
            const unsigned pix = rc.item_pixel[item];
            Col acc = czero();
            for (unsigned s = 0; s < rc.spp; s++) {
                Rng rng = given ? load_sample_state(rc, s, item) : rng_seed(rng_next_u64(pixel_rng), rc.seed_variant);
                acc = acc + mc_compute_pixel<KIND>(sc, recs, stack, mp, pix % rc.W, pix / rc.W, rng, n_draws, n_ext, n_shadow, n_vertices);
                n_samples++;
            }
            Col px = scale_unguarded(acc, inv);
            rc.out[3 * (size_t)pix] = px.r; rc.out[3 * (size_t)pix + 1] = px.g; rc.out[3 * (size_t)pix + 2] = px.b;
        } else {
            unsigned bx, by, bw, bh;
            const unsigned b = rc.owned_blocks[item];
            block_geometry(rc, b, &bx, &by, &bw, &bh);
            Rng rng = rng_seed(rc.block_seeds[b], rc.seed_variant);
            for (unsigned iy = 0; iy < bh; iy++)
                for (unsigned ix = 0; ix < bw; ix++) {
                    Col acc = czero();
                    for (unsigned s = 0; s < rc.spp; s++) {
                        acc = acc + mc_compute_pixel<KIND>(sc, recs, stack, mp, bx + ix, by + iy, rng, n_draws, n_ext, n_shadow, n_vertices);
                        n_samples++;
                    }
                    Col px = scale_unguarded(acc, inv);
                    const size_t pix = (size_t)(by + iy) * rc.W + (bx + ix);
                    rc.out[3 * pix] = px.r; rc.out[3 * pix + 1] = px.g; rc.out[3 * pix + 2] = px.b;
                }
        }
    }
    {
        const int which[5] = {STAT_SAMPLES, STAT_VERTICES, STAT_DRAWS, STAT_SHADOW_RAYS, STAT_EXT_RAYS};
        const unsigned vals[5] = {n_samples, n_vertices, n_draws, n_shadow, n_ext};
        block_stats<5>(rc.partials, which, vals);
    }
}

// k_mc_chain<KIND> — first pass of reference-order streams for `ao` / `direct` (the form k_stream_chain has for `path`): one lane per 16x16 block, the chains
// dealt to every 2^item_shift-th lane, walks (iy, ix, sample) on the block's own sampler and records the state at the start of every camera sample.  How many
// numbers a sample takes is decided by its camera ray alone: 2, + 2 for `ao` when the hit faces the ray (or normal_correction), + 4 per light sample + 2 per BSDF
// sample for `direct` when it does (ao.rs:20-70, direct.rs:21-233: every later draw is taken whether or not it contributes).
template <int KIND, bool LDS_SCENE>
__global__ void __launch_bounds__(256, LDS_SCENE ? RL_FUSED_WAVES : RL_FUSED_WAVES_STREAMING) k_mc_chain(RenderConst rc, DeviceScene sc, StackConf stc, McConst mp) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    float4* after_scene = smem;
    if (LDS_SCENE) {
        stage_scene_lds(sc, smem, smem + lds_nodes_float4s(sc.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes_float4s(sc.n_nodes);
        after_scene = smem + lds_scene_float4s(sc.n_nodes, sc.n_prims);
    } else {
        recs.nodes = streamed_nodes<TravStackT<false>>(sc);
        recs.tris = reinterpret_cast<const float4*>(sc.tris);
    }
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStackT<LDS_SCENE> stack = make_stack<LDS_SCENE>(stc, reinterpret_cast<unsigned*>(after_scene), tid);
    const unsigned item = (tid & ((1u << rc.item_shift) - 1u)) == 0u ? (tid >> rc.item_shift) : 0xffffffffu;
    if (item >= rc.n_items) return;
    unsigned bx, by, bw, bh;
    const unsigned b = rc.owned_blocks[item];
    block_geometry(rc, b, &bx, &by, &bw, &bh);
    Rng rng = rng_seed(rc.block_seeds[b], rc.seed_variant);
    const unsigned base = rc.block_item_base[item];
    const V3 o = mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]);
    const unsigned after_hit = KIND == 0 ? 2u : 4u * mp.nb_light_samples + 2u * mp.nb_bsdf_samples;
    for (unsigned c = 0; c < bw * bh; c++) {
        const unsigned px = bx + c % bw, py = by + c / bw;
        for (unsigned s = 0; s < rc.spp; s++) {
            store_sample_state(rc, s, base + c, rng);
            const float u = (float)px + rng_next_f32(rng);
            const float v = (float)py + rng_next_f32(rng);
            const V3 d = camera_direction(sc, u, v);
            Hit hit;
            if (!trace_closest(sc, recs, stack, o, d, hit)) continue;
            const SurfacePoint sp = fill_intersection(sc, hit.prim, hit.u, hit.v, o, d, hit.t);
            const bool more = KIND == 0 ? (mp.normal_correction || !(sp.wi.z <= 0.0f)) : !(sp.wi.z <= 0.0f);
            if (more) for (unsigned k = 0; k < after_hit; k++) rng_next_u64(rng);
        }
    }
}

void launch_mc_chain(int kind, bool lds_scene, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const McConst& mp) {
    if (kind == 0) { if (lds_scene) hipLaunchKernelGGL((k_mc_chain<0, true>), grid, block, lds_bytes, st, rc, ds, stc, mp); else hipLaunchKernelGGL((k_mc_chain<0, false>), grid, block, lds_bytes, st, rc, ds, stc, mp); }
    else { if (lds_scene) hipLaunchKernelGGL((k_mc_chain<1, true>), grid, block, lds_bytes, st, rc, ds, stc, mp); else hipLaunchKernelGGL((k_mc_chain<1, false>), grid, block, lds_bytes, st, rc, ds, stc, mp); }
}

void launch_pixel_mc(int kind, bool lds_scene, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const McConst& mp) {
    if (kind == 0) { if (lds_scene) hipLaunchKernelGGL((k_pixel_mc<0, true>), grid, block, lds_bytes, st, rc, ds, stc, mp); else hipLaunchKernelGGL((k_pixel_mc<0, false>), grid, block, lds_bytes, st, rc, ds, stc, mp); }
    else { if (lds_scene) hipLaunchKernelGGL((k_pixel_mc<1, true>), grid, block, lds_bytes, st, rc, ds, stc, mp); else hipLaunchKernelGGL((k_pixel_mc<1, false>), grid, block, lds_bytes, st, rc, ds, stc, mp); }
}

}  // namespace rl
