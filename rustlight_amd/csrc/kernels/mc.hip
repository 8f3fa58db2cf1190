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
        recs.nodes = reinterpret_cast<const float4*>(sc.nodes);
        recs.tris = reinterpret_cast<const float4*>(sc.tris);
    }
    const unsigned item = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStackT<LDS_SCENE> stack = make_stack<LDS_SCENE>(stc, reinterpret_cast<unsigned*>(after_scene), item);
    unsigned n_samples = 0, n_draws = 0, n_ext = 0, n_shadow = 0, n_vertices = 0;
    if (item < rc.n_items) {
        const float inv = rc.inv_spp;
        if (rc.stream_mode == RL_STREAM_PER_SAMPLE) {
            Rng pixel_rng = rng_seed(rc.item_seed[item], rc.seed_variant);
            const unsigned pix = rc.item_pixel[item];
            Col acc = czero();
            for (unsigned s = 0; s < rc.spp; s++) {
                Rng rng = rng_seed(rng_next_u64(pixel_rng), rc.seed_variant);
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


void launch_pixel_mc(int kind, bool lds_scene, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const McConst& mp) {
    if (kind == 0) { if (lds_scene) hipLaunchKernelGGL((k_pixel_mc<0, true>), grid, block, lds_bytes, st, rc, ds, stc, mp); else hipLaunchKernelGGL((k_pixel_mc<0, false>), grid, block, lds_bytes, st, rc, ds, stc, mp); }
    else { if (lds_scene) hipLaunchKernelGGL((k_pixel_mc<1, true>), grid, block, lds_bytes, st, rc, ds, stc, mp); else hipLaunchKernelGGL((k_pixel_mc<1, false>), grid, block, lds_bytes, st, rc, ds, stc, mp); }
}

}  // namespace rl
