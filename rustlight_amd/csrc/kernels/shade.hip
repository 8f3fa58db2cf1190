// shade.hip — the shading kernels of the wavefront pipeline: k_shade<MAT, MEDIUM> (single-BSDF scenes) and k_shade_sorted (material sort)
#include "common.hip.h"

namespace rl {

// k_shade<MAT, MEDIUM>: scenes with a single BSDF type — every live slot goes straight to that BSDF's code.
template <int MAT, bool MEDIUM>
__global__ void __launch_bounds__(256) k_shade(RenderConst rc, DeviceScene sc, Pool pool) {
    unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned n_vertices = 0, n_draws = 0, n_shadow = 0, n_ext = 0;
    PoolState ps{pool, slot};
    unsigned flags = slot < pool.P ? PU(U_FLAGS) : 0u;
    if (flags & ST_RAY) shade_slot<MAT, MEDIUM>(rc, sc, ps, flags, n_vertices, n_draws, n_shadow, n_ext);
    {
        const int which[4] = {STAT_VERTICES, STAT_DRAWS, STAT_SHADOW_RAYS, STAT_EXT_RAYS};
        const unsigned vals[4] = {n_vertices, n_draws, n_shadow, n_ext};
        block_stats<4>(rc.partials, which, vals);
    }
}

// k_shade_sorted<MEDIUM>: mixed-material scenes.  Stream compaction + material sort with wave64 ballot and
// prefix popcounts, local to the workgroup's 256 slots (no global atomics, no queue in HBM): live slots are
// binned by the BSDF type of the surface they hit, then each BSDF's code runs once over its packed bin, so
// a wave never mixes two BSDFs.
static constexpr int kNumBins = 5;
// CHUNKS: 256-slot chunks of the pool per workgroup.  With sample-parallel pixels on a scene most camera rays miss, only one
// slot in eight carries a vertex; one chunk would leave a single part-filled wave per workgroup (and this kernel's register
// footprint allows 3 workgroups per CU), so sparse pools are gathered four chunks at a time into full waves
// (508 k-triangle scene, 128 spp: 781 -> 645 ms; the traversal kernels gain nothing from the same trick).
// 3 waves/SIMD (168 VGPRs, 11 spilled) beat the unconstrained 182-VGPR build at 2 waves and a 128-VGPR build at 4 (647 / 621 / 653 ms)
#ifndef RL_SORT_WAVES
#define RL_SORT_WAVES 3
#endif
template <bool MEDIUM, unsigned CHUNKS>
__global__ void __launch_bounds__(256, RL_SORT_WAVES) k_shade_sorted(RenderConst rc, DeviceScene sc, Pool pool) {
    __shared__ unsigned s_list[256 * CHUNKS];
    __shared__ unsigned s_cnt[kNumBins][CHUNKS][4];
    unsigned n_vertices = 0, n_draws = 0, n_shadow = 0, n_ext = 0;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned slot[CHUNKS], rank[CHUNKS];
    int bin[CHUNKS];
#pragma unroll
    for (unsigned c = 0; c < CHUNKS; c++) {
        slot[c] = (blockIdx.x * CHUNKS + c) * blockDim.x + threadIdx.x;
        const unsigned flags = slot[c] < pool.P ? pool.u[(size_t)U_FLAGS * pool.P + slot[c]] : 0u;
        bin[c] = -1;
        if (flags & ST_RAY) {
            const int prim = (int)pool.u[(size_t)U_PRIM * pool.P + slot[c]];
            bin[c] = 0;
            if (prim >= 0) bin[c] = sc.materials[sc.meshes[sc.tris[prim].mesh].material].type;
        }
        rank[c] = 0;
#pragma unroll
        for (int b = 0; b < kNumBins; b++) {
            const unsigned long long mask = __ballot(bin[c] == b);
            if (bin[c] == b) rank[c] = __popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0u) s_cnt[b][c][wave] = (unsigned)__popcll(mask);
        }
    }
    __syncthreads();
    unsigned bin_begin[kNumBins + 1];
    unsigned my_off[CHUNKS];
    unsigned run = 0;
#pragma unroll
    for (int b = 0; b < kNumBins; b++) {
        bin_begin[b] = run;
#pragma unroll
        for (unsigned c = 0; c < CHUNKS; c++)
#pragma unroll
            for (unsigned w = 0; w < 4u; w++) { if (b == bin[c] && w == wave) my_off[c] = run; run += s_cnt[b][c][w]; }
    }
    bin_begin[kNumBins] = run;
#pragma unroll
    for (unsigned c = 0; c < CHUNKS; c++) if (bin[c] >= 0) s_list[my_off[c] + rank[c]] = slot[c];
    __syncthreads();
#define RL_SHADE_BIN(B)                                                                                       \
    { const unsigned n = bin_begin[(B) + 1] - bin_begin[B];                                                   \
      for (unsigned i = threadIdx.x; i < n; i += blockDim.x) { PoolState pb{pool, s_list[bin_begin[B] + i]};    \
          shade_slot<B, MEDIUM>(rc, sc, pb, pb.u(U_FLAGS), n_vertices, n_draws, n_shadow, n_ext); } }
    RL_SHADE_BIN(0) RL_SHADE_BIN(1) RL_SHADE_BIN(2) RL_SHADE_BIN(3) RL_SHADE_BIN(4)
#undef RL_SHADE_BIN
    {
        const int which[4] = {STAT_VERTICES, STAT_DRAWS, STAT_SHADOW_RAYS, STAT_EXT_RAYS};
        const unsigned vals[4] = {n_vertices, n_draws, n_shadow, n_ext};
        block_stats<4>(rc.partials, which, vals);
    }
}

template <int MAT>
static void launch_shade(bool medium, dim3 grid, dim3 block, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const Pool& pool) {
    if (medium) hipLaunchKernelGGL((k_shade<MAT, true>), grid, block, 0, st, rc, ds, pool);
    else hipLaunchKernelGGL((k_shade<MAT, false>), grid, block, 0, st, rc, ds, pool);
}
void launch_shade_type(int type, bool medium, dim3 grid, dim3 block, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const Pool& pool) {
    switch (type) {
        case BSDF_DIFFUSE: launch_shade<BSDF_DIFFUSE>(medium, grid, block, st, rc, ds, pool); break;
        case BSDF_PHONG: launch_shade<BSDF_PHONG>(medium, grid, block, st, rc, ds, pool); break;
        case BSDF_METAL: launch_shade<BSDF_METAL>(medium, grid, block, st, rc, ds, pool); break;
        case BSDF_GLASS: launch_shade<BSDF_GLASS>(medium, grid, block, st, rc, ds, pool); break;
        default: launch_shade<BSDF_SUBSTRATE>(medium, grid, block, st, rc, ds, pool); break;
    }
}


void launch_shade_sorted(bool medium, unsigned chunks, dim3 grid, dim3 block, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const Pool& pool) {
    if (chunks == 4u) {
        if (medium) hipLaunchKernelGGL((k_shade_sorted<true, 4>), grid, block, 0, st, rc, ds, pool);
        else hipLaunchKernelGGL((k_shade_sorted<false, 4>), grid, block, 0, st, rc, ds, pool);
    } else {
        if (medium) hipLaunchKernelGGL((k_shade_sorted<true, 1>), grid, block, 0, st, rc, ds, pool);
        else hipLaunchKernelGGL((k_shade_sorted<false, 1>), grid, block, 0, st, rc, ds, pool);
    }
}

}  // namespace rl
