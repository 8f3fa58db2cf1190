// fused_stream.hip — k_path_fused instantiations for scenes that stream their BVH from L2 / HBM; see fused.hip.h
// (fused_stream_fast.hip compiles this file again with RL_FAST_MATH: same code, tolerance numerics, launcher launch_fused_stream_fast)
#include <cstdio>
#include <cstring>

#include "common.hip.h"
#include "fused.hip.h"

namespace rl {
#ifdef RL_FAST_MATH
// test hook (rl_debug_trace_batch_fast): batched closest hits through the tolerance build's traversal of streaming scenes — the quantised BVH4
__global__ void __launch_bounds__(256) k_trace_batch_fast(DeviceScene sc, StackConf stc, unsigned n, const float* o, const float* d, float* t_out, int* mesh_out, int* tri_out, int* steps_out) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    recs.nodes = reinterpret_cast<const float4*>(sc.nodes4);
    recs.tris = reinterpret_cast<const float4*>(sc.tris);
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStack stack = make_stack(stc, reinterpret_cast<unsigned*>(smem), i);
    if (i >= n) return;
    V3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    traverse<false>(recs, sc.root4, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]), ro, rd, kEps, kF32Max, hit, stack);
    t_out[i] = hit.t; steps_out[i] = hit.steps;
    if (hit.prim >= 0) { mesh_out[i] = sc.tris[hit.prim].mesh; tri_out[i] = sc.tris[hit.prim].tri; } else { mesh_out[i] = -1; tri_out[i] = -1; }
}
void launch_trace_batch_fast(dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const DeviceScene& ds, const StackConf& stc, unsigned n, const float* o, const float* d, float* t_out,
                             int* mesh_out, int* tri_out, int* steps_out) {
    hipLaunchKernelGGL(k_trace_batch_fast, grid, block, lds_bytes, st, ds, stc, n, o, d, t_out, mesh_out, tri_out, steps_out);
}
void launch_fused_stream_fast(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_fused_impl<false>(mat, medium, area_only, grid, block, lds_bytes, st, rc, ds, stc);
}
#else
void launch_fused_stream(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_fused_impl<false>(mat, medium, area_only, grid, block, lds_bytes, st, rc, ds, stc);
}
void dump_stage_timers_stream() { dump_stage_timers_impl<false>(); }
#endif
}  // namespace rl
