// fused_stream.hip — k_path_fused instantiations for scenes that stream their BVH from L2 / HBM; see fused.hip.h
// (fused_stream_fast.hip compiles this file again with RL_FAST_MATH: same code, tolerance numerics, launcher launch_fused_stream_fast)
#include <cstdio>
#include <cstring>

#include "common.hip.h"
#include "fused.hip.h"

namespace rl {
#ifdef RL_FAST_MATH
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
