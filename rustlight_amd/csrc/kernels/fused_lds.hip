// fused_lds.hip — k_path_fused instantiations for scenes staged in LDS (nodes + triangles <= 48 KiB); see fused.hip.h
// (fused_lds_fast.hip compiles this file again with RL_FAST_MATH: same code, tolerance numerics, launcher launch_fused_lds_fast)
#include <cstdio>
#include <cstring>

#include "common.hip.h"
#include "fused.hip.h"

namespace rl {
#ifdef RL_FAST_MATH
void launch_fused_lds_fast(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_fused_impl<true>(mat, medium, area_only, grid, block, lds_bytes, st, rc, ds, stc);
}
#else
void launch_fused_lds(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_fused_impl<true>(mat, medium, area_only, grid, block, lds_bytes, st, rc, ds, stc);
}
void dump_stage_timers(bool lds_scene) { if (lds_scene) dump_stage_timers_impl<true>(); else dump_stage_timers_stream(); }
#endif
}  // namespace rl
