// fusedq_stream.hip — the queue-fed instantiations of k_path_fused for scenes that stream their BVH; see fusedq_lds.hip
#include <cstdio>
#include <cstring>

#define RL_FUSED_QUEUE 1
#include "common.hip.h"
#include "fused.hip.h"

namespace rl {
void launch_fusedq_stream(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_fused_impl<false>(mat, medium, area_only, grid, block, lds_bytes, st, rc, ds, stc);
}
}  // namespace rl
