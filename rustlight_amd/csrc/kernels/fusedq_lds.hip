// fusedq_lds.hip — the queue-fed instantiations of k_path_fused for scenes staged in LDS: the evaluation pass of reference-order streams, launched beside the chain pass
// and fed by its completion queue (fused.hip.h, QUEUE = true; pathstate.hip.h: DoneQueue).  A translation unit of its own: compiles next to fused_lds.hip.
#include <cstdio>
#include <cstring>

#define RL_FUSED_QUEUE 1
#include "common.hip.h"
#include "fused.hip.h"

namespace rl {
void launch_fusedq_lds(int mat, bool medium, bool area_only, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_fused_impl<true>(mat, medium, area_only, grid, block, lds_bytes, st, rc, ds, stc);
}
}  // namespace rl
