// spec_stream.hip — k_stream_spec instantiations for scenes that stream their BVH from L2 / HBM; see spec.hip.h
// (full waves walk here, so the traversal keeps its voted trips — unlike chain_stream.hip)
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "common.hip.h"
#include "spec.hip.h"

namespace rl {
void launch_spec_stream(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const SpecConf& spc) {
    launch_spec_impl<false>(mat, medium, grid, block, lds_bytes, st, rc, ds, stc, spc);
}
}  // namespace rl
