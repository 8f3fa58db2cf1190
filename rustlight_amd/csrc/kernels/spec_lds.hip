// spec_lds.hip — k_stream_spec instantiations for scenes staged in LDS; see spec.hip.h
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "common.hip.h"
#include "spec.hip.h"

namespace rl {
void launch_spec_lds(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const SpecConf& spc) {
    launch_spec_impl<true>(mat, medium, grid, block, lds_bytes, st, rc, ds, stc, spc);
}
}  // namespace rl
