// chain_lds.hip — k_stream_chain instantiations for scenes staged in LDS; see chain.hip.h
// (chain_lds_fast.hip compiles this file again with RL_FAST_MATH: launcher launch_chain_lds_fast)
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "common.hip.h"
#include "chain.hip.h"

namespace rl {
#ifdef RL_FAST_MATH
void launch_chain_lds_fast(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_chain_impl<true>(mat, medium, grid, block, lds_bytes, st, rc, ds, stc);
}
#else
void launch_chain_lds(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_chain_impl<true>(mat, medium, grid, block, lds_bytes, st, rc, ds, stc);
}
void dump_chain_timers_lds() { dump_chain_timers_impl<true>(); }
#endif
}  // namespace rl
