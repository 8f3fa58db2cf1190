// chain_stream.hip — k_stream_chain instantiations for scenes that stream their BVH from L2 / HBM; see chain.hip.h
// (chain_stream_fast.hip compiles this file again with RL_FAST_MATH: launcher launch_chain_stream_fast)
#include <cstddef>
#include <cstdio>
#include <cstring>

#define RL_TRAVERSE_SPARSE 1      // trace.hip.h: no voted trips in this translation unit (k_stream_chain: one or two live lanes per wave)
#include "common.hip.h"
#include "chain.hip.h"

namespace rl {
#ifdef RL_FAST_MATH
void launch_chain_stream_fast(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_chain_impl<false>(mat, medium, grid, block, lds_bytes, st, rc, ds, stc);
}
#else
void launch_chain_stream(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc) {
    launch_chain_impl<false>(mat, medium, grid, block, lds_bytes, st, rc, ds, stc);
}
void dump_chain_timers_stream() { dump_chain_timers_impl<false>(); }
#endif
}  // namespace rl
