// chain_stream_fast.hip — the `numerics = fast` build of chain_stream.hip
#define RL_FAST_MATH 1
#include "chain_stream.hip"
