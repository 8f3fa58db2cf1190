// fused_stream_fast.hip — the `numerics = fast` build of fused_stream.hip (rl_path_params.numerics = RL_NUMERICS_FAST; DESIGN.md §2 "Tolerance mode")
#define RL_FAST_MATH 1
#include "fused_stream.hip"
