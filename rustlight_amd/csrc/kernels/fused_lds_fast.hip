// fused_lds_fast.hip — the `numerics = fast` build of fused_lds.hip (rl_path_params.numerics = RL_NUMERICS_FAST; DESIGN.md §2 "Tolerance mode")
#define RL_FAST_MATH 1
#include "fused_lds.hip"
