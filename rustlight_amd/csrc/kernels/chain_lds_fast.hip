// chain_lds_fast.hip — the `numerics = fast` build of chain_lds.hip
#define RL_FAST_MATH 1
#include "chain_lds.hip"
