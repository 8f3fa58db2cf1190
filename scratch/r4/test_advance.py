import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from rustlight_amd import api
from oracle import orc
L = api.lib()
n = 256
rng = np.random.default_rng(5)
states = rng.integers(1, 2**63, size=(n, 4), dtype=np.uint64)
counts = np.concatenate([[0, 1, 2, 255, 256, 257, 511, 512, 1023, 65536, 65537, 1000003], rng.integers(0, 300000, size=n - 12)]).astype(np.uint32)
out = np.zeros_like(states)
L.rl_debug_rng_advance.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
rc = L.rl_debug_rng_advance(0, n, states.ctypes.data, counts.ctypes.data, out.ctypes.data)
assert rc == 0, rc
bad = 0
for i in range(n):
    r = orc.Rng.from_state([int(x) for x in states[i]])
    for _ in range(int(counts[i])): orc.lib().orc_rng_next_u64(r.state)
    if list(r.state) != [int(x) for x in out[i]]: bad += 1
print('rng_advance mismatches:', bad, 'of', n)
assert bad == 0
