"""Derive the jump table of Xoshiro256++'s state transition: POLY[b] = x^(2^b) mod P(x), P = characteristic polynomial of the
(GF(2)-linear) state update, b = 0..31.  advance(state, n) = product over set bits b of n of jump(state, POLY[b]), where
jump(state, poly) = XOR over set bits i of poly of T^i(state)  (the construction of the generator's published jump() function).
P is found with Berlekamp-Massey on one output bit of the state sequence; checked against the published JUMP constant (x^(2^128))
and against plain stepping.  Prints the table as C initialisers (pasted into csrc/kernels/rngjump.h)."""
M64 = (1 << 64) - 1
def rotl(x, k): return ((x << k) | (x >> (64 - k))) & M64
def step(s):
    s0, s1, s2, s3 = s
    t = (s1 << 17) & M64
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45)
    return (s0, s1, s2, s3)

def berlekamp_massey(bits):
    # returns connection polynomial C (int bitset, C[0] = 1) and L
    n = len(bits); C = 1; B = 1; L = 0; m = 1
    for i in range(n):
        d = bits[i]
        for j in range(1, L + 1):
            if (C >> j) & 1: d ^= bits[i - j]
        if d == 0: m += 1
        elif 2 * L <= i:
            T = C; C ^= B << m; L = i + 1 - L; B = T; m = 1
        else:
            C ^= B << m; m += 1
    return C, L

def polymulmod(a, b, P, deg):
    r = 0
    while b:
        if b & 1: r ^= a
        b >>= 1; a <<= 1
        if (a >> deg) & 1: a ^= P
    return r

s = (0x0123456789abcdef, 0xfedcba9876543210, 0xdeadbeefcafef00d, 0x1234567811223344)
bits = []
t = s
for i in range(1024):
    bits.append(t[0] & 1); t = step(t)
C, L = berlekamp_massey(bits)
assert L == 256, L
# connection polynomial: sum_j C_j s_{i-j} = 0  ->  characteristic polynomial P(x) = x^L * C(1/x)
P = 0
for j in range(L + 1):
    if (C >> j) & 1: P |= 1 << (L - j)
polys = []
x = 2  # the polynomial "x"
cur = x
for b in range(256):
    polys.append(cur)
    cur = polymulmod(cur, cur, P, 256)
JUMP = [0x180ec6d33cfd0aba, 0xd5a61266f0c9392c, 0xa9582618e03fc9aa, 0x39abdc4529b1661c]
jump128 = sum(w << (64 * i) for i, w in enumerate(JUMP))
assert polys[128] == jump128, 'x^(2^128) mod P differs from the published JUMP constant'
LONG = [0x76e15d3efefdcbbf, 0xc5004e441c522fb3, 0x77710069854ee241, 0x39109bb02acbe635]
assert polys[192] == sum(w << (64 * i) for i, w in enumerate(LONG))

def jump(s, poly):
    acc = (0, 0, 0, 0)
    for i in range(256):
        if (poly >> i) & 1: acc = tuple(a ^ b for a, b in zip(acc, s))
        s = step(s)
    return acc
def advance(s, n):
    b = 0
    while n:
        if n & 1: s = jump(s, polys[b])
        n >>= 1; b += 1
    return s
import random
random.seed(1)
for n in (0, 1, 2, 63, 64, 255, 256, 1000, 4097, 123457):
    t = s
    for _ in range(n): t = step(t)
    assert advance(s, n) == t, n
print('// x^(2^b) mod P(x), b = 0..31, P = characteristic polynomial of the Xoshiro256 state transition (scratch/r4/xoshiro_jump.py;')
print('// entry 128 of the same recurrence reproduces the generator\'s published JUMP constant)')
print('static constexpr unsigned long long kRngJumpPoly[32][4] = {')
for b in range(32):
    w = [(polys[b] >> (64 * i)) & M64 for i in range(4)]
    print('    {' + ', '.join('0x%016xull' % v for v in w) + '},')
print('};')
