// rngjump.h — rng_advance: the block sampler n draws further down its stream without drawing them one by one.
//
// Xoshiro256++'s state update (SmallRng of rand 0.8.5, samplers/independent.rs:5-34) is linear over GF(2), so advancing by n is a
// polynomial in the transition T evaluated at the state: state(n) = XOR over the set bits i of (x^n mod P) of T^i(state), P the
// characteristic polynomial of T — the construction of the generator's own published jump() (n = 2^128).  The table holds
// x^(2^b) mod P for b = 0..31; advancing by an arbitrary n < 2^32 applies the entries of n's set bits (bits 0..7 are plain steps).
// Used by k_stream_spec (spec.hip.h) to start a lane in the middle of a block's stream.  Derived and checked against the published
// JUMP / LONG_JUMP constants and against plain stepping by scratch/r4/xoshiro_jump.py; device check: tests/test_gpu_parity.py
// (test_rng_advance_equals_stepping).
#pragma once

namespace rl {

// x^(2^b) mod P(x), b = 0..31, P = characteristic polynomial of the Xoshiro256 state transition (scratch/r4/xoshiro_jump.py;
// entry 128 of the same recurrence reproduces the generator's published JUMP constant)
static __constant__ unsigned long long c_rng_jump[32][4] = {
    {0x0000000000000002ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000000004ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000000010ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000000100ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000010000ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000100000000ull, 0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000000000ull, 0x0000000000000001ull, 0x0000000000000000ull, 0x0000000000000000ull},
    {0x0000000000000000ull, 0x0000000000000000ull, 0x0000000000000001ull, 0x0000000000000000ull},
    {0x9d116f2bb0f0f001ull, 0x0280002bcefd1a5eull, 0x04b4edcf26259f85ull, 0x0003c03c3f3ecb19ull},
    {0xc7327d130e34b489ull, 0x81f675e7a4ef7d84ull, 0x6dd49b656055c9daull, 0xbe7976372e930435ull},
    {0x060106bbbe4ff028ull, 0x1be1d76854ddda93ull, 0x8456faeb6230d984ull, 0x65507439cf43f0e2ull},
    {0x876c2301125a85c0ull, 0x15fe822628b16f04ull, 0x3c8ca36ec9a74fa7ull, 0x51edef31819e01ffull},
    {0xd7f4e8da7e228b85ull, 0xd638d47ec5bcf595ull, 0xaa6eb691cbf9ce10ull, 0x0f41cce3698fad39ull},
    {0x669da12373880674ull, 0xb1df898a4a6f1548ull, 0x32104b94fe2534d3ull, 0xda66e09e52b341d1ull},
    {0x4f20eb915e780231ull, 0x3886af219b885248ull, 0x023ecbee3f717fceull, 0x3cec2c375bef249cull},
    {0x449b3ae793888c8cull, 0xc3ce2f061f077568ull, 0xa69393ac0d837e54ull, 0x1a9dcf944ae47603ull},
    {0x7e89ac5ca2fbf2c7ull, 0x92ae7ca370c0bf6bull, 0xef43beaa06f02fb8ull, 0xd87f8ce230817a21ull},
    {0x6c4adbe18e29df8aull, 0x54adade3697d477full, 0xf0c168649cdba61full, 0xbd53027696368bbbull},
    {0x1a673fecf40e36b8ull, 0xf2c602feb5ed002bull, 0x1ea49b5067452594ull, 0xf78a97c0d882cd37ull},
    {0xef4606da56224c47ull, 0x770323eab8d437bdull, 0x590923d02ec52531ull, 0x1639a36e0968e3c5ull},
    {0x31d9d05c5d95f3cdull, 0x7cde241817a3ce0full, 0x2f679f694a74c76aull, 0x8b3919a9d298a415ull},
    {0x6b6622ae9590047aull, 0xeace6d3840b79fefull, 0xd9b36372fd70ec83ull, 0x624eb7b63c322e71ull},
    {0x1b91fd9ba98d9e23ull, 0xeb2c7e29d3c33d2eull, 0xcebbfd2ef4e9aff4ull, 0x2bac5517c9469796ull},
    {0x01f356e6083fe109ull, 0xba0ffb6562a3a28aull, 0x657a6b736317866bull, 0xfb678bd3e5dac186ull},
    {0xc5461100f197a7e8ull, 0xe46916a1426b676dull, 0xf3469dbb4fe25d26ull, 0xf5c010059e83bc3full},
    {0x22dc028cb8c259dcull, 0x3eec4eb6495ce5aaull, 0x5de3e273dc7b84dcull, 0xe677849e207f6afdull},
    {0x832d418900fd3b0full, 0x114e10c3b7c36788ull, 0xdf2332a778d9c8dcull, 0x0d19a1bdceb7522cull},
    {0xe2d0c9c10e8d7157ull, 0x8b3ed7c37e947e38ull, 0x98273f4d18ad073eull, 0xf38f7e750d5f4f2aull},
    {0xe7109518f3510d70ull, 0x34f30137eadb90b9ull, 0x6d48dd206d56754dull, 0xafa9e3fe5fea15c3ull},
    {0x8ee774f507ec9f39ull, 0xd7c26ebd51ecf6c4ull, 0xc76a456d998ddc4cull, 0x1ca234ff511bcb05ull},
    {0x4905d8261158a7bcull, 0x352f8b5d2137de83ull, 0xe0e9fa345826626dull, 0x3e667662caa54d16ull},
    {0x272a32be4bac7912ull, 0xe1185a166bb38173ull, 0x82b9aa358fe2ed58ull, 0xa43d37468704d536ull},
};

// one application of a table entry: 256 state steps, the states at the polynomial's set bits XORed together
static __device__ __noinline__ void rng_jump(Rng& r, int b) {
    unsigned long long a0 = 0ull, a1 = 0ull, a2 = 0ull, a3 = 0ull;
#pragma unroll 1
    for (int w = 0; w < 4; w++) {
        const unsigned long long pw = c_rng_jump[b][w];        // wave-uniform: scalar loads, scalar branches
#pragma unroll 8
        for (int i = 0; i < 64; i++) {
            if ((pw >> i) & 1ull) { a0 ^= r.s0; a1 ^= r.s1; a2 ^= r.s2; a3 ^= r.s3; }
            const unsigned long long t = r.s1 << 17;
            r.s2 ^= r.s0; r.s3 ^= r.s1; r.s1 ^= r.s2; r.s0 ^= r.s3; r.s2 ^= t;
            r.s3 = rotl64(r.s3, 45);
        }
    }
    r.s0 = a0; r.s1 = a1; r.s2 = a2; r.s3 = a3;
}
// the sampler after n more draws (n may differ per lane: lanes whose bit b is clear wait while the others apply entry b)
RL_DEV void rng_advance(Rng& r, unsigned n) {
    for (unsigned k = n & 255u; k > 0u; k--) rng_next_u64(r);
    for (int b = 8; b < 32 && (n >> b) != 0u; b++)
        if ((n >> b) & 1u) rng_jump(r, b);
}

}  // namespace rl
