"""Oracle pinning, part 1: the random stream.  CPU only.

B.1 is the *published* Xoshiro256++ reference vector — the one external known-answer test this
oracle is pinned to.  B.2/B.3 restate rand 0.8.5 / rand_core 0.6.4 from their documented algorithm
(third-party crates, not vendored, no Rust toolchain here) and are self-consistency checks."""
import numpy as np

from oracle import orc
from rustlight_amd import api


def test_xoshiro256pp_published_vector(built):
    r = orc.Rng.from_state([1, 2, 3, 4])
    got = [r.next_u64() for _ in range(10)]
    assert got == [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205,
                   9973669472204895162, 14011001112246962877, 12406186145184390807,
                   15849039046786891736, 10450023813501588000]
    s = api.abi.Sampler()
    s.s[0], s.s[1], s.s[2], s.s[3] = 1, 2, 3, 4
    assert [int(api.lib().rl_sampler_next_u64(s)) for _ in range(10)] == got


def test_seed_from_u64_pcg32_fill(built):
    r = orc.Rng(0, 0)
    assert list(r.state) == [0x45cdb581f973f2ec, 0xad6cad067346f087, 0x67e71733e3a3d0d0, 0xfe7d8ad772ea9bf2]
    assert [r.next_u64() for _ in range(3)] == [0x7283e4c96896188c, 0x706b7f2de031bf37, 0xfad96ea1180d0e12]
    assert list(orc.Rng(1, 0).state) == [0x4e10265d721dd8ea, 0x2e78ce42f83b9c89, 0xc2d29799da03d3ba, 0x1bfb6673ac560212]
    assert list(orc.Rng(42, 0).state) == [0x0a3d32587ba18fa4, 0xb8140169cca1b8ea, 0x54f7b41875c88c2b, 0xf220dfe4a16e448d]


def test_seed_from_u64_rand_core_value_breakage_vector(built):
    """The second PUBLISHED vector: rand_core 0.6.4's own value-breakage test of the default `SeedableRng::seed_from_u64` — the PCG32 fill that
    `SmallRng::seed_from_u64` inherits in rand 0.8.5 (its SeedableRng impl forwards only from_seed / from_rng; src/samplers/independent.rs:11,20,
    examples/cli.rs:886-890) — on an 8-byte seed: `seed_from_u64(0)` read back as one little-endian u64 must be 5029875928683246316 (rand_core/src/lib.rs,
    `test_seed_from_u64`: `assert_eq!(results[0], 5029875928683246316)`).  An 8-byte seed is the first two PCG32 words, i.e. the first state word of the 32-byte
    seed Xoshiro256++ takes: checked for the oracle AND for the product's host sampler (rl_sampler_seed)."""
    want = 5029875928683246316
    assert int(orc.Rng(0, 0).state[0]) == want
    s = api.IndependentSampler(0, 0)
    assert int(s.s.s[0]) == want
    # the same fill restated here from rand_core's documented algorithm (PCG32, MUL 6364136223846793005, INC 11634580027462260723, XSH-RR output), all four state words
    def pcg32_fill(state, n_words):
        MUL, INC, M = 6364136223846793005, 11634580027462260723, (1 << 64) - 1
        out = []
        for _ in range(n_words):
            state = (state * MUL + INC) & M
            xorshifted = (((state >> 18) ^ state) >> 27) & 0xffffffff
            rot = state >> 59
            out.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xffffffff)
        return out
    for seed in (0, 1, 42, 2 ** 63 + 12345):
        w = pcg32_fill(seed, 8)
        words = [w[2 * k] | (w[2 * k + 1] << 32) for k in range(4)]
        assert [int(x) for x in orc.Rng(seed, 0).state] == words
        p = api.IndependentSampler(seed, 0)
        assert [int(p.s.s[k]) for k in range(4)] == words
    assert pcg32_fill(0, 2)[0] | (pcg32_fill(0, 2)[1] << 32) == want


def test_seed_from_u64_splitmix_variant(built):
    r = orc.Rng(0, 1)
    assert list(r.state) == [0xe220a8397b1dcdaf, 0x6e789e6aa1b965f4, 0x06c45d188009454f, 0xf88bb8a8724c81ec]
    assert r.next_u64() == 0x53175d61490b23df


def test_f32_mapping(built):
    r = orc.Rng(0, 0)
    got = [r.next_f32() for _ in range(4)]
    assert got == [0.4473249912261963, 0.439140260219574, 0.9798802137374878, 0.4621672034263611]
    r2 = orc.Rng(0, 0)
    for g in got:                      # (next_u64 >> 40) * 2^-24
        assert g == np.float32((r2.next_u64() >> 40) * 2.0 ** -24)
    assert all(0.0 <= g < 1.0 for g in got)


def test_f32_mapping_rand_standard_edge_vectors(built):
    """rand 0.8.5's own `Standard` f32 vectors (src/distributions/float.rs, test `floats`: a generator that returns all zeros gives 0.0, one that returns 1 << 8 as its
    u32 gives EPSILON / 2, all ones gives 1 - EPSILON / 2): `gen::<f32>()` is (next_u32() >> 8) * 2^-24, and a 64-bit generator's next_u32 is the upper half of next_u64
    (rand_xoshiro: `(self.next_u64() >> 32) as u32`).  Xoshiro256++'s output is rotl(s0 + s3, 23) + s0, so states with s0 = 0 and a chosen s3 produce those three words
    as their first output: checked for the oracle's sampler and for the product's (rl_sampler_next_f32), bit for bit."""
    eps = float(np.finfo(np.float32).eps)
    rotr23 = lambda x: ((x >> 23) | (x << 41)) & (2 ** 64 - 1)
    for word, want in ((0, 0.0), (256 << 32, eps / 2), (2 ** 64 - 1, 1.0 - eps / 2)):
        state = [0, 0x9e3779b97f4a7c15, 0x0123456789abcdef, rotr23(word)]
        assert orc.Rng.from_state(state).next_u64() == word
        got = orc.Rng.from_state(state).next_f32()
        assert got == np.float32(want) and np.float32(got).tobytes() == np.float32(want).tobytes()
        s = api.abi.Sampler()
        for k in range(4): s.s[k] = state[k]
        fn = api.lib().rl_sampler_next_f32
        pv = np.float32(fn(s))
        assert pv.tobytes() == np.float32(want).tobytes(), (word, pv)


def test_block_forking_order(built):
    # master independent:0 -> block 0 seed = first next_u64; block sampler's first draws (App. B.3)
    seeds = orc.block_seeds(0, 64, 48)
    assert seeds.shape[0] == 4 * 3
    assert int(seeds[0]) == 0x7283e4c96896188c
    b0 = orc.Rng(int(seeds[0]), 0)
    assert [b0.next_f32() for _ in range(3)] == [0.12112051248550415, 0.6834843158721924, 0.47241508960723877]
    # x-major creation order: consecutive seeds walk down a column of blocks first
    m = orc.Rng(0, 0)
    assert [int(s) for s in seeds] == [m.next_u64() for _ in range(12)]


def test_product_host_sampler_matches_oracle(built):
    for variant in (0, 1):
        for seed in (0, 1, 42, 2 ** 63 + 12345):
            a = api.IndependentSampler(seed, variant)
            b = orc.Rng(seed, variant)
            assert [a.next_u64() for _ in range(5)] == [b.next_u64() for _ in range(5)]
            assert [a.next() for _ in range(5)] == [b.next_f32() for _ in range(5)]
    np.testing.assert_array_equal(api.IndependentSampler(7).block_seeds(1920, 1080), orc.block_seeds(7, 1920, 1080))
    assert api.lib().rl_block_count(1920, 1080) == 120 * 68
