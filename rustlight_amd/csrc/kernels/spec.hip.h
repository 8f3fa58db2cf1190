// spec.hip.h — k_stream_spec: the first pass of RL_STREAM_REFERENCE_ORDER with every lane busy (round 4), and its launcher; instantiated by
// spec_lds.hip / spec_stream.hip.  Writes exactly what k_stream_chain (chain.hip.h) writes — the block sampler's state at the start of every
// camera sample — and is followed by the same second pass (k_path_fused from the recorded states).
//
// rustlight's compute_mc (src/integrators/mod.rs:420-435) consumes ONE sampler per 16x16 block over (iy, ix, sample): where sample k+1 starts
// in the block's stream is where sample k stopped, so the stream positions of a block form a serial chain t -> t + n_p(t), with n_p(t) the number
// of draws a camera sample of pixel p takes when it starts at stream offset t.  k_stream_chain walks that chain with one lane.  Here the chain is
// still followed exactly, but almost none of it is walked serially:
//   * n_p is a pure function of (block seed, pixel, offset).  Two walks of the SAME pixel that ever stand on the same offset stay together
//     for good — and walks started a few draws apart meet after a sample count of the order of the mean draws per sample (the offsets a walk
//     visits are a renewal process: measured on the Cornell box, 16 samples on average; scratch/r4/merge_sim.py).
//   * So the lanes of a block's group each take ONE pixel of the next `group` pixels and walk that pixel's function over a WINDOW of stream
//     offsets around where the pixel is expected to lie — expected from the lengths the same lanes resolved a batch earlier (the pixel one or
//     a few rows up), wide enough for the spread those lengths showed — recording (offset, sampler state) of every sample start: a TRACK.
//   * Then the true chain is threaded through the tracks pixel by pixel: the pixel's true start offset is looked up in its track; if it is
//     not on it, the owner lane walks from the true start (slow path, one lane) until it lands on a track entry — from there on the track
//     IS the chain, and the pixel's remaining samples and its end offset are read off the track.  A track that starts too late, ends too
//     early or never meets the chain only costs slow samples; nothing a track holds is used unless the chain provably stands on it.
//   * Pixels whose camera rays cannot reach the scene's bounding box (host: conservative test on the pixel's footprint) take exactly two draws
//     per sample: their states are skip-aheads (rng_advance), no walk.
//   * A pixel's window can be cut into `sub` segments walked by `sub` lanes (one wave then holds fewer pixels per batch: a shorter look-ahead,
//     narrower windows, and `sub` times fewer samples per lane): each lane walks its segment and on into the next one for a lead-in's length,
//     then finds where its walk first stands on an entry of the next lane's walk (LINK) — from there the two are the same walk.  The chain follows
//     a pixel's sub-tracks link by link; a missing link is bridged by the slow path.
//   * Where the slow path runs inside a pixel whose samples nearly all take the same number of draws c (two walks of such a pixel sit on different
//     residues of c and stay apart for most of the pixel), the chain's next positions are predictable: the group's lanes — idle while a pixel is
//     threaded — evaluate the samples that start c, 2c, ... draws on from the chain's head, and the leader consumes them in order for as long as the
//     chain really stands on them (HELPERS, below).
// The kernel is bound by LDS for occupancy (the lanes' parked state: 30 KB per workgroup): three workgroups per CU on the Cornell box, which is what
// lets every heavy block of a 1080p frame (4900 of 8160, two per wave) start at once instead of in two generations.
// The result is bit-for-bit the serial chain (tests: test_reference_order_two_pass_equals_single_pass, the fuzzer's reference-order cases;
// RL_CHAIN_SERIAL=1 keeps k_stream_chain as the cross-check).  What it buys: a block's serial work drops from 256 x spp samples to a few
// samples per pixel, and the rest runs on full waves.
#pragma once
#include "rngjump.h"

// Waves per SIMD the register budget is set for.  What limits the kernel's occupancy is LDS (the lanes' parked state: 30 KB per workgroup, + stacks + scene:
// three workgroups per CU on the Cornell box — which is what lets every heavy block of a 1080p frame start at once), so the budget is that of
// three waves: 168 VGPRs, none spilled in any instantiation (at 128 the generic-material instantiations spilled 50-100).
#ifndef RL_SPEC_WAVES
#define RL_SPEC_WAVES 3
#endif
// Scenes that stream their BVH (round 6): the kernel waits on dependent record fetches half of its wave time with 18 % of the VALU issue slots busy
// (profiles/r06_pmc_living_reference.json) and 8160 blocks at one per wave run in 2.7 generations of 3072 waves — a fourth wave per SIMD pays for its price: 128 VGPRs
// (56 spilled, 96 B of scratch per lane) and three instead of six traversal-stack levels in LDS (38.8 KB per workgroup: four of them per CU).  508 k triangles, 1080p x
// 128 spp, same box: 3 waves x 6 levels 2479-2540 ms, 4 x 4 (40.8 KB: does not fit four) 2589-2707, 4 x 3 2335-2348, 4 x 2 2321-2397, 4 x 1 2397-2447, 4 x 0 2487-2500,
// 5 x 0 2793-2850 (profiles/r06_spec_occupancy_ab_living.txt).
#ifndef RL_SPEC_WAVES_STREAMING
#define RL_SPEC_WAVES_STREAMING 4
#endif
// (the stack levels kept in LDS there: pathstate.hip.h, RL_SPEC_LDS_LEVELS_STREAMING)

namespace rl {


enum : unsigned { SM_IDLE = 0u, SM_PROBE, SM_WALK, SM_TRUTH, SM_EXTRA, SM_HELP };      // SM_EXTRA: a pixel's last lane walking on past its window while the rest of the group still walks (free filler: it would idle)
enum : unsigned { SP_PROBE = 0u, SP_WALK, SP_RESOLVE, SP_DONE };

RL_DEV Rng shfl_rng(const Rng& r, int src) {
    Rng o;
    o.s0 = __shfl(r.s0, src, 64); o.s1 = __shfl(r.s1, src, 64); o.s2 = __shfl(r.s2, src, 64); o.s3 = __shfl(r.s3, src, 64);
    return o;
}

template <int MAT, bool MEDIUM, bool LDS_SCENE, int NUM>
__global__ void __launch_bounds__(256, LDS_SCENE ? RL_SPEC_WAVES : RL_SPEC_WAVES_STREAMING) k_stream_spec(RenderConst rc_arg, DeviceScene sc_arg, StackConf stc, SpecConf spc_arg) {
    const DeviceScene& sc0 = sc_arg;
    const SpecConf* spcp = &spc_arg;      // (inside the loop: re-read from the kernarg segment like rc and sc, so that none of its fields is kept in a register across the traversal)
#define spc (*spcp)
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    float4* after_scene = smem;
    if (LDS_SCENE) {
        stage_scene_lds(sc0, smem, smem + lds_nodes_float4s(sc0.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes_float4s(sc0.n_nodes);
        after_scene = smem + lds_scene_float4s(sc0.n_nodes, sc0.n_prims);
    } else {
        recs.nodes = streamed_nodes<TravStackT<false>>(sc0);
        recs.tris = reinterpret_cast<const float4*>(sc0.tris);
    }
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, tid0 = blockIdx.x * blockDim.x;
    if (rc_arg.queue && threadIdx.x == 0u) queue_workgroup_started(rc_arg);      // (the host launches the evaluation pass beside this kernel once EVERY workgroup of it runs)
    const TravStackT<LDS_SCENE> stack = make_stack<LDS_SCENE>(stc, reinterpret_cast<unsigned*>(after_scene), tid);
    RegState ps;
#pragma unroll
    for (int i = 0; i < F_COUNT; i++) ps.fv[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < U_COUNT; i++) ps.uv[i] = 0u;
#pragma unroll
    for (int i = 0; i < Q_COUNT; i++) ps.qv[i] = 0ull;
    PU(U_PRIM) = 0xffffffffu;
    PU(U_FLAGS) = 0u;

    // ---- the group: G lanes of one wave own one block; S of them share a pixel (lane sk walks segment sk of the pixel's window), so a batch is
    // NP = G / S consecutive pixels.  The pixel's first lane (sk = 0, the LEADER) keeps the pixel's state and threads the chain through it.
    const unsigned G = spc.group, S = spc.sub, NP = G / S, lane = threadIdx.x & 63u, gl = threadIdx.x & (G - 1u);      // G = 256: the whole workgroup (4 waves) owns one block
    const unsigned pl = gl / S, sk = gl - pl * S, lt = threadIdx.x - sk;          // pixel slot, segment, the leader's thread index in the workgroup
    const unsigned long long gmask = (G >= 64u ? ~0ull : ((1ull << G) - 1ull)) << (lane & ~(G - 1u) & 63u);
    const unsigned item = tid / G;
    const bool have_block = item < rc_arg.n_owned;
    const unsigned spp = rc_arg.spp;
    unsigned bx = 0, by = 0, bw = 1, bh = 1;
    if (have_block) block_geometry(rc_arg, rc_arg.owned_blocks[item], &bx, &by, &bw, &bh);
    const unsigned c_begin = rc_arg.cursor_begin, c_end = have_block ? min(rc_arg.cursor_end, bw * bh) : 0u;
    const unsigned pix_base = have_block ? rc_arg.block_item_base[item] : 0u;
    Rng anc; anc.s0 = anc.s1 = anc.s2 = anc.s3 = 0ull;            // the block sampler where the batch begins
    if (have_block && c_begin < c_end) {
        if (c_begin == 0u) anc = rng_seed(rc_arg.block_seeds[rc_arg.owned_blocks[item]], rc_arg.seed_variant);      // the block's own sampler (mod.rs:371)
        else { const unsigned long long* q = rc_arg.chain_states + 4 * (size_t)item; anc.s0 = q[0]; anc.s1 = q[1]; anc.s2 = q[2]; anc.s3 = q[3]; }
    }
    unsigned phase = (have_block && c_begin < c_end) ? SP_PROBE : SP_DONE;
    if (rc_arg.queue && have_block && !(c_begin < c_end) && (threadIdx.x & (spc_arg.group - 1u)) == 0u) queue_push(rc_arg, item);      // nothing of the block in this chunk: done as it stands
    const unsigned* const triv_bits = spc.trivial + (size_t)(have_block ? item : 0u) * 8u;
    // a lane's track: trk_off / trk_st rows of workgroup thread t
#define OFF_OF(t) (spc.trk_off + (size_t)(tid0 + (t)) * spc.cap)
#define ST_OF(t) (spc.trk_st + (size_t)(tid0 + (t)) * spc.cap * 2u)
#define my_off OFF_OF(threadIdx.x)
#define my_st ST_OF(threadIdx.x)

    // ---- per-lane state.  Only what the traversal and shading code touches lives in registers; everything the bookkeeping between two samples
    // needs is parked in LDS ([field][thread], conflict-free) — with it in VGPRs the kernel spilled into scratch inside the traversal loop.
    // Fields of the second row are the pixel's: only its leader's copy is used (the owner of a pixel reads and writes the other lanes' rows).
    unsigned* const coldbase = reinterpret_cast<unsigned*>(after_scene) + 2 * 256 * stc.lds_levels;
    unsigned* const cold = coldbase + threadIdx.x;
    enum { K_CUR_OFF, K_M, K_STOP, K_NB_LO, K_CNT, K_SUM_N, K_SUM_N2, K_LINK_I, K_LINK_J, K_CP_I0, K_CP_J0, K_CP_K0, K_Q0,
           K_EST_L, K_EST_V, K_RES_I, K_RES_J, K_RES_K, K_RES_OFF, K_RES_ST, K_NMIN = K_RES_ST + 8, K_DN, K_DR, K_COLD_COUNT };
    static_assert(K_COLD_COUNT == kSpecColdWords, "LDS budget of k_stream_spec (host: wavefront.hip)");
    // scenes that stream their BVH run without helpers (measured: between 0 and - 2 % there once a fourth wave fits a SIMD), so their lanes do not park the two helper words
    constexpr int kCold = LDS_SCENE ? (int)K_COLD_COUNT : (int)K_COLD_COUNT - kSpecHelperWords;
    static_assert(K_DN == K_COLD_COUNT - 2 && K_DR == K_COLD_COUNT - 1, "the helper words are the last two");
#define COLD(k) cold[(k) * 256]
#define COLD_OF(t, k) coldbase[(k) * 256 + (t)]
    unsigned& cur_off = COLD(K_CUR_OFF);          // stream offset (relative to the anchor) where the sample being walked started
    unsigned& M = COLD(K_M);                      // track: entries 0..M-1 are walked samples, entry M the frontier
    unsigned& nb_lo = COLD(K_NB_LO);              // where the next lane's segment (same pixel) begins
    unsigned& stop = COLD(K_STOP);                // the walk ends with the first sample that starts at or beyond this offset
    unsigned& cnt = COLD(K_CNT); unsigned& sum_n = COLD(K_SUM_N); float& sum_n2 = reinterpret_cast<float&>(COLD(K_SUM_N2));     // draw statistics of this lane's walk
    unsigned& link_i = COLD(K_LINK_I); unsigned& link_j = COLD(K_LINK_J);     // entry link_i of this track = entry link_j of the next lane's (0xffffffff: no link)
    unsigned& cp_i0 = COLD(K_CP_I0); unsigned& cp_j0 = COLD(K_CP_J0); unsigned& cp_k0 = COLD(K_CP_K0);        // samples cp_i0 .. cp_i0 + cp_k0 - 1 of the pixel are this track's entries cp_j0 ..
    unsigned& q0 = COLD(K_Q0);                    // first block cursor of the current batch (group-uniform); this lane's pixel is block cursor q0 + pl
#define pix (pix_base + (q0 + pl - c_begin))      /* its index into sample_states */
    float& estL = reinterpret_cast<float&>(COLD(K_EST_L)); float& estV = reinterpret_cast<float&>(COLD(K_EST_V));   // leader: predicted length of the pixel in draws, its variance
    unsigned& res_i = COLD(K_RES_I); unsigned& res_j = COLD(K_RES_J); unsigned& res_k = COLD(K_RES_K);      // leader, slow walk: truth samples done, the sub-track ahead (segment, entry)
    unsigned& res_off = COLD(K_RES_OFF);          // leader: where the chain stands after the pixel (offset; the state: K_RES_ST, 8 words)
    unsigned& nmin_cnt = COLD(K_NMIN);            // fewest draws a sample of this lane's walk took (<< 16) | how many of its samples took exactly that
    // K_DN / K_DR: helper of a serial walk (below): draws of the sample it evaluated (0: not yet), the round it belongs to; its sampler after the sample: the lane's K_RES_ST
    for (int k = 0; k < kCold; k++) COLD(k) = 0u;
    q0 = c_begin;
    // the block sampler where the batch begins: one copy per group, after the per-thread planes
    unsigned long long* const ganc = reinterpret_cast<unsigned long long*>(coldbase + kCold * 256) + 4u * (threadIdx.x / G);
    auto load_anc = [&]() -> Rng { Rng r; r.s0 = ganc[0]; r.s1 = ganc[1]; r.s2 = ganc[2]; r.s3 = ganc[3]; return r; };
    auto store_anc = [&](const Rng& r) { ganc[0] = r.s0; ganc[1] = r.s1; ganc[2] = r.s2; ganc[3] = r.s3; };      // (every lane of the group writes the same value)
    store_anc(anc);
    // group-wide votes and sums: within a wave by ballot / shuffles; a group that spans the workgroup (G = 256) through the barrier (every thread of the
    // workgroup then belongs to the one group, so every call site is reached by all of them together)
    float* const gscratch = reinterpret_cast<float*>(coldbase + kCold * 256 + 8 * (256 / 16));
    // ---- a serial walk taken several samples at a time (HELPERS).  Where the chain leaves the tracks inside a pixel whose samples nearly all take the same number of draws c,
    // two walks of the pixel stay apart for long (they sit on different residues of c) and the serial walk is most of the pixel.  But then the chain's next positions are
    // predictable: from the head at offset b they are b + c, b + 2c, ... until a sample takes another count.  The group's lanes — idle while a pixel is threaded — evaluate
    // the samples that start there (lane h: state stepped h c draws on), and the pixel's leader consumes them in order for as long as the chain really stands on them:
    // sample h is on the chain iff samples 0 .. h-1 all took c draws; its state is then the sampler lane h-1 was left with.  The first sample that takes another count ends
    // the round; the next one starts from where it leaves the chain.  Nothing is used unless the chain provably stands on it.
    const unsigned gshift = 31u - (unsigned)__builtin_clz(G);
    // the group's head, 16 words: [0..7] sampler at the head, [8] its offset, [9] c, [10] round, [11] active, [12] the leader's lane in the group, [13] its pixel
    // (addresses recomputed at every use: held in registers across the loop they push the traversal into scratch)
#define ghead (reinterpret_cast<unsigned*>(gscratch + 16) + 16u * (threadIdx.x >> gshift))
    enum { GH_B = 8, GH_C, GH_ROUND, GH_ACTIVE, GH_LGL, GH_PXY };
    if (gl < 16u) ghead[gl] = 0u;
    // the group's counters: samples walked, serial, probed
#define gstat (reinterpret_cast<unsigned*>(gscratch + 16) + 16u * (256u / 16u) + 4u * (threadIdx.x >> gshift))
    if (gl < 4u) gstat[gl] = 0u;
#define st_spec_inc atomicAdd(&gstat[0], 1u)
#define st_slow_inc atomicAdd(&gstat[1], 1u)
#define st_probe_inc atomicAdd(&gstat[2], 1u)
#ifdef RL_PROBE_NO_DENSE
    const unsigned H = 0u;
#else
    const unsigned H = (LDS_SCENE && G <= 64u && spc.dense > 1u) ? min(spc.dense, G) : 0u;
#endif
    auto group_any = [&](bool p) -> bool { return G <= 64u ? (__ballot(p) & gmask) != 0ull : __syncthreads_or((int)p) != 0; };
    auto group_scan2 = [&](float& a, float& b) {      // inclusive prefix sums over the lanes of the group
        const unsigned w = G <= 64u ? G : 64u, li = G <= 64u ? gl : lane;
        for (unsigned d = 1u; d < w; d <<= 1) { const float ta = __shfl_up(a, d, 64), tb = __shfl_up(b, d, 64); if (li >= d) { a += ta; b += tb; } }
        if (G > 64u) {
            const unsigned wv = threadIdx.x >> 6;
            if (lane == 63u) { gscratch[wv] = a; gscratch[4u + wv] = b; }
            __syncthreads();
            for (unsigned k = 0; k < wv; k++) { a += gscratch[k]; b += gscratch[4u + k]; }
            __syncthreads();
        }
    };
    auto group_sum2 = [&](float& a, float& b) {
        const unsigned w = G <= 64u ? G : 64u;
        for (unsigned d = 1u; d < w; d <<= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
        if (G > 64u) {
            const unsigned wv = threadIdx.x >> 6;
            if (lane == 0u) { gscratch[8u + wv] = a; gscratch[12u + wv] = b; }
            __syncthreads();
            a = ((gscratch[8] + gscratch[9]) + gscratch[10]) + gscratch[11]; b = ((gscratch[12] + gscratch[13]) + gscratch[14]) + gscratch[15];
            __syncthreads();
        }
    };
    auto load_res_st = [&](unsigned t) -> Rng {        // of workgroup thread t (a leader)
        const unsigned* q = coldbase + K_RES_ST * 256 + t;
        Rng r;
        r.s0 = (unsigned long long)q[0] | ((unsigned long long)q[256] << 32); r.s1 = (unsigned long long)q[512] | ((unsigned long long)q[768] << 32);
        r.s2 = (unsigned long long)q[1024] | ((unsigned long long)q[1280] << 32); r.s3 = (unsigned long long)q[1536] | ((unsigned long long)q[1792] << 32);
        return r;
    };
    auto store_res_st = [&](const Rng& r) {
        COLD(K_RES_ST) = (unsigned)r.s0; COLD(K_RES_ST + 1) = (unsigned)(r.s0 >> 32); COLD(K_RES_ST + 2) = (unsigned)r.s1; COLD(K_RES_ST + 3) = (unsigned)(r.s1 >> 32);
        COLD(K_RES_ST + 4) = (unsigned)r.s2; COLD(K_RES_ST + 5) = (unsigned)(r.s2 >> 32); COLD(K_RES_ST + 6) = (unsigned)r.s3; COLD(K_RES_ST + 7) = (unsigned)(r.s3 >> 32);
    };
    auto entry_state = [&](unsigned t, unsigned k) -> Rng {      // state of entry k of workgroup thread t's track
        const ulonglong2* q = ST_OF(t) + 2u * k;
        const ulonglong2 a = q[0], b = q[1];
        Rng r; r.s0 = a.x; r.s1 = a.y; r.s2 = b.x; r.s3 = b.y;
        return r;
    };
    unsigned mode = SM_IDLE;
    unsigned pxy = 0;                             // image position of this lane's pixel, x | y << 16
    // the lane's flags, one register (as separate bools each is a lane mask in a pair of SGPRs, and the kernel is short of those too)
    unsigned bits = 0u;
    struct Bit {
        unsigned& w; const unsigned m;
        RL_DEV operator bool() const { return (w & m) != 0u; }
        RL_DEV void operator=(bool b) { w = b ? (w | m) : (w & ~m); }
    };
    Bit valid{bits, 1u}, triv{bits, 2u}, resolved{bits, 4u}, copied{bits, 8u}, have_est{bits, 16u}, planned{bits, 32u}, linked{bits, 64u};
    unsigned nd = 0;                              // draws the sample being walked has taken so far
    unsigned own = 0;                             // resolve: the pixel slot the chain stands in (group-uniform)
    unsigned dummy = 0;
    unsigned st_iter = 0;
#ifdef RL_SPEC_TIMERS
    unsigned long long tms[6] = {0, 0, 0, 0, 0, 0}; bool ser_prev = false;
    unsigned long long tmr[6] = {0, 0, 0, 0, 0, 0}, e_lanes = 0, e_iters = 0, e_slow_iters = 0, tq;
    const unsigned long long wave_t0 = wall_clock64();
    unsigned long long it_t0 = 0, cyc_serial = 0, cyc_full = 0, n_serial = 0, n_full = 0, n_idle = 0; bool it_serial = false, it_traced = false;
#define RL_ST0 { tq = __builtin_readcyclecounter(); }
#define RL_ST1(K) { const unsigned long long t1 = __builtin_readcyclecounter(); tmr[K] += t1 - tq; if (ser_prev) tms[K] += t1 - tq; tq = t1; }
#else
#define RL_ST0
#define RL_ST1(K)
#endif

#define RL_SPEC_KERNARGS \
        const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr(); \
        asm volatile("" : "+s"(ka)); \
        constexpr size_t sc_off = (sizeof(RenderConst) + alignof(DeviceScene) - 1) / alignof(DeviceScene) * alignof(DeviceScene); \
        static_assert(sc_off == offsetof(PathKernargs, sc), "kernarg layout of k_stream_spec"); \
        const DeviceScene& sc = *(const DeviceScene*)(ka + sc_off); \
        static_assert(offsetof(SpecKernargs, sc) == offsetof(PathKernargs, sc), "kernarg layout of k_stream_spec"); \
        spcp = (const SpecConf*)(ka + offsetof(SpecKernargs, spec)); \
        const RenderConst& rc = *(const RenderConst*)ka;

    // Path::from_sensor + Camera::generate for the sample that starts at `rng` (raygen_chain_slot without the cursor logic)
    auto begin_sample_at = [&](const RenderConst& rc, const DeviceScene& sc, Rng rng, unsigned pxy_) {
        const float u = (float)(pxy_ & 0xffffu) + rng_next_f32(rng);
        const float v = (float)(pxy_ >> 16) + rng_next_f32(rng);
        nd = 2u;
        const bool expand = (!rc.has_max || 1u < rc.max_depth);
        if (!expand) { store_rng(ps, Q_R0, rng); PU(U_FLAGS) = ST_REGEN; return; }
        store3(ps, F_DX, camera_direction(sc, u, v));
        if (sc.medium.enabled) { PF(F_XI) = rng_next_f32(rng); nd++; }
        store_rng(ps, Q_R0, rng);
        PU(U_DEPTH) = 1u;
        PU(U_FLAGS) = ST_RAY | (PREV_SENSOR << ST_PREV_SHIFT) | ST_PDF_SA;
    };
    auto begin_sample = [&](const RenderConst& rc, const DeviceScene& sc, const Rng& rng) { begin_sample_at(rc, sc, rng, pxy); };
    // (leader) a round of helpers from the chain's head (off, st)
    auto dense_round = [&](unsigned off, const Rng& st) {
        ghead[0] = (unsigned)st.s0; ghead[1] = (unsigned)(st.s0 >> 32); ghead[2] = (unsigned)st.s1; ghead[3] = (unsigned)(st.s1 >> 32);
        ghead[4] = (unsigned)st.s2; ghead[5] = (unsigned)(st.s2 >> 32); ghead[6] = (unsigned)st.s3; ghead[7] = (unsigned)(st.s3 >> 32);
#ifdef RL_SPEC_TIMERS
        if (spc.stats) atomicAdd(&spc.stats[24], 1ull);
#endif
        ghead[GH_B] = off; ghead[GH_ROUND] = ghead[GH_ROUND] + 1u; ghead[GH_ACTIVE] = 1u; ghead[GH_LGL] = gl; ghead[GH_PXY] = pxy;
    };
    // (leader) the chain has reached the end of this lane's pixel
    auto finish_pixel = [&](unsigned off, const Rng& st) {
        res_off = off; store_res_st(st);
        resolved = true; mode = SM_IDLE;
        PU(U_FLAGS) = 0u;
        estL = (float)(off - (pl == 0u ? 0u : COLD_OF(threadIdx.x - S, K_RES_OFF)));      // (the pixel began where its predecessor ended)
        if (cnt >= 8u) { const float m = (float)sum_n / (float)cnt; estV = fmaxf(sum_n2 / (float)cnt - m * m, 0.0f) * (float)spp; }
        have_est = true;
    };
    // (leader) the pixel's chain leaves the tracks at (off, st) with i samples done: walk on, one lane, towards entry j of segment k's track
    auto slow_walk = [&](const RenderConst& rc, const DeviceScene& sc, unsigned off, const Rng& st, unsigned i, unsigned k, unsigned j) {
        res_i = i; res_k = k; res_j = j; cur_off = off;
        store_sample_state(rc, i, pix, st);
        if (H && cnt >= (spc.dense_frac > 0.0f ? 8u : 1u) && (float)(nmin_cnt & 0xffffu) >= spc.dense_frac * (float)cnt) {      // (dense_frac = 0, tests: every serial walk that has a draw count to go by)      // several samples at a time (the lanes start in section D2)
#ifdef RL_SPEC_TIMERS
            if (spc.stats) atomicAdd(&spc.stats[26], 1ull);
#endif
            mode = SM_IDLE; PU(U_FLAGS) = 0u;
            ghead[GH_C] = nmin_cnt >> 16;
            dense_round(off, st);
            return;
        }
#ifdef RL_SPEC_TIMERS
        if (spc.stats) atomicAdd(&spc.stats[27], 1ull);
#endif
        mode = SM_TRUTH;
        begin_sample(rc, sc, st);
    };
    // (leader) the chain stands on entry j of segment k's track with i samples of the pixel done: that track is the chain up to its link into the next
    // segment's track (or to its frontier), and so on
    auto follow = [&](const RenderConst& rc, const DeviceScene& sc, unsigned i, unsigned k, unsigned j) {
        for (;;) {
            const unsigned t = lt + k;
            const unsigned Mk = COLD_OF(t, K_M), li = COLD_OF(t, K_LINK_I);
            const bool lk = li != 0xffffffffu && j <= li;
            const unsigned lim = lk ? li : Mk;
            const unsigned take = min(spp - i, lim - j);
            COLD_OF(t, K_CP_I0) = i; COLD_OF(t, K_CP_J0) = j; COLD_OF(t, K_CP_K0) = take;
            i += take;
            if (i == spp) { finish_pixel(OFF_OF(t)[j + take], entry_state(t, j + take)); return; }
            if (lk) { j = COLD_OF(t, K_LINK_J); k++; continue; }
#ifdef RL_SPEC_TIMERS
            if (spc.stats) atomicAdd(&spc.stats[k + 1u < S ? 13 : 14], 1ull);
#endif
            slow_walk(rc, sc, OFF_OF(t)[Mk], entry_state(t, Mk), i, k + 1u, 0u);      // this track ends before the pixel does (and nothing links it on)
            return;
        }
    };

    for (;;) {
        RL_SPEC_KERNARGS
        st_iter++;
#ifdef RL_SPEC_TIMERS
        { const unsigned long long tn = wall_clock64();
          if (it_t0) { if (!it_traced) n_idle++; else if (it_serial) { cyc_serial += tn - it_t0; n_serial++; } else { cyc_full += tn - it_t0; n_full++; } }
          it_t0 = tn; }
#endif
        RL_ST0
        // ---- A. a sample of a walking lane has ended (or its walk is about to start): bookkeeping by mode
        bool fin_now = false;
        {
            const unsigned flags = PU(U_FLAGS);
            if (mode != SM_IDLE && (flags & ST_REGEN)) {
                const bool fresh = (flags & ST_FRESH) != 0u;
                const Rng rng = load_rng(ps, Q_R0);            // the sampler after the sample (or where the walk starts)
                if (mode == SM_PROBE) {
                    if (!fresh) { cnt++; sum_n += nd; sum_n2 += (float)nd * (float)nd; st_probe_inc; }
                    if (cnt >= spc.probe) { mode = SM_IDLE; PU(U_FLAGS) = 0u; }
                    else begin_sample(rc, sc, rng);
                } else if (mode == SM_WALK || mode == SM_EXTRA) {
                    if (!fresh) {
                        cnt++; sum_n += nd; sum_n2 += (float)nd * (float)nd; cur_off += nd; st_spec_inc;
                        const unsigned nm = nmin_cnt >> 16, ndc = min(nd, 0xffffu);
                        if (ndc < nm) nmin_cnt = (ndc << 16) | 1u; else if (ndc == nm && (nmin_cnt & 0xffffu) != 0xffffu) nmin_cnt++;
                    }
                    my_off[M] = cur_off;
                    my_st[2u * M] = make_ulonglong2(rng.s0, rng.s1); my_st[2u * M + 1u] = make_ulonglong2(rng.s2, rng.s3);
                    // past its own segment the lane walks on until it stands on an entry of the next lane's walk (same pixel): from there the two are one walk.
                    // (the neighbour writes its track while this lane reads it: entries below its M are complete; the loads bypass this CU's L1)
                    bool met = false;
                    if (sk + 1u < S && cur_off >= nb_lo) {
                        const unsigned* b = OFF_OF(threadIdx.x + 1u);
                        const unsigned Mb = COLD_OF(threadIdx.x + 1u, K_M);
                        unsigned lj = link_j, bv = 0u;
                        while (lj < Mb && (bv = __hip_atomic_load(b + lj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < cur_off) lj++;
                        link_j = lj;
                        if (lj < Mb && bv == cur_off) { link_i = M; met = true; }
                    }
                    if (mode == SM_WALK && !met && cur_off >= stop && sk + 1u == S && M + 1u < spc.cap && spc.extra) mode = SM_EXTRA;      // the window is covered: walk on while others still walk
                    if (met || M + 1u >= spc.cap || (mode == SM_WALK && cur_off >= stop)) { mode = SM_IDLE; PU(U_FLAGS) = 0u; }      // entry M is the frontier
                    else { M++; begin_sample(rc, sc, rng); }
                } else if (mode == SM_HELP) {      // a helper's sample is done: its draw count and the sampler after it, for the leader to pick up
                    store_res_st(rng);
                    COLD(K_DN) = nd; st_spec_inc;
                    mode = SM_IDLE; PU(U_FLAGS) = 0u;
                } else {   // SM_TRUTH: the pixel's leader walks where no track carries the chain
                    res_i++; cur_off += nd; st_slow_inc;
                    // the first entry at or beyond the chain in the tracks ahead (segment res_k, then the next ones)
                    for (;;) {
                        if (res_k >= S) break;
                        const unsigned t = lt + res_k, Mk = COLD_OF(t, K_M);
                        const unsigned* offs = OFF_OF(t);
                        while (res_j <= Mk && offs[res_j] < cur_off) res_j++;
                        if (res_j <= Mk) break;
                        res_k++; res_j = 0u;
                    }
                    if (res_i == spp) { finish_pixel(cur_off, rng); fin_now = true; }
                    else if (res_k < S && OFF_OF(lt + res_k)[res_j] == cur_off) { follow(rc, sc, res_i, res_k, res_j); fin_now = resolved; }
                    else { store_sample_state(rc, res_i, pix, rng); begin_sample(rc, sc, rng); }
                }
            }
        }
        if (H && ghead[GH_ACTIVE] && ghead[GH_LGL] == gl) {
            // ---- A2. (leader) take the helpers' samples in chain order for as long as the chain stands on them
            const unsigned hb = ghead[GH_B], hc = ghead[GH_C], hr = ghead[GH_ROUND], gbase = threadIdx.x - gl;
            for (;;) {
                const unsigned d = cur_off - hb, h = d / hc;
                const unsigned t = gbase + ((gl + h) & (G - 1u));
                const unsigned n = COLD_OF(t, K_DN);
                if (n == 0u) break;            // (lane h is on the chain — the loop only comes here for such a lane — and still tracing)
                const Rng e = load_res_st(t);
#ifdef RL_SPEC_TIMERS
                if (spc.stats) atomicAdd(&spc.stats[25], 1ull);
#endif
                res_i++; cur_off += n; st_slow_inc;
                for (;;) {
                    if (res_k >= S) break;
                    const unsigned tk = lt + res_k, Mk = COLD_OF(tk, K_M);
                    const unsigned* offs = OFF_OF(tk);
                    while (res_j <= Mk && offs[res_j] < cur_off) res_j++;
                    if (res_j <= Mk) break;
                    res_k++; res_j = 0u;
                }
                if (res_i == spp) { ghead[GH_ACTIVE] = 0u; finish_pixel(cur_off, e); fin_now = true; break; }
                if (res_k < S && OFF_OF(lt + res_k)[res_j] == cur_off) { ghead[GH_ACTIVE] = 0u; follow(rc, sc, res_i, res_k, res_j); fin_now = resolved; break; }
                store_sample_state(rc, res_i, pix, e);
                const unsigned d2 = cur_off - hb, h2 = d2 / hc;
                if (h2 * hc != d2 || h2 >= H || COLD_OF(gbase + ((gl + h2) & (G - 1u)), K_DR) != hr) { dense_round(cur_off, e); break; }       // off the lanes' offsets: the next round starts here
            }
        }
        if (group_any(fin_now)) own++;
        RL_ST1(0)

        // ---- B. group transitions, taken when no lane of the group is walking
        const bool group_idle = !group_any(mode != SM_IDLE && mode != SM_EXTRA);
        if (group_idle && mode == SM_EXTRA) {      // the group's windows are all walked: the extra walk stops where it stands (the sample in flight is dropped, its start stays the frontier)
            if (!(PU(U_FLAGS) & ST_REGEN) || (PU(U_FLAGS) & ST_FRESH)) M--;
            else { cur_off += nd; my_off[M] = cur_off; const Rng r2 = load_rng(ps, Q_R0); my_st[2u * M] = make_ulonglong2(r2.s0, r2.s1); my_st[2u * M + 1u] = make_ulonglong2(r2.s2, r2.s3); }
            mode = SM_IDLE; PU(U_FLAGS) = 0u;
        }
        if (group_idle && phase == SP_PROBE) {
            if (!planned) {
                // a new batch: this lane's pixel
                planned = true;
                const unsigned c = q0 + pl;
                valid = c < c_end;
                pxy = (bx + c % bw) | ((by + c / bw) << 16);
                triv = valid && ((triv_bits[(c >> 5) & 7u] >> (c & 31u)) & 1u) != 0u;
                resolved = false; copied = false; linked = false; cp_k0 = 0u; M = 0u; link_i = 0xffffffffu;
                cnt = 0u; sum_n = 0u; sum_n2 = 0.0f; nmin_cnt = 0xffff0000u;
                if (sk == 0u && valid && !triv && (!have_est || spc.probe_every) && spc.probe > 0u) {
                    // nothing to predict this pixel's length from: a short walk somewhere in the stream nobody else probes
                    Rng r = load_anc(); rng_advance(r, (pl << 10) + 512u);
                    store_rng(ps, Q_R0, r);
                    PU(U_FLAGS) = ST_REGEN | ST_FRESH;
                    mode = SM_PROBE;
                }
            }
        }
        const bool group_idle2 = !group_any(mode != SM_IDLE && mode != SM_EXTRA);
        if (group_idle2 && phase == SP_PROBE && planned) {
            // ---- the windows of the batch
            if (sk == 0u && valid && !triv && (!have_est || cnt > 0u)) {
                const float m = cnt ? (float)sum_n / (float)cnt : 16.0f;
                const float var = cnt ? fmaxf(sum_n2 / (float)cnt - m * m, 0.0f) : m * m;
                const float pL = m * (float)spp;
                const float pV = var * (float)spp * (1.0f + (float)spp / (float)(cnt ? cnt : 1u));      // spread of the pixel's length + error of this estimate of its mean
                // a pixel that has a length to go by (the pixel a batch earlier) keeps it unless the probe contradicts it: across a geometry edge the
                // pixel a row or two up is a different kind of pixel
                const float dl = pL - estL;
                if (!have_est || dl * dl > 9.0f * (pV + estV)) { estL = pL; estV = pV; }
                have_est = true;
            }
            cnt = 0u; sum_n = 0u; sum_n2 = 0.0f; nmin_cnt = 0xffff0000u;
            const float eL = reinterpret_cast<const float&>(COLD_OF(lt, K_EST_L)), eV = reinterpret_cast<const float&>(COLD_OF(lt, K_EST_V));      // the pixel's (its leader's)
            const float Lp = valid ? (triv ? 2.0f * (float)spp : eL) : 0.0f;
            const float Vp = (valid && !triv) ? 2.0f * eV : 0.0f;        // the length it is predicted from is itself one draw of the same spread
            float Ti = sk == 0u ? Lp : 0.0f, Si = sk == 0u ? Vp : 0.0f;      // sums over the pixels up to and including this lane's
            group_scan2(Ti, Si);
            const float T_ex = Ti - Lp, S_ex = Si - Vp;
            const bool any_walk = group_any(valid && !triv);
            // Two walks of a pixel meet after about as many samples as a sample takes draws (their sample starts are renewal processes of that density), so a
            // pixel of spp samples can only be threaded through a speculative track when its samples are short next to spp: a batch whose pixels take more
            // than spp / serial_ratio draws per sample on average (participating media, deep specular chains, few samples per pixel) is walked by its
            // leaders alone, pixel after pixel — the one-lane walk of k_stream_chain — instead of paying for tracks the chain would not land on.
            bool serial_batch = false;
            if (any_walk && spc.serial_ratio > 0.0f) {
                float sl = (sk == 0u && valid && !triv) ? Lp : 0.0f, sn = (sk == 0u && valid && !triv) ? 1.0f : 0.0f;
                group_sum2(sl, sn);
                serial_batch = sl * spc.serial_ratio > sn * (float)spp * (float)spp;       // mean draws per sample = sl / (sn spp)
            }
            if (any_walk && serial_batch) {
                if (valid && !triv) { my_off[0] = 0xffffffffu; M = 0u; }        // an empty track: an offset no chain reaches
                linked = true;
                own = 0u;
            } else
            if (!any_walk) {
                // every pixel of the batch looks past the scene: two draws per sample, the states are skip-aheads
                const unsigned n_valid = min(NP, c_end - q0);
                Rng r = load_anc(); rng_advance(r, 2u * spp * min(pl, n_valid));
                if (valid && sk == 0u) for (unsigned s = 0; s < spp; s++) { store_sample_state(rc, s, pix, r); rng_next_u64(r); rng_next_u64(r); }
                if (sk == 0u) { store_res_st(r); res_off = 0u; }
                resolved = true; copied = true;
                own = NP;
            } else {
                if (valid && !triv) {
                    const float nbar = fmaxf(Lp / (float)spp, 1.0f);
                    const unsigned nn = (unsigned)(nbar + 0.5f);
                    const unsigned t_ex = (unsigned)fminf(T_ex + 0.5f, 4.0e9f);
                    // lead-in: two walks of a pixel whose samples nearly all take the same number of draws fall in with each other slowly (they only
                    // change their relative position when one of them takes an unusual sample), so such pixels — whose samples are cheap — get a longer one
                    float lead_s = (float)spc.lead;
                    if (spc.lead_var > 0.0f) lead_s = fminf(fmaxf(lead_s, lead_s * spc.lead_var * (float)spp / fmaxf(eV, 1.0e-3f)), (float)spc.lead_max);
                    const unsigned lead_d = (unsigned)fminf(lead_s * nbar, 1.0e9f);
                    unsigned margin = pl == 0u ? 0u : (unsigned)fminf(spc.ks * __builtin_sqrtf(S_ex), 1.0e9f) + lead_d;
                    margin = (margin + nn - 1u) / nn * nn;          // a pixel whose samples all take nn draws only meets the chain on its own residue
                    margin = min(margin, t_ex / nn * nn);
                    const unsigned lo = t_ex - margin;
                    const unsigned hi = max(lo, (unsigned)fminf(T_ex + Lp + spc.ke * __builtin_sqrtf(Si) + 0.5f, 4.0e9f));
                    // this lane's segment of [lo, hi): it walks on into the next one for a lead-in's length, so that the two walks can fall in with each other
                    const unsigned seg = ((hi - lo) / S + nn - 1u) / nn * nn;
                    const unsigned s_lo = lo + sk * seg;
                    nb_lo = lo + (sk + 1u) * seg;
                    stop = sk + 1u == S ? hi : min(hi, lo + (sk + 2u) * seg);        // (a walk that has not met the next lane's by the end of THAT lane's segment gives up)
                    link_j = 0u;
                    Rng r = load_anc(); rng_advance(r, s_lo);
                    store_rng(ps, Q_R0, r);
                    PU(U_FLAGS) = ST_REGEN | ST_FRESH;
                    cur_off = s_lo;
                    mode = SM_WALK;
                }
                own = 0u;
            }
            phase = (any_walk && !serial_batch) ? SP_WALK : SP_RESOLVE;
        } else if (group_idle2 && phase == SP_WALK) {
            phase = SP_RESOLVE;
        }
        // ---- the walks of the batch are done: where does each lane's walk first stand on an entry of the next lane's (same pixel)?  Lane-parallel;
        // the tracks were written by other lanes of this wave: their stores are made visible first.
        if (S > 1u && __ballot(phase == SP_RESOLVE && !linked && own < NP) != 0ull) {
            __threadfence();
            if (G > 64u) __syncthreads();          // (the tracks of a pixel's lanes are in one wave, but the threading below reads across waves)
            if (phase == SP_RESOLVE && !linked) {
                linked = true;
                if (valid && !triv && sk + 1u < S && link_i == 0xffffffffu) {
                    const unsigned* a = my_off; const unsigned* b = OFF_OF(threadIdx.x + 1u);
                    const unsigned Ma = M, Mb = COLD_OF(threadIdx.x + 1u, K_M);
                    unsigned i = 0u, j = 0u;
                    const unsigned b0 = b[0];
                    { unsigned lo_i = 0u, hi_i = Ma + 1u; while (lo_i < hi_i) { const unsigned mid = (lo_i + hi_i) >> 1; if (a[mid] < b0) lo_i = mid + 1u; else hi_i = mid; } i = lo_i; }
                    unsigned av = i <= Ma ? a[i] : 0u, bv = b0;
                    while (i <= Ma && j <= Mb) {
                        if (av == bv) { link_i = i; link_j = j; break; }
                        if (av < bv) { i++; if (i <= Ma) av = a[i]; } else { j++; if (j <= Mb) bv = b[j]; }
                    }
                }
            }
        }

        RL_ST1(1)
        // ---- C. thread the chain through the tracks: pixels whose chain never leaves the tracks are done back to back
        if (__ballot(phase == SP_RESOLVE && own < NP) != 0ull) {
            for (;;) {
                bool fin = false;
                if (phase == SP_RESOLVE && own < NP && sk == 0u && pl == own && mode == SM_IDLE && !resolved && !(H && ghead[GH_ACTIVE])) {
                    // where the chain stands: after the predecessor's pixel (its leader's parked res_off / res_st), or at the anchor
                    unsigned b_off = 0u; Rng b_st;
                    if (own == 0u) b_st = load_anc();
                    else { b_off = COLD_OF(threadIdx.x - S, K_RES_OFF); b_st = load_res_st(threadIdx.x - S); }
                    if (!valid) { res_off = b_off; store_res_st(b_st); resolved = true; copied = true; fin = true; }       // past the block's end: hand the chain on
                    else if (triv) {
                        my_st[0] = make_ulonglong2(b_st.s0, b_st.s1); my_st[1] = make_ulonglong2(b_st.s2, b_st.s3);      // the copy-out steps through the pixel from here
                        Rng r = b_st; rng_advance(r, 2u * spp);
                        res_off = b_off + 2u * spp; store_res_st(r); resolved = true; fin = true;
                    } else {
                        // the first of the pixel's tracks that reaches the start (a track runs on past its own segment until it has fallen in with the next
                        // one, and the longer a walk has been going the likelier it is the chain); the next track too where the two overlap
                        unsigned k0 = 0u;
                        while (k0 < S && OFF_OF(lt + k0)[COLD_OF(lt + k0, K_M)] < b_off) k0++;
                        bool on_track = false; unsigned j0 = 0u;
                        for (unsigned k = k0; k < S && k < k0 + 2u && !on_track; k++) {
                            const unsigned* offs = OFF_OF(lt + k);
                            const unsigned Mk = COLD_OF(lt + k, K_M);
                            if (k > k0 && offs[0] > b_off) break;
                            unsigned lo = 0u, hi_e = Mk + 1u;            // entries 0 .. M
                            while (lo < hi_e) { const unsigned mid = (lo + hi_e) >> 1; if (offs[mid] < b_off) lo = mid + 1u; else hi_e = mid; }
                            if (k == k0) j0 = lo;
                            if (lo <= Mk && offs[lo] == b_off) { on_track = true; k0 = k; j0 = lo; }
                        }
                        if (on_track) { follow(rc, sc, 0u, k0, j0); fin = resolved; }
                        else {
#ifdef RL_SPEC_TIMERS
                            if (spc.stats) atomicAdd(&spc.stats[12], 1ull);
#endif
                            slow_walk(rc, sc, b_off, b_st, 0u, k0, j0);      // not on a track: walk from the true start until one is met (or the pixel ends)
                        }
                    }
                }
                const bool gfin = group_any(fin);
                if (gfin) own++;
                if (G <= 64u ? __ballot(fin) == 0ull : !gfin) break;
            }
        }

        RL_ST1(2)
        // ---- D. the batch is threaded: copy the track entries the chain ran over into sample_states, then the next batch
        if (__ballot(phase == SP_RESOLVE && own >= NP) != 0ull) {
            if (phase == SP_RESOLVE && own >= NP) {
                if (valid && !copied) {
                    if (triv) {
                        if (sk == 0u) {
                            Rng r = entry_state(threadIdx.x, 0u);
                            for (unsigned s = 0; s < spp; s++) { store_sample_state(rc, s, pix, r); rng_next_u64(r); rng_next_u64(r); }
                        }
                    } else {
                        const unsigned i0 = cp_i0, j0 = cp_j0, n = cp_k0;
                        for (unsigned t = 0; t < n; t++) {
                            const ulonglong2 a = my_st[2u * (j0 + t)], b = my_st[2u * (j0 + t) + 1u];
                            ulonglong2* q = reinterpret_cast<ulonglong2*>(rc.sample_states + 4 * ((size_t)(i0 + t) * rc.n_state_pixels + pix));
                            q[0] = a; q[1] = b;
                        }
                    }
                    copied = true;
                }
                const Rng last = load_res_st(threadIdx.x - gl + (NP - 1u) * S);      // the chain after the group's last pixel = the next batch's anchor
                store_anc(last);
                q0 += NP;
                planned = false;
                own = 0u;
                if (q0 >= c_end) {
                    if (gl == 0u) { unsigned long long* q = rc.chain_states + 4 * (size_t)item; q[0] = last.s0; q[1] = last.s1; q[2] = last.s2; q[3] = last.s3; }   // the next chunk resumes here
                    // every sample state of the block is recorded (by lanes of this wave: the host only overlaps with groups of one wave): the evaluation pass may start on it
                    if (rc.queue && gl == 0u) queue_push(rc, item);
                    phase = SP_DONE;
                } else phase = SP_PROBE;
            }
        }
        // ---- D2. helpers of a serial walk: lanes of a group whose leader runs one start on the round's sample h (h c draws on from its head); a round that is over stops them
        if (H) {
            const unsigned act = ghead[GH_ACTIVE], hr = ghead[GH_ROUND];
            if (mode == SM_HELP && (!act || COLD(K_DR) != hr)) { mode = SM_IDLE; PU(U_FLAGS) = 0u; }
            if (act && mode == SM_IDLE && COLD(K_DR) != hr) {
                const unsigned h = (gl - ghead[GH_LGL]) & (G - 1u);
                if (h < H) {
                    Rng r;
                    r.s0 = (unsigned long long)ghead[0] | ((unsigned long long)ghead[1] << 32); r.s1 = (unsigned long long)ghead[2] | ((unsigned long long)ghead[3] << 32);
                    r.s2 = (unsigned long long)ghead[4] | ((unsigned long long)ghead[5] << 32); r.s3 = (unsigned long long)ghead[6] | ((unsigned long long)ghead[7] << 32);
                    for (unsigned k = h * ghead[GH_C]; k > 0u; k--) rng_next_u64(r);
                    COLD(K_DR) = hr; COLD(K_DN) = 0u;
                    mode = SM_HELP;
                    begin_sample_at(rc, sc, r, ghead[GH_PXY]);
                }
            }
        }
        RL_ST1(3)
        if (__ballot(phase != SP_DONE) == 0ull) break;

        // ---- E. one vertex of every walking lane
#ifdef RL_SPEC_TIMERS
        { const unsigned long long rm = __ballot((PU(U_FLAGS) & ST_RAY) != 0u); it_traced = rm != 0ull; it_serial = __ballot(mode == SM_WALK || mode == SM_PROBE) == 0ull; if (rm) { e_iters++; e_lanes += __popcll(rm); if (it_serial) e_slow_iters++; } }
#endif
        if (PU(U_FLAGS) & ST_RAY) {
            extend_slot(sc, recs, stack, ps);
            shade_slot<MAT, MEDIUM, LIGHTS_AREA_ONLY, true>(rc, sc, ps, PU(U_FLAGS), dummy, nd, dummy, dummy);
        }
#ifdef RL_SPEC_TIMERS
        RL_ST1(4)
        ser_prev = it_serial && it_traced;
#endif
    }
#undef RL_SPEC_KERNARGS
#ifdef RL_SPEC_TIMERS
    if (spc.stats && lane == 0u) { { unsigned long long* w = spc.stats + 32 + 8 * (tid >> 6); w[0] = wave_t0; w[1] = wall_clock64(); w[2] = n_full; w[3] = n_serial; w[4] = cyc_full; w[5] = cyc_serial; w[6] = n_idle; w[7] = gstat[1]; }
        for (int k = 0; k < 5; k++) { atomicAdd(&spc.stats[4 + k], tmr[k]); atomicAdd(&spc.stats[16 + k], tms[k]); } atomicAdd(&spc.stats[9], e_lanes); atomicAdd(&spc.stats[10], e_iters); atomicAdd(&spc.stats[11], e_slow_iters); }
#endif
    if (spc.stats) {
        if (G > 64u) __syncthreads();
        if (gl == 0u) { atomicAdd(&spc.stats[0], (unsigned long long)gstat[0]); atomicAdd(&spc.stats[1], (unsigned long long)gstat[1]); atomicAdd(&spc.stats[2], (unsigned long long)gstat[2]); }
        unsigned it = st_iter;
        for (int off = 32; off > 0; off >>= 1) it += __shfl_down(it, off, 64);
        if (lane == 0u) atomicAdd(&spc.stats[3], (unsigned long long)it);
    }
#undef my_off
#undef spc
#undef pix
#undef ghead
#undef gstat
#undef st_spec_inc
#undef st_slow_inc
#undef st_probe_inc
#undef my_st
#undef OFF_OF
#undef ST_OF
#undef COLD
#undef COLD_OF
}

template <bool LDS_SCENE, int MAT>
static void launch_spec_mat(bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const SpecConf& spc) {
    if (medium) hipLaunchKernelGGL((k_stream_spec<MAT, true, LDS_SCENE, 0>), grid, block, lds_bytes, st, rc, ds, stc, spc);
    else hipLaunchKernelGGL((k_stream_spec<MAT, false, LDS_SCENE, 0>), grid, block, lds_bytes, st, rc, ds, stc, spc);
}
template <bool LDS_SCENE>
static void launch_spec_impl(int mat, bool medium, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, const RenderConst& rc, const DeviceScene& ds, const StackConf& stc, const SpecConf& spc) {
    switch (mat) {
        case BSDF_DIFFUSE: launch_spec_mat<LDS_SCENE, BSDF_DIFFUSE>(medium, grid, block, lds_bytes, st, rc, ds, stc, spc); break;
        case BSDF_PHONG: launch_spec_mat<LDS_SCENE, BSDF_PHONG>(medium, grid, block, lds_bytes, st, rc, ds, stc, spc); break;
        case BSDF_METAL: launch_spec_mat<LDS_SCENE, BSDF_METAL>(medium, grid, block, lds_bytes, st, rc, ds, stc, spc); break;
        case BSDF_GLASS: launch_spec_mat<LDS_SCENE, BSDF_GLASS>(medium, grid, block, lds_bytes, st, rc, ds, stc, spc); break;
        case -1: launch_spec_mat<LDS_SCENE, -1>(medium, grid, block, lds_bytes, st, rc, ds, stc, spc); break;
        default: launch_spec_mat<LDS_SCENE, BSDF_SUBSTRATE>(medium, grid, block, lds_bytes, st, rc, ds, stc, spc); break;
    }
}

}  // namespace rl
