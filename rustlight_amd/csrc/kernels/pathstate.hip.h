// pathstate.hip.h — path-state pool (SoA fields, flags), the three state accessors the stage functions are written against,
// render constants, block-local statistics / stream compaction helpers.
// Part of the single translation unit wavefront.hip (included once, after devmath / shading / trace).
#pragma once

namespace rl {

// ------------------------------------------------------------------------------------------
// path-state pool (SoA): field f of slot i is at base[f * P + i]
enum FField {
    F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ,         // extension ray (origin doubles as the NEE origin)
    F_T, F_U, F_V,                               // hit record (with U_PRIM)
    F_BR, F_BG, F_BB,                            // beta: throughput of evaluate()'s edge products
    F_TR, F_TG, F_TB,                            // thr: generate()'s Russian-roulette throughput
    F_LR, F_LG, F_LB,                            // radiance of the current sample
    F_AR, F_AG, F_AB,                            // pixel accumulator (sum over samples, in sample order)
    F_WR, F_WG, F_WB,                            // weight of the edge being traced (BSDF / phase weight)
    F_RR, F_PDF,                                 // its rr_weight and directional pdf
    F_SX, F_SY, F_SZ,                            // NEE target point on the light
    F_CR, F_CG, F_CB,                            // NEE contribution, already times beta and MIS weight
    F_XI,                                        // medium distance-sampling random number of the edge
    F_COUNT
};
enum UField { U_FLAGS, U_DEPTH, U_PRIM, U_ITEM, U_CURSOR, U_SAMPLE, U_COUNT };
enum QField { Q_R0, Q_R1, Q_R2, Q_R3, Q_I0, Q_I1, Q_I2, Q_I3, Q_COUNT };

enum : unsigned {
    ST_FINISHED = 1u,      // slot has no work left
    ST_REGEN = 2u,         // sample ended: raygen must fold it and start the next one
    ST_FRESH = 4u,         // no sample has run in this slot yet
    ST_RAY = 8u,           // extension ray valid
    ST_SHADOW = 16u,       // NEE shadow ray valid
    ST_PREV_SHIFT = 5u,    // 2 bits: kind of the vertex the traced edge leaves
    ST_PDF_SA = 128u,      // edge pdf is PDF::SolidAngle (else Discrete)
    ST_ZEROED = 256u,      // single_scattering: a surface vertex has been passed (path.rs:122-124)
};
enum : unsigned { PREV_SENSOR = 0u, PREV_SURFACE = 1u, PREV_SURFACE_SMOOTH = 2u, PREV_VOLUME = 3u };

#ifndef RL_FUSED_WAVES
#define RL_FUSED_WAVES 4   // waves per SIMD requested for the persistent fused kernel on LDS-staged scenes (128 VGPRs; 3 / 4 / 5 / 6: 70.3 / 60.3 / 67.8 / 76.5 ms on cbox)
#endif
#ifndef RL_FUSED_WAVES_STREAMING
#define RL_FUSED_WAVES_STREAMING 6   // scenes that stream their BVH are latency-bound: 85 VGPRs (508 k triangles, 32 spp, 4 / 5 / 6 / 8 waves: 127.3 / 118.4 / 114.8 / 138.5 ms)
#endif
static constexpr unsigned kDepthCap = 2048u;   // same cut as the oracle (NaN-throughput paths never die)

struct Pool {
    float* f;
    unsigned* u;
    unsigned long long* q;
    unsigned P;
};

struct Counters {
    unsigned int active;        // slots that still own work
    unsigned int next_item;     // work-item dispenser
    unsigned int pad[2];
};

static constexpr int kLdsStackLevels = 12;   // stack levels kept in LDS per lane
static constexpr size_t kSampleBufBudget = (size_t)16 << 30;   // bytes of HBM the per-sample parking buffer may take
// traversal stack configuration (see TravStack)
struct StackConf { int lds_levels; int* overflow; size_t overflow_stride; int pre_group; };   // pre_group: k_stream_chain on tiny LDS scenes — lanes per chain that precompute its ray's records (0: off; trace.hip.h: precompute_records)

#ifndef RL_SPEC_LDS_LEVELS_STREAMING
#define RL_SPEC_LDS_LEVELS_STREAMING 4
#endif
static constexpr int kSpecLdsLevelsStreaming = RL_SPEC_LDS_LEVELS_STREAMING;      // k_stream_spec on scenes that stream their BVH: traversal-stack levels in LDS (spec.hip.h)
static constexpr int kSpecColdWords = 30;
static constexpr int kSpecHelperWords = 2;       // ... of which the helpers' (K_DN, K_DR): not parked on scenes that stream their BVH (no helpers there)
static constexpr size_t kSpecGroupLdsBytes = (256 / 16) * 32 + 64 + (256 / 16) * 64 + (256 / 16) * 16;      // per workgroup: the groups' anchors, the scan scratch, the groups' serial-walk heads, their counters
// k_stream_spec (spec.hip.h): launch configuration and scratch of the speculative first pass of reference-order streams
struct SpecConf {
    unsigned group;              // lanes per 16x16 block: 16, 32 or 64 (a batch = `group` consecutive pixels of the block, one per lane)
    unsigned sub;                // lanes per pixel: 1, 2, 4 or 8 (the pixel's window is walked in `sub` segments); group / sub pixels per batch
    unsigned cap;                // entries a lane's track can hold
    unsigned probe;              // samples of the estimate probe (lanes that have no resolved pixel to go by: the first batch)
    unsigned lead;               // samples of lead-in before a window (a track needs a few samples to fall in with the chain)
    unsigned lead_max; float lead_var;   // pixels whose draws per sample vary less than lead_var get lead x lead_var / variance samples of it, at most lead_max
    unsigned extra;              // != 0: a pixel's last lane walks on past its window while other lanes of the group still walk (it would idle; fewer serial samples past the track's end)
    unsigned probe_every;        // != 0: every batch probes (a pixel keeps the length of the pixel a batch earlier unless the probe contradicts it)
    unsigned dense; float dense_frac;    // serial walks of a pixel whose samples nearly all take the same number of draws c (at least dense_frac of its walked samples) are taken `dense` samples at a time: the group's idle
                                 // lanes evaluate the samples that start c, 2c, ... draws on (0: off)
    float serial_ratio;          // a batch whose samples take more than spp / serial_ratio draws on average is walked serially (0: never)
    float ks, ke;                // window margins in standard deviations of the predicted start / end offset
    unsigned* trk_off;           // [thread][cap] stream offsets of the samples a lane walked, relative to the batch's anchor, ascending
    ulonglong2* trk_st;          // [thread][cap][2] the sampler state at each of them
    const unsigned* trivial;     // [owned block][8] bit c: every camera ray of block cursor c misses the scene (two draws per sample)
    unsigned long long* stats;   // dev / bench: [0] spec samples walked, [1] slow samples, [2] probe samples, [3] wave loop iterations
};

template <bool LDS_ONLY = false>
RL_DEV TravStackT<LDS_ONLY> make_stack(const StackConf& sc_, unsigned* lds_after_list, size_t global_thread) {
    TravStackT<LDS_ONLY> st;
    st.lds = (typename TravStackT<LDS_ONLY>::LdsPtr)(lds_after_list) + threadIdx.x;
    st.lds_levels = sc_.lds_levels;
    st.glob = sc_.overflow ? (typename TravStackT<LDS_ONLY>::GlobPtr)(sc_.overflow) + global_thread : nullptr;
    st.glob_stride = sc_.overflow_stride;
    return st;
}

struct RenderConst {
    // IntegratorPathTracing fields (explicit/path.rs:14-20)
    unsigned spp;
    int has_min, has_max, has_rr;
    unsigned min_depth, max_depth, rr_depth;
    int strategy, single_scattering;
    int stream_mode, seed_variant;
    float inv_spp;
    // image / work decomposition
    unsigned W, H, nby;
    unsigned n_items;
    unsigned item_shift;                // persistent kernel: item i starts on thread i << item_shift (sparse item sets are spread over more waves)
    unsigned split;                     // per-sample mode: lanes per pixel (sample s of a pixel runs on lane s % split)
    float* sample_buf;                  // split > 1: [spp][n_items / split][3] per-sample radiance, folded in order by k_fold_samples
    const unsigned* owned_blocks;       // block ids of this shard, in creation order
    const unsigned* block_item_base;    // per owned block: first pixel item (per-sample mode)
    unsigned n_owned;
    const unsigned long long* block_seeds;   // one per block of the whole image
    unsigned long long* item_seed;      // per pixel item (per-sample mode)
    unsigned* item_pixel;               // per pixel item: y * W + x
    float* out;                         // W*H*3 framebuffer
    Counters* counters;
    unsigned long long* partials;       // [max grid blocks][STAT_COUNT] statistics rows
    // reference-order streams in two passes (chain.hip.h): the block sampler's state at the start of every camera sample,
    // [sample][chunk pixel][4] u64 — written by k_stream_chain, read by the per-sample kernel (stream_mode = kStreamGivenStates)
    unsigned long long* sample_states;
    unsigned n_state_pixels;            // pixels of this chunk (second index of sample_states)
    unsigned cursor_begin, cursor_end;  // k_stream_chain: the block cursors [begin, end) this chunk covers
    unsigned long long* chain_states;   // [owned block][4]: where a block's stream stands between two chunks
    // the evaluation pass overlapped with the chain pass (round 5).  NOTHING ON THE DEVICE WAITS: the chain kernels flag every block whose sample states are all recorded
    // (a word per owned block in mapped host memory), the host thread inside rl_render_path collects the flagged blocks and launches k_path_fused<.., QUEUE = true> over
    // explicit lists of complete blocks on other streams while the chain pass still runs.  (The first forms of this pass kept evaluation waves waiting on the device for
    // blocks to come — s_sleep + polling loads: about one render in a hundred the whole device then stood still, a 512-byte copy included, until the waiters gave up:
    // scratch/r5/stall_repro.py, profiles/NEGATIVES.md round 5.)
    unsigned* queue;                    // chain kernels: device word counting their workgroups that have started (null: no overlap);
    unsigned* done_flags;               // ... mapped host memory, [owned block]: `queue_seq` once the block's states are all recorded;
    unsigned* started_flag;             // ... mapped host word: the workgroup that starts LAST stores `queue_seq` (the host launches nothing beside the chain pass before every workgroup of it runs)
    unsigned queue_seq, chain_grid;     // this render's tag; workgroups of the chain kernel
    unsigned queue_mode;                // k_path_fused: != 0 = the QUEUE form
    const unsigned* q_list;             // QUEUE form: owned-block indices this launch renders, q_n of them; q_ctr: its item-claim counter
    unsigned q_n;
    unsigned* q_ctr;
};
// called by thread 0 of every workgroup of a chain kernel
RL_DEV void queue_workgroup_started(const RenderConst& rc) {
    if (atomicAdd(rc.queue, 1u) + 1u == rc.chain_grid) __hip_atomic_store(rc.started_flag, rc.queue_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// a block's chain is complete: its sample states (this wave's stores — all lanes': the counters the fence waits on are the wave's) are made visible, then the host is told
RL_DEV void queue_push(const RenderConst& rc, unsigned item) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(&rc.done_flags[item], rc.queue_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// internal third value of RenderConst::stream_mode (never accepted from a caller): per-pixel work items as in RL_STREAM_PER_SAMPLE, but every
// camera sample starts from the sampler state k_stream_chain recorded for it — the image and the counters of RL_STREAM_REFERENCE_ORDER
enum : int { kStreamGivenStates = 2 };

// k_path_fused / k_stream_chain take (RenderConst, DeviceScene, StackConf) by value and re-read the first two from the kernarg segment inside
// their loops: the segment lays the arguments out like this struct (each at its natural alignment, in order); the kernels static_assert their
// hand-computed offset against it, so a reordered or added argument fails to compile instead of reading garbage.
struct PathKernargs { RenderConst rc; DeviceScene sc; StackConf stc; };
struct SpecKernargs { RenderConst rc; DeviceScene sc; StackConf stc; SpecConf spec; };      // k_stream_spec

// Path-state accessors.  The stage functions below are written once against `ps.f/u/q(field)`:
//  * PoolState: the wavefront kernels — state lives in the HBM pool, one coalesced word per lane;
//  * RegState:  the persistent fused kernel — the same fields are plain locals (every index is a compile-time
//    constant, so the arrays are scalarised into VGPRs and untouched fields disappear).
struct PoolState {
    Pool pool; unsigned slot;
    RL_DEV float& f(int field) const { return pool.f[(size_t)field * pool.P + slot]; }
    RL_DEV unsigned& u(int field) const { return pool.u[(size_t)field * pool.P + slot]; }
    RL_DEV unsigned long long& q(int field) const { return pool.q[(size_t)field * pool.P + slot]; }
};
struct RegState {
    float fv[F_COUNT]; unsigned uv[U_COUNT]; unsigned long long qv[Q_COUNT];
    RL_DEV float& f(int field) { return fv[field]; }
    RL_DEV unsigned& u(int field) { return uv[field]; }
    RL_DEV unsigned long long& q(int field) { return qv[field]; }
};
// FusedState: RegState whose "cold" fields (touched once per camera sample, by raygen only) are parked in LDS
// ([field][thread] layout, conflict-free) instead of occupying VGPRs through the traversal and shading code.
struct FusedState {
    float fv[F_COUNT]; unsigned uv[U_COUNT]; unsigned long long qv[Q_COUNT];
    float* cold_f; unsigned* cold_u; unsigned long long* cold_q;     // already offset by threadIdx.x
    static constexpr int kColdF = 3, kColdU = 3, kColdQ = 4;
    RL_DEV float& f(int field) { return (field >= F_AR && field <= F_AB) ? cold_f[(field - F_AR) * 256] : fv[field]; }
    RL_DEV unsigned& u(int field) { return (field >= U_ITEM && field <= U_SAMPLE) ? cold_u[(field - U_ITEM) * 256] : uv[field]; }
    RL_DEV unsigned long long& q(int field) { return (field >= Q_I0) ? cold_q[(field - Q_I0) * 256] : qv[field]; }
};
static constexpr size_t kFusedColdBytes = 256 * (FusedState::kColdQ * 8 + FusedState::kColdF * 4 + FusedState::kColdU * 4);
#define PF(field) ps.f(field)
#define PU(field) ps.u(field)
#define PQ(field) ps.q(field)

template <class PS> RL_DEV V3 load3(PS& ps, int f0) { return mk3(PF(f0), PF(f0 + 1), PF(f0 + 2)); }
template <class PS> RL_DEV Col loadc(PS& ps, int f0) { return mkc(PF(f0), PF(f0 + 1), PF(f0 + 2)); }
template <class PS> RL_DEV void store3(PS& ps, int f0, V3 v) { PF(f0) = v.x; PF(f0 + 1) = v.y; PF(f0 + 2) = v.z; }
template <class PS> RL_DEV void storec(PS& ps, int f0, Col c) { PF(f0) = c.r; PF(f0 + 1) = c.g; PF(f0 + 2) = c.b; }
template <class PS> RL_DEV Rng load_rng(PS& ps, int q0) { Rng r; r.s0 = PQ(q0); r.s1 = PQ(q0 + 1); r.s2 = PQ(q0 + 2); r.s3 = PQ(q0 + 3); return r; }
template <class PS> RL_DEV void store_rng(PS& ps, int q0, const Rng& r) { PQ(q0) = r.s0; PQ(q0 + 1) = r.s1; PQ(q0 + 2) = r.s2; PQ(q0 + 3) = r.s3; }

// Statistics without atomics on shared words (a device-scope atomic on one word costs ~10 ns and
// serialises: MI355X_MICROARCH "fanin"): wave64 shuffle sum -> LDS per block -> one plain
// read-modify-write of this block's own row of `partials` (rows are private to a block index;
// launches on one stream are ordered).  The host sums the rows after the render.
enum { STAT_SAMPLES, STAT_VERTICES, STAT_EXT_RAYS, STAT_SHADOW_RAYS, STAT_DRAWS, STAT_COUNT = 8 };
RL_DEV unsigned wave_sum(unsigned v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
template <int N>
RL_DEV void block_stats(unsigned long long* partials, const int (&which)[N], const unsigned (&vals)[N]) {
    __shared__ unsigned s_acc[N];
    if (threadIdx.x < N) s_acc[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        unsigned v = wave_sum(vals[k]);
        if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&s_acc[k], v);
    }
    __syncthreads();
    if (threadIdx.x < N && s_acc[threadIdx.x]) partials[(size_t)blockIdx.x * STAT_COUNT + which[threadIdx.x]] += s_acc[threadIdx.x];
}

// Block-local stream compaction with wave64 ballot + prefix popcount (no global atomics): the threads of a
// workgroup whose slot satisfies `pred` are packed to the front, so the traversal / shading loops run on
// full waves and the remaining waves exit at once.  `list` = 256 + 4 words of LDS.  Returns the number of
// packed entries; thread `t < n` then works on slot `list[t]`.
RL_DEV unsigned block_compact(bool pred, unsigned slot, unsigned* list) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(pred);
    const unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
    unsigned* wave_cnt = list + 256;
    if (lane == 0u) wave_cnt[wave] = (unsigned)__popcll(mask);
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (unsigned w = 0; w < 4u; w++) { unsigned c = wave_cnt[w]; if (w < wave) base += c; total += c; }
    if (pred) list[base + rank] = slot;
    __syncthreads();
    return total;
}

RL_DEV void block_geometry(const RenderConst& rc, unsigned b, unsigned* bx, unsigned* by, unsigned* bw, unsigned* bh) {
    *bx = (b / rc.nby) * 16u;            // block index b = (ix/16) * ceil(H/16) + iy/16 (mod.rs:357-358)
    *by = (b % rc.nby) * 16u;
    *bw = min(16u, rc.W - *bx);
    *bh = min(16u, rc.H - *by);
}

}  // namespace rl
