// wavefront.hip — the gfx950 wavefront path tracer behind `IntegratorPathTracing::compute`
// (src/integrators/explicit/path.rs:186-238 -> compute_mc, src/integrators/mod.rs:403-450).
//
// One path "slot" per GPU lane; all per-path state lives in HBM as structure-of-arrays so that
// lane i of every kernel touches element i of every array (fully coalesced, 4/8-byte words).
// A slot owns one work item — a pixel (throughput stream mode) or a whole 16x16 block
// (reference-order stream mode) — and regenerates a fresh camera sample in place when its path
// ends, so every slot carries exactly one extension ray per iteration until its item is done.
//
// Per iteration (one HIP launch each, same stream):
//   k_raygen   persistent threads: finished samples are folded into the pixel accumulator in
//              sample order, the next (pixel, sample) is seeded and its camera ray generated
//              (Path::from_sensor + Camera::generate; paths/path.rs:56-73, camera.rs:81-91)
//   k_extend   BVH2 closest-hit traversal (Acceleration::trace; accel.rs:292-315)
//   k_shade<M> per-BSDF kernels: medium distance sampling, fill_intersection, emission + MIS of
//              the arriving edge, BSDF / phase sampling, Russian roulette, NEE light sampling with
//              its MIS weight (strategies/directional.rs, strategies/emitters.rs, path.rs:37-111)
//   k_shadow   any-hit traversal for the NEE shadow rays (Acceleration::visible; accel.rs:316-343),
//              adds the pre-weighted contribution to the path's radiance
// (k_shade_sorted when the scene mixes BSDF types: block-local stream compaction + material sort with
// wave64 ballot + prefix popcounts, then one packed pass per BSDF).  Radiance is accumulated front-to-back (DESIGN.md §Radiance order).
// k_path_fused runs the same four stages back to back in one persistent launch with the state in registers / LDS — the form every
// per-sample render takes by default (BSDF code specialised for single-BSDF scenes, a run-time switch otherwise);
// k_pixel_mc is the `ao` / `direct` form.  This file holds the kernels and the host driver behind the C-ABI
// (rl_context_*, rl_render_path / _ao / _direct, rl_trace_batch, rl_visible_batch); the device code it instantiates is in
//   pathstate.hip.h  pool layout, state accessors, block-local statistics / compaction
//   stages.hip.h     raygen_slot, extend_slot, shade_slot, shadow_slot
//   mc.hip.h         ao / direct pixel estimators
//   trace.hip.h      BVH2 traversal;   shading.hip.h  BSDFs, emitters, light tree, medium;   devmath.hip.h  f32 contract, RNG
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.hip.h"
#include "rngjump.h"
#include "fused.hip.h"     // RL_COOP_FETCH (LDS budget of the persistent kernel); the kernel template is not instantiated here
#include "../host/scene.h"
#include "wavefront.h"
#include "knobs.h"

namespace rl {

// ------------------------------------------------------------------------------------------
// per-sample stream mode: fork the block sampler once per pixel in the block's (iy, ix) loop order
// with the reference's own clone_box rule (samplers/independent.rs:18-22)
__global__ void k_seed_pixels(RenderConst rc) {
    unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= rc.n_owned) return;
    unsigned b = rc.owned_blocks[j];
    unsigned bx, by, bw, bh;
    block_geometry(rc, b, &bx, &by, &bw, &bh);
    Rng rng = rng_seed(rc.block_seeds[b], rc.seed_variant);
    unsigned base = rc.block_item_base[j];
    for (unsigned iy = 0; iy < bh; iy++)
        for (unsigned ix = 0; ix < bw; ix++) {
            unsigned k = base + iy * bw + ix;
            rc.item_seed[k] = rng_next_u64(rng);
            rc.item_pixel[k] = (by + iy) * rc.W + (bx + ix);
        }
}

// reference-order streams in two passes: pixel of every work item of the chunk (block cursors [cursor_begin, cursor_end) of each owned block)
__global__ void k_chunk_pixels(RenderConst rc) {
    unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= rc.n_owned) return;
    unsigned bx, by, bw, bh;
    block_geometry(rc, rc.owned_blocks[j], &bx, &by, &bw, &bh);
    const unsigned c_end = min(rc.cursor_end, bw * bh), base = rc.block_item_base[j];
    for (unsigned c = rc.cursor_begin; c < c_end; c++) rc.item_pixel[base + (c - rc.cursor_begin)] = (by + c / bw) * rc.W + (bx + c % bw);
}

__global__ void k_init(RenderConst rc, Pool pool) {
    unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= pool.P) return;
    PoolState ps{pool, slot};
    PU(U_ITEM) = slot;
    PU(U_CURSOR) = 0u;
    PU(U_SAMPLE) = 0u;
    PU(U_DEPTH) = 0u;
    PU(U_PRIM) = 0xffffffffu;
    PU(U_FLAGS) = slot < rc.n_items ? (ST_REGEN | ST_FRESH) : ST_FINISHED;
    storec(ps, F_AR, czero());
    storec(ps, F_LR, czero());
}

// k_raygen — persistent threads (grid-stride loop over the pool).
__global__ void __launch_bounds__(256) k_raygen(RenderConst rc, DeviceScene sc, Pool pool) {
    unsigned n_samples = 0, n_draws = 0;
    for (unsigned slot = blockIdx.x * blockDim.x + threadIdx.x; slot < pool.P; slot += gridDim.x * blockDim.x) {
        PoolState ps{pool, slot};
        raygen_slot<true>(rc, sc, ps, n_samples, n_draws);
    }
    { const int which[2] = {STAT_SAMPLES, STAT_DRAWS}; const unsigned vals[2] = {n_samples, n_draws}; block_stats<2>(rc.partials, which, vals); }
}

// k_fold_samples — sample-parallel pixels (split > 1): add the parked per-sample radiances of each pixel in sample
// order and scale by 1 / spp, i.e. exactly the accumulate / scale sequence of compute_mc (mod.rs:431-436).
__global__ void __launch_bounds__(256) k_fold_samples(RenderConst rc) {
    const unsigned n_pix = rc.n_items / rc.split;
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pix) return;
    Col acc = czero();
    for (unsigned s = 0; s < rc.spp; s++) {
        const float* src = rc.sample_buf + 3 * ((size_t)s * n_pix + p);
        acc = acc + mkc(src[0], src[1], src[2]);
    }
    const Col px = scale_unguarded(acc, rc.inv_spp);
    const size_t pix = rc.item_pixel[p];
    rc.out[3 * pix] = px.r; rc.out[3 * pix + 1] = px.g; rc.out[3 * pix + 2] = px.b;
}

#ifdef RL_TRAV_STATS
// dev-only: traversal statistics of k_extend, totals over the render: [0] rays, [1] node steps, [2] triangle tests,
// [3] sum over waves of 64 * (max node steps in the wave) = what the wave pays, [4] waves, [5] rays with zero steps
__device__ unsigned long long g_trav_stats[8];
#endif
// k_extend / k_shadow — traversal kernels.  Dynamic LDS = [staged scene][compaction list][per-lane stacks].
template <bool LDS_SCENE, bool SHADOW>
RL_DEV void trace_kernel_body(const DeviceScene& sc, const Pool& pool, const StackConf& stc) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* after_scene = LDS_SCENE ? smem + lds_scene_float4s(sc.n_nodes, sc.n_prims) : smem;
    unsigned* list = reinterpret_cast<unsigned*>(after_scene);
    unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStackT<LDS_SCENE> stack = make_stack<LDS_SCENE>(stc, list + 272, slot);
    const unsigned flags = slot < pool.P ? pool.u[(size_t)U_FLAGS * pool.P + slot] : 0u;
    const unsigned n_live = block_compact((flags & (SHADOW ? ST_SHADOW : ST_RAY)) != 0u, slot, list);
    if (n_live == 0u) return;          // whole tile idle (finished pixels): skip the scene staging too
    SceneRecs recs;
    if (LDS_SCENE) {
        stage_scene_lds(sc, smem, smem + lds_nodes_float4s(sc.n_nodes));
        recs.nodes = smem; recs.tris = smem + lds_nodes_float4s(sc.n_nodes);
    } else {
        recs.nodes = streamed_nodes<TravStackT<false>>(sc);
        recs.tris = reinterpret_cast<const float4*>(sc.tris);
    }
#ifdef RL_TRAV_STATS
    int dbg[2] = {-1, 0};
    if (threadIdx.x < n_live) {
        PoolState ps{pool, list[threadIdx.x]};
        if (SHADOW) shadow_slot(sc, recs, stack, ps);
        else extend_slot(sc, recs, stack, ps, dbg);
    }
    if (!SHADOW) {
        int mx = dbg[0];
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
        if (dbg[0] >= 0) { atomicAdd(&g_trav_stats[0], 1ull); atomicAdd(&g_trav_stats[1], (unsigned long long)dbg[0]); atomicAdd(&g_trav_stats[2], (unsigned long long)dbg[1]); if (dbg[0] == 0) atomicAdd(&g_trav_stats[5], 1ull); }
        if ((threadIdx.x & 63u) == 0u && mx >= 0) { atomicAdd(&g_trav_stats[3], 64ull * (unsigned long long)mx); atomicAdd(&g_trav_stats[4], 1ull); }
    }
#else
    if (threadIdx.x < n_live) {
        PoolState ps{pool, list[threadIdx.x]};
        if (SHADOW) shadow_slot(sc, recs, stack, ps);
        else extend_slot(sc, recs, stack, ps);
    }
#endif
}

template <bool LDS_SCENE>
__global__ void __launch_bounds__(256) k_extend(RenderConst rc, DeviceScene sc, Pool pool, StackConf stc) { trace_kernel_body<LDS_SCENE, false>(sc, pool, stc); }
template <bool LDS_SCENE>
__global__ void __launch_bounds__(256) k_shadow(RenderConst rc, DeviceScene sc, Pool pool, StackConf stc) { trace_kernel_body<LDS_SCENE, true>(sc, pool, stc); }

// ------------------------------------------------------------------------------------------
// operator-level kernels: batched Acceleration::trace / visible for the parity tests
__global__ void __launch_bounds__(256) k_trace_batch(DeviceScene sc, StackConf stc, unsigned n, const float* o, const float* d, float* t_out, float* u_out,
                                                     float* v_out, int* mesh_out, int* tri_out) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    recs.nodes = streamed_nodes<TravStackT<false>>(sc);
    recs.tris = reinterpret_cast<const float4*>(sc.tris);
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStack stack = make_stack(stc, reinterpret_cast<unsigned*>(smem), i);
    if (i >= n) return;
    V3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    traverse<false>(recs, sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                    ro, rd, kEps, kF32Max, hit, stack);
    t_out[i] = hit.t; u_out[i] = hit.u; v_out[i] = hit.v;
    if (hit.prim >= 0) { mesh_out[i] = sc.tris[hit.prim].mesh; tri_out[i] = sc.tris[hit.prim].tri; }
    else { mesh_out[i] = -1; tri_out[i] = -1; }
}

// test hook: the same batch through the two-level records (trace.hip.h: traverse2), whatever the build traverses by default
__global__ void __launch_bounds__(256) k_trace_batch_two_level(DeviceScene sc, StackConf stc, unsigned n, const float* o, const float* d, float* t_out, float* u_out,
                                                               float* v_out, int* mesh_out, int* tri_out, int* steps_out, int any_hit) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    recs.nodes = reinterpret_cast<const float4*>(sc.nodes2);
    recs.tris = reinterpret_cast<const float4*>(sc.tris);
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStack2 stack(make_stack(stc, reinterpret_cast<unsigned*>(smem), i));
    if (i >= n) return;
    V3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    const V3 lo = mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), hi = mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]);
    if (any_hit) { hit.t = t_out[i]; const bool f = traverse<true>(recs, sc.root, lo, hi, ro, rd, kEps, hit.t, hit, stack); t_out[i] = f ? 1.0f : 0.0f; steps_out[i] = hit.steps; return; }   // t_out in: the segment length
    traverse<false>(recs, sc.root, lo, hi, ro, rd, kEps, kF32Max, hit, stack);
    t_out[i] = hit.t; u_out[i] = hit.u; v_out[i] = hit.v; steps_out[i] = hit.steps;
    if (hit.prim >= 0) { mesh_out[i] = sc.tris[hit.prim].mesh; tri_out[i] = sc.tris[hit.prim].tri; }
    else { mesh_out[i] = -1; tri_out[i] = -1; }
}

__global__ void __launch_bounds__(256) k_visible_batch(DeviceScene sc, StackConf stc, unsigned n, const float* p0a, const float* p1a, unsigned char* out) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    SceneRecs recs;
    recs.nodes = streamed_nodes<TravStackT<false>>(sc);
    recs.tris = reinterpret_cast<const float4*>(sc.tris);
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const TravStack stack = make_stack(stc, reinterpret_cast<unsigned*>(smem), i);
    if (i >= n) return;
    V3 p0 = mk3(p0a[3 * i], p0a[3 * i + 1], p0a[3 * i + 2]), p1 = mk3(p1a[3 * i], p1a[3 * i + 1], p1a[3 * i + 2]);
    V3 d = p1 - p0;
    float len = length(d);
    d = d / len;
    float tfar = len * (1.0f - 0.00001f);
    Hit hit; hit.t = tfar; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    bool occluded = traverse<true>(recs, sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                                   p0, d, kEps, tfar, hit, stack);
    V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float te;
    bool root_hit = slab(mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]), p0, inv_d, kEps, tfar, &te);
    out[i] = (root_hit && !occluded) ? 1 : 0;
}

// device self-test of the numerics contract: IEEE divide / sqrt, denormals, no contraction
__global__ void k_numerics_probe(unsigned n, const float* a, const float* b, float* out_div, float* out_sqrt, float* out_mad, float* out_sin,
                                 float* out_cos, float* out_exp, float* out_log, float* out_pow, float* out_acos, float* out_atan2) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_div[i] = div_rn(a[i], b[i]);
    out_sqrt[i] = sqrt_rn(fabsf(a[i]));
    out_mad[i] = a[i] * b[i] + a[i];
    out_sin[i] = dm::sinf_det(a[i]);
    out_cos[i] = dm::cosf_det(a[i]);
    out_exp[i] = dm::expf_det(a[i]);
    out_log[i] = dm::logf_det(fabsf(a[i]));
    out_pow[i] = dm::powf_det(fabsf(a[i]), b[i]);
    out_acos[i] = dm::acosf_det(a[i] * 0.15f);
    out_atan2[i] = dm::atan2f_det(a[i], b[i]);
}

}  // namespace rl

// ==========================================================================================
// host side
// ==========================================================================================
using namespace rl;

static thread_local std::string g_last_error;
void rl_set_error(const std::string& s) { g_last_error = s; }
extern "C" const char* rl_last_error(void) { return g_last_error.c_str(); }

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            rl_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                           \
            (void)hipGetLastError();   /* the error is reported here; do not leave it for the next call */ \
            return RL_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

struct rl_context {
    int device = 0;
    hipStream_t stream = nullptr;
    DeviceScene ds{};
    std::vector<void*> allocs;
    uint32_t width = 0, height = 0;
    bool single_bsdf = true;
    int bsdf_type = 0;
    bool lds_scene = false;
    bool area_lights_only = false;   // every emitter is a mesh area light, no light tree: the fused kernel's NEE code is specialised (same results)
    size_t scene_lds_bytes = 0;
    size_t lds_limit = 64 * 1024;     // dynamic LDS a workgroup may ask for on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
    // render scratch (grown on demand)
    Pool pool{};
    size_t pool_capacity = 0;
    unsigned* d_owned = nullptr; size_t owned_capacity = 0;
    unsigned* d_item_base = nullptr; size_t item_base_capacity = 0;
    unsigned long long* d_block_seeds = nullptr; size_t seeds_capacity = 0;
    unsigned long long* d_item_seed = nullptr; size_t item_capacity = 0;
    unsigned* d_item_pixel = nullptr; size_t item_pixel_capacity = 0;
    unsigned* d_queues = nullptr; unsigned* d_qcounts = nullptr; size_t queue_capacity = 0;
    float* d_out = nullptr; size_t out_capacity = 0;
    Counters* d_counters = nullptr;
    Counters* h_counters = nullptr;   // pinned
    unsigned long long* d_partials = nullptr; size_t partials_capacity = 0;
    int* d_overflow = nullptr; size_t overflow_capacity = 0;
    int* d_overflow2 = nullptr; size_t overflow2_capacity = 0;      // the overlapped evaluation pass's own overflow levels (it runs beside the chain pass, which uses d_overflow)
    float* d_sample_buf = nullptr; size_t sample_buf_capacity = 0;   // sample-parallel pixels: [spp][pixel item][3]
    unsigned long long* d_sample_states = nullptr; size_t sample_states_capacity = 0;   // reference-order streams, two passes: [spp][chunk pixel][4]
    unsigned long long* d_chain_states = nullptr; size_t chain_states_capacity = 0;     // [owned block][4]
    // k_stream_spec (spec.hip.h): per-lane tracks of the speculative first pass, the trivial-pixel masks and its counters
    unsigned* d_trk_off = nullptr; size_t trk_off_capacity = 0;
    ulonglong2* d_trk_st = nullptr; size_t trk_st_capacity = 0;
    unsigned* d_trivial = nullptr; size_t trivial_capacity = 0;
    uint64_t trivial_key = ~0ull;         // (shard index, shard count, sensor expanded?) the masks on the device were computed for
    unsigned long long* d_spec_stats = nullptr; size_t spec_stats_capacity = 0;
    std::vector<hipEvent_t> events;
    // the evaluation pass overlapped with the chain pass (reference-order streams): its own low-priority stream, the completion queue, ordering events
    static constexpr int kEvalStreams = 4;      // the evaluation launches beside the chain pass go round these (a launch lasts as long as its slowest pixel: several may have to be in flight)
    hipStream_t stream2 = nullptr; hipStream_t eval_streams[kEvalStreams] = {};
    unsigned* d_queue = nullptr; size_t done_queue_capacity = 0;       // device: [0] chain workgroups started, [16 + k] item-claim counter of the k-th evaluation launch, then the block lists
    hipEvent_t ev_chain_done = nullptr;
    unsigned* h_flags = nullptr; unsigned* d_flags = nullptr; size_t flags_capacity = 0;      // pinned, mapped: [0] the chain kernel's "every workgroup runs" word, [16 + j] block j's chain is complete
    unsigned* h_list = nullptr; size_t list_capacity = 0;              // pinned: the block lists of the evaluation launches (staging of the copies to the device)
    unsigned queue_seq = 0;
    BvhBuild bvh_dump;                // kept for rl_debug_bvh and for the two-level records (built on first use)
    Knobs knobs;                      // execution options: the environment as rl_context_create found it, then rl_context_set_option (knobs.h)
};

template <typename T>
static int upload(rl_context* ctx, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, bytes));
    ctx->allocs.push_back(p);
    if (!v.empty()) HIP_OK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T*>(p);
    return RL_OK;
}

// BvhNode2 records of the context's BVH, built and uploaded on first use
static int ensure_two_level(rl_context* ctx) {
    if (ctx->ds.nodes2) return RL_OK;
    std::vector<BvhNode2> n2;
    two_level_nodes(ctx->bvh_dump, &n2);
    return upload(ctx, n2, &ctx->ds.nodes2);
}

extern "C" int rl_device_count(int* count) {
    if (!count) return RL_ERR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { *count = 0; return RL_ERR_NO_DEVICE; }
    *count = n;
    return RL_OK;
}

extern "C" int rl_context_create(const rl_scene* scene, int device, rl_context** out) {
    if (!scene || !out) return RL_ERR_INVALID_ARGUMENT;
    if (!scene->has_camera) return RL_ERR_INVALID_ARGUMENT;
    if (!scene->emitters_built) return RL_ERR_NOT_BUILT;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        rl_set_error("no HIP device available (hipGetDeviceCount); the MI355X path has no CPU fallback");
        return RL_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n_dev) { rl_set_error("device ordinal out of range"); return RL_ERR_NO_DEVICE; }
    HIP_OK(hipSetDevice(device));
    rl_context* ctx = new rl_context();
    ctx->device = device;
    ctx->width = scene->width; ctx->height = scene->height;
    ctx->knobs.from_environment();       // the ONLY look at the environment: renders read the context's table (knobs.h)
    int rc = RL_OK;
    do {
        {   // the context's stream takes the highest priority the device offers, the overlapped evaluation pass's the lowest: where both have workgroups to place, the chain pass goes first
            int prio_least = 0, prio_greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = prio_greatest = 0; }
            if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) { rl_set_error("hipStreamCreate failed"); rc = RL_ERR_HIP; break; }
            if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_least) != hipSuccess) { (void)hipGetLastError(); ctx->stream2 = nullptr; }      // (no second stream: no overlap)
            if (ctx->stream2 && hipEventCreateWithFlags(&ctx->ev_chain_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); hipStreamDestroy(ctx->stream2); ctx->stream2 = nullptr; }
            if (ctx->stream2) {
                ctx->eval_streams[0] = ctx->stream2;
                for (int k = 1; k < rl_context::kEvalStreams; k++)
                    if (hipStreamCreateWithPriority(&ctx->eval_streams[k], hipStreamNonBlocking, prio_least) != hipSuccess) { (void)hipGetLastError(); ctx->eval_streams[k] = ctx->stream2; }      // (fewer streams: launches share them)
            }
        }
        BvhBuild& bvh = ctx->bvh_dump;
        build_bvh(*scene, &bvh);                  // BVHAccel::new — untimed (mod.rs:280)
        FlatScene flat;
        flatten_scene(*scene, &flat);
        DeviceScene& ds = ctx->ds;
        if ((rc = upload(ctx, bvh.nodes, &ds.nodes)) != RL_OK) break;
        if ((rc = upload(ctx, bvh.tris, &ds.tris)) != RL_OK) break;
        for (int i = 0; i < 3; i++) { ds.root_min[i] = bvh.root_min[i]; ds.root_max[i] = bvh.root_max[i]; }
        ds.root = bvh.root;
        ds.n_nodes = (uint32_t)bvh.nodes.size();
        ds.n_prims = (uint32_t)bvh.tris.size();
        ds.stack_depth = bvh.stack_depth;
        ds.nodes4 = nullptr; ds.root4 = RL_CHILD_NONE; ds.stack_depth4 = 0;
        ds.nodes_t = nullptr; ds.root_t = RL_CHILD_NONE;
        ds.nodes2 = nullptr;
        // the two-level records (traverse2; 128 B per node, twice the node array) are read by kernels built with RL_TWO_LEVEL / RL_LDS_TWO_LEVEL only — both off in
        // the shipped build (profiles/NEGATIVES.md round 5) — and by the test hook rl_debug_trace_batch_two_level, which builds them on its first call
        if ((RL_TWO_LEVEL || RL_LDS_TWO_LEVEL) && (rc = ensure_two_level(ctx)) != RL_OK) break;
        if ((rc = upload(ctx, flat.tri_indices, &ds.tri_indices)) != RL_OK) break;
        if ((rc = upload(ctx, flat.positions, &ds.positions)) != RL_OK) break;
        if ((rc = upload(ctx, flat.normals, &ds.normals)) != RL_OK) break;
        if ((rc = upload(ctx, flat.uvs, &ds.uvs)) != RL_OK) break;
        // `-x ats`: every emissive mesh learns where its triangles' leaves are listed; the tree itself is uploaded below
        for (MeshRecord& mr : flat.meshes) mr.ats_base = 0xffffffffu;
        std::vector<int32_t> ats_light_mesh;
        if (scene->ats_root >= 0) {
            for (size_t e = 0; e < scene->emitters.size(); e++) flat.meshes[scene->emitters[e].mesh].ats_base = scene->ats_emitter_base[e];
            for (int32_t e : scene->ats_light_emitter) ats_light_mesh.push_back(scene->emitters[e].mesh);
        }
        if ((rc = upload(ctx, flat.meshes, &ds.meshes)) != RL_OK) break;
        ds.ats_root = scene->ats_root;
        if ((rc = upload(ctx, scene->ats_nodes, &ds.ats_nodes)) != RL_OK) break;
        if ((rc = upload(ctx, ats_light_mesh, &ds.ats_light_mesh)) != RL_OK) break;
        if ((rc = upload(ctx, scene->ats_light_prim, &ds.ats_light_prim)) != RL_OK) break;
        if ((rc = upload(ctx, scene->ats_leaf_of, &ds.ats_leaf_of)) != RL_OK) break;
        if ((rc = upload(ctx, flat.materials, &ds.materials)) != RL_OK) break;
        if ((rc = upload(ctx, flat.bitmaps, &ds.bitmaps)) != RL_OK) break;
        if ((rc = upload(ctx, flat.bitmap_texels, &ds.bitmap_texels)) != RL_OK) break;
        if ((rc = upload(ctx, scene->emitters, &ds.emitters)) != RL_OK) break;
        if ((rc = upload(ctx, scene->emitters_cdf, &ds.emitters_cdf)) != RL_OK) break;
        ds.n_emitters = (uint32_t)scene->emitters.size();
        ds.env_emitter = -1;
        for (size_t e = 0; e < scene->emitters.size(); e++)
            if (scene->emitters[e].kind == EMITTER_ENV) {
                ds.env_emitter = (int32_t)e;
                for (int k = 0; k < 3; k++) ds.env_color[k] = scene->emitters[e].c[k];
                // EnvironmentLight::direct_pdf = SolidAngle(1 / (4 pi)) * EmitterSampler::pdf(env)
                ds.env_pdf = (1.0f / (3.14159265358979323846f * 4.0f)) * (scene->emitters_cdf[e + 1] - scene->emitters_cdf[e]);
                ds.env_sel_pdf = scene->emitters_cdf[e + 1] - scene->emitters_cdf[e];
            }
        ds.env_w = ds.env_h = 0;
        if (ds.env_emitter >= 0 && scene->env_map.w) {
            ds.env_w = scene->env_map.w; ds.env_h = scene->env_map.h;
            ds.env_marg_func_int = scene->env_marg_func_int;
            if ((rc = upload(ctx, scene->env_map.rgb, &ds.env_texels)) != RL_OK) break;
            if ((rc = upload(ctx, scene->env_cond_cdf, &ds.env_cond_cdf)) != RL_OK) break;
            if ((rc = upload(ctx, scene->env_cond_func, &ds.env_cond_func)) != RL_OK) break;
            if ((rc = upload(ctx, scene->env_marg_cdf, &ds.env_marg_cdf)) != RL_OK) break;
        }
        if (ds.env_emitter >= 0 && scene->medium.enabled) { rl_set_error("an environment emitter cannot be combined with a medium (paths/edge.rs:94)"); rc = RL_ERR_UNSUPPORTED; break; }
        if ((rc = upload(ctx, flat.mesh_cdf, &ds.mesh_cdf)) != RL_OK) break;
        ds.n_meshes = (uint32_t)flat.meshes.size();
        scene->sample_to_camera.to_cols(ds.camera.sample_to_camera);
        scene->to_world.to_cols(ds.camera.to_world);
        ds.camera.position[0] = scene->cam_pos.x; ds.camera.position[1] = scene->cam_pos.y; ds.camera.position[2] = scene->cam_pos.z;
        ds.camera.width = scene->width; ds.camera.height = scene->height;
        ds.medium = scene->medium;
        ctx->area_lights_only = scene->ats_root < 0 && !ctx->knobs.has(K_GENERIC_LIGHTS);
        for (const EmitterRecord& e : scene->emitters) if (e.kind != EMITTER_MESH) ctx->area_lights_only = false;
        for (const HostMesh& hm : scene->meshes) if (hm.is_light && hm.emission_type != RL_EMISSION_COLOR) ctx->area_lights_only = false;      // uv-dependent emission (`-x hvs-light | texture-light`): the generic instantiation
        ctx->single_bsdf = true;
        ctx->bsdf_type = flat.materials.empty() ? 0 : flat.materials[0].type;
        for (const Material& m : flat.materials) if (m.type != ctx->bsdf_type) ctx->single_bsdf = false;
        // stage the scene in LDS when nodes + triangles are small (<= 48 KiB leaves room for the stacks)
        ctx->scene_lds_bytes = (size_t)16 * lds_scene_float4s(ds.n_nodes, ds.n_prims);     // padded LDS layout (trace.hip.h)
        // LDS-staged scenes also keep their whole traversal stack in LDS (TravStackT<true>)
        ctx->lds_scene = ctx->scene_lds_bytes <= 48 * 1024 && ds.stack_depth <= (uint32_t)kLdsStackLevels && (!RL_LDS_TWO_LEVEL || (size_t)16 * lds_scene2_float4s(ds.n_nodes, ds.n_prims) <= 64 * 1024);
        {   // worst-case dynamic LDS of any kernel that stages the scene: [scene][compaction list | cold path state][12 stack levels];
            // it must fit what a workgroup may ask for on this device, else the scene streams from L2 / HBM instead
            int lds_limit = 64 * 1024;
            if (hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) { (void)hipGetLastError(); lds_limit = 64 * 1024; }
            ctx->lds_limit = (size_t)std::max(lds_limit, 16 * 1024);
            const size_t scene2 = RL_LDS_TWO_LEVEL ? (size_t)16 * lds_scene2_float4s(ds.n_nodes, ds.n_prims) : 0;      // k_path_fused's staging (two-level node records)
            const size_t worst = std::max(ctx->scene_lds_bytes, scene2) + std::max<size_t>(272 * sizeof(unsigned), kFusedColdBytes) + (size_t)2 * kLdsStackLevels * 256 * sizeof(int);
            if (worst > (size_t)lds_limit) ctx->lds_scene = false;
        }
        if (ctx->knobs.has(K_FORCE_STREAMING)) ctx->lds_scene = false;     // dev / test knob: small scenes through the kernels that stream the BVH (tests/parity_fuzz.py)
        if (!ctx->lds_scene) {
            // scenes that stream their BVH: the tolerance build (`numerics = fast`) traverses the same tree collapsed into quantised BVH4 nodes
            Bvh4Build b4;
            build_bvh4(bvh, &b4);
            if ((rc = upload(ctx, b4.nodes, &ds.nodes4)) != RL_OK) break;
            ds.root4 = b4.root; ds.stack_depth4 = b4.stack_depth;
            // ... and the draw-count pass of reference-order streams reads the exact BVH2 through a copy laid out in 16-node treelet blocks (traverse_treelet)
            std::vector<BvhNode> blocks;
            int32_t root_t = RL_CHILD_NONE;
            treelet_blocks(bvh, &blocks, &root_t);
            if ((rc = upload(ctx, blocks, &ds.nodes_t)) != RL_OK) break;
            ds.root_t = root_t;
        }
        if (hipMalloc((void**)&ctx->d_counters, sizeof(Counters)) != hipSuccess) { rl_set_error("hipMalloc counters"); rc = RL_ERR_HIP; break; }
        if (hipHostMalloc((void**)&ctx->h_counters, sizeof(Counters)) != hipSuccess) { rl_set_error("hipHostMalloc counters"); rc = RL_ERR_HIP; break; }
    } while (0);
    if (rc != RL_OK) { rl_context_destroy(ctx); return rc; }
    *out = ctx;
    return RL_OK;
}

extern "C" void rl_context_destroy(rl_context* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (void* p : ctx->allocs) hipFree(p);
    void* scratch[] = {ctx->pool.f, ctx->pool.u, ctx->pool.q, ctx->d_owned, ctx->d_item_base, ctx->d_block_seeds, ctx->d_item_seed,
                       ctx->d_item_pixel, ctx->d_queues, ctx->d_qcounts, ctx->d_out, ctx->d_counters, ctx->d_partials, ctx->d_overflow, ctx->d_sample_buf,
                       ctx->d_sample_states, ctx->d_chain_states, ctx->d_trk_off, ctx->d_trk_st, ctx->d_trivial, ctx->d_spec_stats, ctx->d_queue, ctx->d_overflow2};
    for (void* p : scratch) if (p) hipFree(p);
    if (ctx->h_counters) hipHostFree(ctx->h_counters);
    for (hipEvent_t ev : ctx->events) hipEventDestroy(ev);
    if (ctx->ev_chain_done) hipEventDestroy(ctx->ev_chain_done);
    if (ctx->h_flags) hipHostFree(ctx->h_flags);
    if (ctx->h_list) hipHostFree(ctx->h_list);
    for (int k = 1; k < rl_context::kEvalStreams; k++) if (ctx->eval_streams[k] && ctx->eval_streams[k] != ctx->stream2) hipStreamDestroy(ctx->eval_streams[k]);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

template <typename T>
static int ensure(T** p, size_t* cap, size_t n) {
    if (*cap >= n && *p) return RL_OK;
    if (*p) hipFree(*p);
    *p = nullptr;
    HIP_OK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
    *cap = n;
    return RL_OK;
}

// Scenes that stream their BVH keep only kLdsStackLevelsStreaming levels in LDS: their pools are sparse (most waves of a
// workgroup exit after the compaction), so what limits the live waves per CU is how many workgroups' stacks fit in LDS —
// 508 k-triangle scene, 32 spp: 12 / 8 / 6 / 4 / 2 / 0 levels in LDS = 258 / 233 / 228 / 229 / 240 / 251 ms.
#ifndef RL_LDS_LEVELS_STREAMING
#define RL_LDS_LEVELS_STREAMING 6
#endif
static constexpr int kLdsStackLevelsStreaming = RL_LDS_LEVELS_STREAMING;
static int lds_levels_of(const rl_context* ctx) { return std::min<int>((int)ctx->ds.stack_depth, ctx->lds_scene ? kLdsStackLevels : kLdsStackLevelsStreaming); }
static size_t traversal_lds_bytes(const rl_context* ctx, bool lds_scene, unsigned block, bool with_list, int lds_levels = -1) {
    size_t stack = (size_t)2 * (lds_levels >= 0 ? lds_levels : lds_levels_of(ctx)) * block * sizeof(int);
    return (lds_scene ? ctx->scene_lds_bytes : 0) + (with_list ? 272 * sizeof(unsigned) : 0) + stack;   // [scene][compaction list][stacks]
}
// overflow levels beyond the LDS part, [2 * levels][n_threads] ints
static int stack_conf(rl_context* ctx, size_t n_threads, StackConf* out, bool second = false, int lds_levels = -1) {
    int*& d_overflow = second ? ctx->d_overflow2 : ctx->d_overflow;
    size_t& overflow_capacity = second ? ctx->overflow2_capacity : ctx->overflow_capacity;
    out->lds_levels = lds_levels >= 0 ? std::min(lds_levels, lds_levels_of(ctx)) : lds_levels_of(ctx);
    out->pre_group = 0;          // only k_stream_chain uses it (its launch code sets it)
    out->overflow = nullptr;
    out->overflow_stride = n_threads;
    int extra = (int)std::max(ctx->ds.stack_depth, ctx->ds.stack_depth4) - out->lds_levels;     // (the tolerance build's BVH4 stacks are the deeper ones)
    if (extra > 0) {
        size_t need = (size_t)2 * extra * n_threads;
        if (overflow_capacity < need) {
            if (d_overflow) hipFree(d_overflow);
            d_overflow = nullptr;
            overflow_capacity = 0;
            HIP_OK(hipMalloc((void**)&d_overflow, need * sizeof(int)));
            overflow_capacity = need;
        }
        out->overflow = d_overflow;
    }
    return RL_OK;
}

// ---- k_stream_spec: pixels whose camera samples take exactly two draws (Path::from_sensor's jitter) whatever the stream holds.
// Without a medium a camera ray that misses the scene ends its path at once (path.rs:152-166), and every camera ray of a pixel misses when the
// scene's bounding box lies outside the pyramid spanned by the camera position and the pixel's footprint [ix, ix+1) x [iy, iy+1).  The rays
// through a rectangle of the image plane (Camera::generate is a projective map of the raster position, camera.rs:81-91) fill the convex cone of its four
// corner rays; a box that lies wholly beyond ONE side plane of that cone touches none of them.  Evaluated in f64 on a footprint grown by a quarter
// pixel and a box grown by 1e-3 of its size — orders of magnitude more than the f32 rounding of the device's ray generation and slab / triangle
// tests — so a pixel flagged here cannot produce a hit on the device.  Bit c of words [8 b .. 8 b + 7]: block cursor c of owned block b.
struct TrivialInput { uint32_t W, H; bool medium, empty; float root_min[3], root_max[3]; CameraRecord camera; };
static TrivialInput trivial_input(const rl_context* ctx) {
    TrivialInput in{};
    in.W = ctx->width; in.H = ctx->height; in.medium = ctx->ds.medium.enabled != 0; in.empty = ctx->ds.root == RL_CHILD_NONE || ctx->ds.n_prims == 0;
    for (int k = 0; k < 3; k++) { in.root_min[k] = ctx->ds.root_min[k]; in.root_max[k] = ctx->ds.root_max[k]; }
    in.camera = ctx->ds.camera;
    return in;
}
static void trivial_pixel_masks(const TrivialInput& ti, const rl_path_params* params, const std::vector<unsigned>& owned, size_t nby, bool no_shortcut, std::vector<unsigned>* out) {
    const uint32_t W = ti.W, H = ti.H;
    out->assign(owned.size() * 8, 0u);
    const bool expand = !params->has_max_depth || 1u < params->max_depth;
    auto all_of_block = [&](size_t j, unsigned npx) { for (unsigned c = 0; c < npx; c++) (*out)[j * 8 + (c >> 5)] |= 1u << (c & 31u); };
    if (!expand) {      // the sensor vertex is never expanded: two draws per sample everywhere
        for (size_t j = 0; j < owned.size(); j++) {
            const unsigned bx = (unsigned)(owned[j] / nby) * 16u, by = (unsigned)(owned[j] % nby) * 16u;
            all_of_block(j, std::min(16u, W - bx) * std::min(16u, H - by));
        }
        return;
    }
    if (ti.medium || no_shortcut) return;      // Edge::from_ray samples the medium on a miss too: no shortcut
    const bool empty = ti.empty;
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = ti.root_min[k]; hi[k] = ti.root_max[k];
        if (!empty && !(std::isfinite(lo[k]) && std::isfinite(hi[k]) && lo[k] <= hi[k])) return;      // hostile geometry: no shortcut
        const double grow = 1e-3 * (hi[k] - lo[k]) + 1e-4 * std::max(1.0, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
        lo[k] -= grow; hi[k] += grow;
    }
    const float* m = ti.camera.sample_to_camera; const float* tw = ti.camera.to_world;
    const double cam[3] = {ti.camera.position[0], ti.camera.position[1], ti.camera.position[2]};
    // direction of the ray through raster position (u, v), not normalised; false when the projective map degenerates there
    auto ray_dir = [&](double u, double v, double* d) -> bool {
        const double sx = u / (double)W, sy = v / (double)H;
        const double hx = m[0] * sx + m[4] * sy + m[12], hy = m[1] * sx + m[5] * sy + m[13], hz = m[2] * sx + m[6] * sy + m[14], hw = m[3] * sx + m[7] * sy + m[15];
        if (!(std::fabs(hw) > 1e-12) || !std::isfinite(hx + hy + hz + hw)) return false;
        const double nx = hx / hw, ny = hy / hw, nz = hz / hw;
        d[0] = tw[0] * nx + tw[4] * ny + tw[8] * nz; d[1] = tw[1] * nx + tw[5] * ny + tw[9] * nz; d[2] = tw[2] * nx + tw[6] * ny + tw[10] * nz;
        d[3] = hw;
        const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (!(len > 1e-30) || !std::isfinite(len)) return false;
        d[0] /= len; d[1] /= len; d[2] /= len;
        return true;
    };
    // true when no ray through the raster rectangle [u0, u1] x [v0, v1] can meet the grown box
    auto misses = [&](double u0, double v0, double u1, double v1) -> bool {
        if (empty) return true;
        double d[4][4];
        const double us[4] = {u0, u1, u1, u0}, vs[4] = {v0, v0, v1, v1};
        for (int k = 0; k < 4; k++) if (!ray_dir(us[k], vs[k], d[k])) return false;
        for (int k = 1; k < 4; k++) if ((d[k][3] > 0) != (d[0][3] > 0)) return false;       // the footprint crosses the map's pole
        for (int k = 0; k < 4; k++) {
            const double* a = d[k]; const double* b = d[(k + 1) & 3]; const double* o = d[(k + 2) & 3];
            double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const double nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(nl > 1e-14)) return false;
            double inside = (n[0] * o[0] + n[1] * o[1] + n[2] * o[2]) / nl;          // the cone lies on the side of the opposite corner
            if (!(std::fabs(inside) > 1e-9)) return false;
            const double sgn = inside > 0 ? 1.0 : -1.0;
            bool all_out = true;
            for (int cbit = 0; cbit < 8 && all_out; cbit++) {
                const double p[3] = {((cbit & 1) ? hi[0] : lo[0]) - cam[0], ((cbit & 2) ? hi[1] : lo[1]) - cam[1], ((cbit & 4) ? hi[2] : lo[2]) - cam[2]};
                const double pl = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                const double s = sgn * (n[0] * p[0] + n[1] * p[1] + n[2] * p[2]) / nl;
                if (!(s < -1e-7 * pl - 1e-12)) all_out = false;
            }
            if (all_out) return true;
        }
        return false;
    };
    for (size_t j = 0; j < owned.size(); j++) {
        const unsigned bx = (unsigned)(owned[j] / nby) * 16u, by = (unsigned)(owned[j] % nby) * 16u;
        const unsigned bw = std::min(16u, W - bx), bh = std::min(16u, H - by);
        if (misses(bx - 0.25, by - 0.25, bx + bw + 0.25, by + bh + 0.25)) { all_of_block(j, bw * bh); continue; }
        for (unsigned c = 0; c < bw * bh; c++) {
            const double x = bx + c % bw, y = by + c / bw;
            if (misses(x - 0.25, y - 0.25, x + 1.25, y + 1.25)) (*out)[j * 8 + (c >> 5)] |= 1u << (c & 31u);
        }
    }
}

// test hook, host only (no GPU): the pixels k_stream_spec would treat as taking two draws per sample — out[y * W + x] = 1 — for the scene's camera and the bounding box of
// its BVH (built here as rl_context_create builds it).  tests/test_abi.py checks every flagged pixel against the oracle's camera rays and traversal.
extern "C" int rl_debug_trivial_pixels(const rl_scene* scene, int has_max_depth, uint32_t max_depth, uint8_t* out) {
    if (!scene || !out || !scene->has_camera) return RL_ERR_INVALID_ARGUMENT;
    BvhBuild bvh;
    build_bvh(*scene, &bvh);
    TrivialInput ti{};
    ti.W = scene->width; ti.H = scene->height; ti.medium = scene->medium.enabled != 0; ti.empty = bvh.root == RL_CHILD_NONE || bvh.tris.empty();
    for (int k = 0; k < 3; k++) { ti.root_min[k] = bvh.root_min[k]; ti.root_max[k] = bvh.root_max[k]; }
    scene->sample_to_camera.to_cols(ti.camera.sample_to_camera);
    scene->to_world.to_cols(ti.camera.to_world);
    ti.camera.position[0] = scene->cam_pos.x; ti.camera.position[1] = scene->cam_pos.y; ti.camera.position[2] = scene->cam_pos.z;
    ti.camera.width = scene->width; ti.camera.height = scene->height;
    rl_path_params pp{};
    pp.has_max_depth = has_max_depth; pp.max_depth = max_depth;
    const size_t nbx = (ti.W + 15) / 16, nby = (ti.H + 15) / 16;
    std::vector<unsigned> owned(nbx * nby), masks;
    for (size_t b = 0; b < owned.size(); b++) owned[b] = (unsigned)b;
    trivial_pixel_masks(ti, &pp, owned, nby, false, &masks);
    std::memset(out, 0, (size_t)ti.W * ti.H);
    for (size_t b = 0; b < owned.size(); b++) {
        const unsigned bx = (unsigned)(b / nby) * 16u, by = (unsigned)(b % nby) * 16u, bw = std::min(16u, ti.W - bx), bh = std::min(16u, ti.H - by);
        for (unsigned c = 0; c < bw * bh; c++) if ((masks[b * 8 + (c >> 5)] >> (c & 31u)) & 1u) out[(size_t)(by + c / bw) * ti.W + bx + c % bw] = 1;
    }
    return RL_OK;
}

#include "path_render.hip.h"

extern "C" int rl_render_path(rl_context* ctx, const rl_path_params* params, const uint64_t* block_seeds, size_t n_blocks, float* out_rgb,
                              int out_is_device, void* stream_arg, rl_render_stats* stats) {
    if (!ctx || !params || !block_seeds || !out_rgb) return RL_ERR_INVALID_ARGUMENT;
    const uint32_t W = ctx->width, H = ctx->height;
    const size_t nbx = (W + 15) / 16, nby = (H + 15) / 16;
    if (n_blocks != nbx * nby) { rl_set_error("n_blocks does not match the image size"); return RL_ERR_INVALID_ARGUMENT; }
    if (params->spp == 0) { rl_set_error("spp must be > 0 (assert_ne!(scene.nb_samples, 0), mod.rs:410)"); return RL_ERR_INVALID_ARGUMENT; }
    if (params->strategy < 0 || params->strategy > 2) return RL_ERR_INVALID_ARGUMENT;
    if (params->stream_mode != RL_STREAM_REFERENCE_ORDER && params->stream_mode != RL_STREAM_PER_SAMPLE) return RL_ERR_INVALID_ARGUMENT;
    if (params->numerics > RL_NUMERICS_FAST) { rl_set_error("numerics must be 0 (exact) or 1 (fast)"); return RL_ERR_INVALID_ARGUMENT; }
    const uint32_t shard_count = params->shard_count ? params->shard_count : 1;
    if (params->shard_index >= shard_count) return RL_ERR_INVALID_ARGUMENT;
    if (params->strategy != RL_STRATEGY_BSDF && ctx->ds.n_emitters == 0) { rl_set_error("light sampling requested but the scene has no emitter"); return RL_ERR_NO_EMITTER; }
    HIP_OK(hipSetDevice(ctx->device));
    PathRender job(ctx, params, block_seeds, n_blocks, out_rgb, out_is_device, stream_arg ? (hipStream_t)stream_arg : ctx->stream, stats);
    return job.run();
}

// Execution options of a context (kernels/knobs.h): `name` as listed there ("spec_force", "no_overlap", "state_budget_mb", ...), `value` its value as text, NULL = back to
// the default.  rl_context_create fills the table from the environment once; nothing below rl_render_path / rl_render_ao / rl_render_direct reads the environment.
// Not to be called while a render runs on the same context (a render works on the copy it took when it started).  The two options that shape the context itself
// (force_streaming, generic_lights) are read when it is created and refused here.
extern "C" int rl_context_set_option(rl_context* ctx, const char* name, const char* value) {
    if (!ctx || !name) return RL_ERR_INVALID_ARGUMENT;
    const int k = Knobs::find(name);
    if (k < 0) { rl_set_error(std::string("unknown option: ") + name); return RL_ERR_INVALID_ARGUMENT; }
    if (k == K_FORCE_STREAMING || k == K_GENERIC_LIGHTS) { rl_set_error(std::string(name) + " is read when the context is created (environment RL_FORCE_STREAMING / RL_GENERIC_LIGHTS)"); return RL_ERR_UNSUPPORTED; }
    ctx->knobs.e[k].set = value != nullptr;
    ctx->knobs.e[k].value = value ? value : "";
    return RL_OK;
}
// the value an option holds (NULL: not set, the default applies); the pointer stays valid until the option is set again
extern "C" const char* rl_context_get_option(const rl_context* ctx, const char* name) {
    if (!ctx) return nullptr;
    const int k = Knobs::find(name);
    return k < 0 ? nullptr : ctx->knobs.str(k);
}

// test hook: rng_advance (rngjump.h) on the device — states_out[i] = the sampler states_in[i] after counts[i] more draws
namespace rl {
__global__ void k_debug_rng_advance(unsigned n, const unsigned long long* in, const unsigned* counts, unsigned long long* out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    Rng r; r.s0 = r.s1 = r.s2 = r.s3 = 0ull;
    if (i < n) { r.s0 = in[4 * i]; r.s1 = in[4 * i + 1]; r.s2 = in[4 * i + 2]; r.s3 = in[4 * i + 3]; }
    rng_advance(r, i < n ? counts[i] : 0u);
    if (i < n) { out[4 * i] = r.s0; out[4 * i + 1] = r.s1; out[4 * i + 2] = r.s2; out[4 * i + 3] = r.s3; }
}
}  // namespace rl
extern "C" int rl_debug_rng_advance(int device, size_t n, const uint64_t* states_in, const uint32_t* counts, uint64_t* states_out) {
    if (!states_in || !counts || !states_out || n == 0 || n > (1u << 24)) return RL_ERR_INVALID_ARGUMENT;
    HIP_OK(hipSetDevice(device));
    unsigned long long *d_in = nullptr, *d_out = nullptr; unsigned* d_c = nullptr;
    HIP_OK(hipMalloc((void**)&d_in, n * 32)); HIP_OK(hipMalloc((void**)&d_out, n * 32)); HIP_OK(hipMalloc((void**)&d_c, n * 4));
    HIP_OK(hipMemcpy(d_in, states_in, n * 32, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_c, counts, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rl::k_debug_rng_advance, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, (unsigned)n, d_in, d_c, d_out);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(states_out, d_out, n * 32, hipMemcpyDeviceToHost));
    hipFree(d_in); hipFree(d_out); hipFree(d_c);
    return RL_OK;
}

// ---- ao / direct: Integrator::compute through the same tiling driver (one launch)
static int render_mc(rl_context* ctx, int kind, const rl_mc_params* params, const uint64_t* block_seeds, size_t n_blocks, float* out_rgb,
                     int out_is_device, void* stream_arg, rl_render_stats* stats) {
    if (!ctx || !params || !block_seeds || !out_rgb) return RL_ERR_INVALID_ARGUMENT;
    const uint32_t W = ctx->width, H = ctx->height;
    const size_t nbx = (W + 15) / 16, nby = (H + 15) / 16;
    if (n_blocks != nbx * nby || params->spp == 0) return RL_ERR_INVALID_ARGUMENT;
    if (params->stream_mode != RL_STREAM_REFERENCE_ORDER && params->stream_mode != RL_STREAM_PER_SAMPLE) return RL_ERR_INVALID_ARGUMENT;
    const uint32_t shard_count = params->shard_count ? params->shard_count : 1;
    if (params->shard_index >= shard_count) return RL_ERR_INVALID_ARGUMENT;
    if (kind == 1 && params->nb_light_samples > 0 && ctx->ds.n_emitters == 0) { rl_set_error("light sampling requested but the scene has no emitter"); return RL_ERR_NO_EMITTER; }
    HIP_OK(hipSetDevice(ctx->device));
    hipStream_t st = stream_arg ? (hipStream_t)stream_arg : ctx->stream;
    auto t_start = std::chrono::steady_clock::now();
    std::vector<unsigned> owned, item_base;
    unsigned n_pixels = 0;
    for (size_t b = 0; b < n_blocks; b++) {
        if (b % shard_count != params->shard_index) continue;
        unsigned bx = (unsigned)(b / nby) * 16u, by = (unsigned)(b % nby) * 16u;
        owned.push_back((unsigned)b);
        item_base.push_back(n_pixels);
        n_pixels += std::min(16u, W - bx) * std::min(16u, H - by);
    }
    const bool per_sample = params->stream_mode == RL_STREAM_PER_SAMPLE;
    // reference-order streams in two passes, as for `path` (chain.hip.h): k_mc_chain records where every camera sample starts in its block's stream (a sample's
    // draw count follows from its camera ray alone), then the per-pixel form evaluates all samples from those states.  One chunk: when the states do not fit
    // their budget the single-pass walk (one lane per block) runs instead, as it does under RL_REF_SINGLE_PASS.
    const Knobs knobs = ctx->knobs;       // (the context's options as they stand now: knobs.h)
    size_t state_budget = (size_t)24 << 30;
    if (knobs.has(K_STATE_BUDGET_MB)) state_budget = std::max<size_t>(1, (size_t)knobs.i(K_STATE_BUDGET_MB, 0)) << 20;
    const bool two_pass = !per_sample && !owned.empty() && !knobs.has(K_REF_SINGLE_PASS) && (size_t)n_pixels * params->spp * 32 <= state_budget;
    const bool per_pixel = per_sample || two_pass;
    const unsigned n_items = per_pixel ? n_pixels : (unsigned)owned.size();
    unsigned chain_shift = 0;
    if (two_pass) {
        int cus = 256;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        while (chain_shift < 6u && (owned.size() << (chain_shift + 1u)) <= (size_t)cus * 4u * 256u) chain_shift++;
        if (knobs.has(K_ITEM_SHIFT)) chain_shift = std::min(6u, (unsigned)std::max<long long>(0, knobs.i(K_ITEM_SHIFT, 0)));
    }
    const unsigned chain_threads = two_pass ? std::max(256u, (unsigned)((((size_t)owned.size() << chain_shift) + 255u) / 256u * 256u)) : 0u;
    const unsigned n_threads = std::max(chain_threads, std::max(256u, (n_items + 255u) / 256u * 256u));
    int rcode;
    if ((rcode = ensure(&ctx->d_owned, &ctx->owned_capacity, owned.size())) != RL_OK) return rcode;
    if ((rcode = ensure(&ctx->d_item_base, &ctx->item_base_capacity, owned.size())) != RL_OK) return rcode;
    if ((rcode = ensure(&ctx->d_block_seeds, &ctx->seeds_capacity, n_blocks)) != RL_OK) return rcode;
    if (per_pixel) {
        if ((rcode = ensure(&ctx->d_item_seed, &ctx->item_capacity, n_pixels)) != RL_OK) return rcode;
        if ((rcode = ensure(&ctx->d_item_pixel, &ctx->item_pixel_capacity, n_pixels)) != RL_OK) return rcode;
    }
    if (two_pass && (rcode = ensure(&ctx->d_sample_states, &ctx->sample_states_capacity, (size_t)n_pixels * params->spp * 4)) != RL_OK) return rcode;
    float* d_out = out_rgb;
    if (!out_is_device) {
        if ((rcode = ensure(&ctx->d_out, &ctx->out_capacity, (size_t)3 * W * H)) != RL_OK) return rcode;
        d_out = ctx->d_out;
    }
    const size_t n_rows = n_threads / 256;
    if ((rcode = ensure(&ctx->d_partials, &ctx->partials_capacity, n_rows * STAT_COUNT)) != RL_OK) return rcode;
    HIP_OK(hipMemcpyAsync(ctx->d_owned, owned.data(), owned.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(ctx->d_item_base, item_base.data(), item_base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(ctx->d_block_seeds, block_seeds, n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(d_out, 0, (size_t)3 * W * H * sizeof(float), st));
    HIP_OK(hipMemsetAsync(ctx->d_partials, 0, n_rows * STAT_COUNT * sizeof(unsigned long long), st));
    RenderConst rc{};
    rc.spp = params->spp;
    rc.stream_mode = params->stream_mode; rc.seed_variant = params->seed_variant;
    rc.inv_spp = 1.0f / (float)params->spp;
    rc.W = W; rc.H = H; rc.nby = (unsigned)nby;
    rc.n_items = n_items;
    rc.split = 1; rc.sample_buf = nullptr;
    rc.owned_blocks = ctx->d_owned; rc.block_item_base = ctx->d_item_base; rc.n_owned = (unsigned)owned.size();
    rc.block_seeds = ctx->d_block_seeds;
    rc.item_seed = ctx->d_item_seed; rc.item_pixel = ctx->d_item_pixel;
    rc.out = d_out;
    rc.counters = ctx->d_counters;
    rc.partials = ctx->d_partials;
    rc.sample_states = ctx->d_sample_states; rc.n_state_pixels = n_pixels;
    McConst mp{params->has_max_distance, params->max_distance, params->normal_correction, params->nb_bsdf_samples, params->nb_light_samples};
    StackConf stc;
    if ((rcode = stack_conf(ctx, n_threads, &stc)) != RL_OK) return rcode;
    const size_t lds = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false);
    const dim3 grid((std::max(256u, (n_items + 255u) / 256u * 256u)) / 256), block(256);
    const bool timing = stats != nullptr && !knobs.has(K_NO_EVENTS);
    while (timing && ctx->events.size() < 4) { hipEvent_t ev; HIP_OK(hipEventCreate(&ev)); ctx->events.push_back(ev); }
    if (per_pixel && !owned.empty()) hipLaunchKernelGGL(k_seed_pixels, dim3(((unsigned)owned.size() + 63) / 64), dim3(64), 0, st, rc);     // (two-pass: for item_pixel)
    if (two_pass) {
        RenderConst ra = rc;
        ra.n_items = (unsigned)owned.size(); ra.item_shift = chain_shift;
        if (timing) hipEventRecord(ctx->events[0], st);
        launch_mc_chain(kind, ctx->lds_scene, dim3(chain_threads / 256), block, lds, st, ra, ctx->ds, stc, mp);
        if (timing) hipEventRecord(ctx->events[1], st);
        rc.stream_mode = kStreamGivenStates;
    }
    if (timing) hipEventRecord(ctx->events[2], st);
    launch_pixel_mc(kind, ctx->lds_scene, grid, block, lds, st, rc, ctx->ds, stc, mp);
    if (timing) hipEventRecord(ctx->events[3], st);
    if (!out_is_device) HIP_OK(hipMemcpyAsync(out_rgb, d_out, (size_t)3 * W * H * sizeof(float), hipMemcpyDeviceToHost, st));
    std::vector<unsigned long long> partials(n_rows * STAT_COUNT);
    HIP_OK(hipMemcpyAsync(partials.data(), ctx->d_partials, partials.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        unsigned long long totals[STAT_COUNT] = {0};
        for (size_t r = 0; r < n_rows; r++) for (int k = 0; k < STAT_COUNT; k++) totals[k] += partials[r * STAT_COUNT + k];
        stats->camera_samples = totals[STAT_SAMPLES]; stats->vertices = totals[STAT_VERTICES]; stats->extension_rays = totals[STAT_EXT_RAYS];
        stats->shadow_rays = totals[STAT_SHADOW_RAYS]; stats->rng_draws = totals[STAT_DRAWS];
        stats->iterations = 1; stats->kernel_launches = per_sample ? 2 : (two_pass ? 3 : 1);
        if (timing) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ctx->events[2], ctx->events[3]) == hipSuccess) stats->ms_other = t;
            if (two_pass && hipEventElapsedTime(&t, ctx->events[0], ctx->events[1]) == hipSuccess) stats->ms_prepass = t;
            (void)hipGetLastError();
        }
        stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
    return RL_OK;
}
extern "C" int rl_render_ao(rl_context* ctx, const rl_mc_params* params, const uint64_t* block_seeds, size_t n_blocks, float* out_rgb, int out_is_device,
                            void* stream, rl_render_stats* stats) { return render_mc(ctx, 0, params, block_seeds, n_blocks, out_rgb, out_is_device, stream, stats); }
extern "C" int rl_render_direct(rl_context* ctx, const rl_mc_params* params, const uint64_t* block_seeds, size_t n_blocks, float* out_rgb, int out_is_device,
                                void* stream, rl_render_stats* stats) { return render_mc(ctx, 1, params, block_seeds, n_blocks, out_rgb, out_is_device, stream, stats); }

// ---- batched Acceleration::{trace, visible}
namespace {
struct DevBuf {     // frees on every exit path of the batch entry points
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};
constexpr size_t kMaxBatch = 0x7fffff00u;   // grid and kernel argument are 32-bit
}  // namespace
extern "C" int rl_trace_batch(rl_context* ctx, size_t n, const float* origins, const float* directions, float* t_out, float* u_out,
                              float* v_out, int32_t* mesh_out, int32_t* tri_out) {
    if (!ctx || (n && (!origins || !directions || !t_out || !u_out || !v_out || !mesh_out || !tri_out))) return RL_ERR_INVALID_ARGUMENT;
    if (n == 0) return RL_OK;
    if (n > kMaxBatch) { rl_set_error("batch too large"); return RL_ERR_INVALID_ARGUMENT; }
    HIP_OK(hipSetDevice(ctx->device));
    DevBuf b_o, b_d, b_t, b_u, b_v, b_m, b_tr;
    HIP_OK(b_o.alloc(3 * n * 4)); HIP_OK(b_d.alloc(3 * n * 4));
    HIP_OK(b_t.alloc(n * 4)); HIP_OK(b_u.alloc(n * 4)); HIP_OK(b_v.alloc(n * 4));
    HIP_OK(b_m.alloc(n * 4)); HIP_OK(b_tr.alloc(n * 4));
    float *d_o = b_o.as<float>(), *d_d = b_d.as<float>(), *d_t = b_t.as<float>(), *d_u = b_u.as<float>(), *d_v = b_v.as<float>();
    int *d_m = b_m.as<int>(), *d_tr = b_tr.as<int>();
    HIP_OK(hipMemcpy(d_o, origins, 3 * n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_d, directions, 3 * n * 4, hipMemcpyHostToDevice));
    size_t lds = traversal_lds_bytes(ctx, false, 256, false);
    StackConf stc;
    { int r = stack_conf(ctx, (n + 255) / 256 * 256, &stc); if (r != RL_OK) return r; }
    hipLaunchKernelGGL(k_trace_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, ctx->stream, ctx->ds, stc, (unsigned)n, d_o, d_d, d_t, d_u, d_v, d_m, d_tr);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(ctx->stream));
    HIP_OK(hipMemcpy(t_out, d_t, n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(u_out, d_u, n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(v_out, d_v, n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(mesh_out, d_m, n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(tri_out, d_tr, n * 4, hipMemcpyDeviceToHost));
    return RL_OK;
}

extern "C" int rl_visible_batch(rl_context* ctx, size_t n, const float* p0, const float* p1, uint8_t* visible_out) {
    if (!ctx || (n && (!p0 || !p1 || !visible_out))) return RL_ERR_INVALID_ARGUMENT;
    if (n == 0) return RL_OK;
    if (n > kMaxBatch) { rl_set_error("batch too large"); return RL_ERR_INVALID_ARGUMENT; }
    HIP_OK(hipSetDevice(ctx->device));
    DevBuf b_a, b_b, b_o;
    HIP_OK(b_a.alloc(3 * n * 4)); HIP_OK(b_b.alloc(3 * n * 4)); HIP_OK(b_o.alloc(n));
    float *d_a = b_a.as<float>(), *d_b = b_b.as<float>(); unsigned char* d_o = b_o.as<unsigned char>();
    HIP_OK(hipMemcpy(d_a, p0, 3 * n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_b, p1, 3 * n * 4, hipMemcpyHostToDevice));
    size_t lds = traversal_lds_bytes(ctx, false, 256, false);
    StackConf stc;
    { int r = stack_conf(ctx, (n + 255) / 256 * 256, &stc); if (r != RL_OK) return r; }
    hipLaunchKernelGGL(k_visible_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, ctx->stream, ctx->ds, stc, (unsigned)n, d_a, d_b, d_o);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(ctx->stream));
    HIP_OK(hipMemcpy(visible_out, d_o, n, hipMemcpyDeviceToHost));
    return RL_OK;
}

// ---- debug / test hooks (declared in wavefront.h, exported for the test-suite)
extern "C" int rl_debug_numerics(int device, size_t n, const float* a, const float* b, float* out8) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return RL_ERR_NO_DEVICE;
    if (n > kMaxBatch / 16) return RL_ERR_INVALID_ARGUMENT;
    HIP_OK(hipSetDevice(device));
    DevBuf b_a, b_b, b_out;
    HIP_OK(b_a.alloc(n * 4)); HIP_OK(b_b.alloc(n * 4)); HIP_OK(b_out.alloc(10 * n * 4));
    float *d_a = b_a.as<float>(), *d_b = b_b.as<float>(), *d_out = b_out.as<float>();
    HIP_OK(hipMemcpy(d_a, a, n * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_b, b, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_numerics_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned)n, d_a, d_b, d_out, d_out + n, d_out + 2 * n,
                       d_out + 3 * n, d_out + 4 * n, d_out + 5 * n, d_out + 6 * n, d_out + 7 * n, d_out + 8 * n, d_out + 9 * n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out8, d_out, 10 * n * 4, hipMemcpyDeviceToHost));
    return RL_OK;
}

extern "C" int rl_debug_trace_batch_fast(rl_context* ctx, size_t n, const float* origins, const float* directions, float* t_out, int32_t* mesh_out, int32_t* tri_out, int32_t* steps_out) {
    if (!ctx || !n || !origins || !directions || !t_out || !mesh_out || !tri_out || !steps_out) return RL_ERR_INVALID_ARGUMENT;
    if (!ctx->ds.nodes4) { rl_set_error("the scene is staged in LDS: no BVH4"); return RL_ERR_UNSUPPORTED; }
    if (n > kMaxBatch) { rl_set_error("batch too large"); return RL_ERR_INVALID_ARGUMENT; }
    HIP_OK(hipSetDevice(ctx->device));
    DevBuf b_o, b_d, b_t, b_m, b_tr, b_s;
    HIP_OK(b_o.alloc(3 * n * 4)); HIP_OK(b_d.alloc(3 * n * 4)); HIP_OK(b_t.alloc(n * 4)); HIP_OK(b_m.alloc(n * 4)); HIP_OK(b_tr.alloc(n * 4)); HIP_OK(b_s.alloc(n * 4));
    HIP_OK(hipMemcpy(b_o.as<float>(), origins, 3 * n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(b_d.as<float>(), directions, 3 * n * 4, hipMemcpyHostToDevice));
    StackConf stc;
    { int r = stack_conf(ctx, (n + 255) / 256 * 256, &stc); if (r != RL_OK) return r; }
    launch_trace_batch_fast(dim3((unsigned)((n + 255) / 256)), dim3(256), traversal_lds_bytes(ctx, false, 256, false), ctx->stream, ctx->ds, stc, (unsigned)n, b_o.as<float>(), b_d.as<float>(),
                            b_t.as<float>(), b_m.as<int>(), b_tr.as<int>(), b_s.as<int>());
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(ctx->stream));
    HIP_OK(hipMemcpy(t_out, b_t.as<float>(), n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(mesh_out, b_m.as<int>(), n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(tri_out, b_tr.as<int>(), n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(steps_out, b_s.as<int>(), n * 4, hipMemcpyDeviceToHost));
    return RL_OK;
}

// test hook: rl_trace_batch through the two-level records (traverse2) + node trips per ray; any_hit != 0: t_inout holds the segment lengths on entry and 1 / 0
// (a triangle was found / not) on return, as Acceleration::visible's inner `intersect` would (u, v, mesh, tri untouched)
extern "C" int rl_debug_trace_batch_two_level(rl_context* ctx, size_t n, const float* origins, const float* directions, float* t_inout, float* u_out, float* v_out,
                                              int32_t* mesh_out, int32_t* tri_out, int32_t* steps_out, int any_hit) {
    if (!ctx || !n || !origins || !directions || !t_inout || !u_out || !v_out || !mesh_out || !tri_out || !steps_out) return RL_ERR_INVALID_ARGUMENT;
    if (n > kMaxBatch) { rl_set_error("batch too large"); return RL_ERR_INVALID_ARGUMENT; }
    HIP_OK(hipSetDevice(ctx->device));
    { const int r2 = ensure_two_level(ctx); if (r2 != RL_OK) return r2; }
    DevBuf b_o, b_d, b_t, b_u, b_v, b_m, b_tr, b_s;
    HIP_OK(b_o.alloc(3 * n * 4)); HIP_OK(b_d.alloc(3 * n * 4)); HIP_OK(b_t.alloc(n * 4)); HIP_OK(b_u.alloc(n * 4)); HIP_OK(b_v.alloc(n * 4));
    HIP_OK(b_m.alloc(n * 4)); HIP_OK(b_tr.alloc(n * 4)); HIP_OK(b_s.alloc(n * 4));
    HIP_OK(hipMemcpy(b_o.as<float>(), origins, 3 * n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(b_d.as<float>(), directions, 3 * n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(b_t.as<float>(), t_inout, n * 4, hipMemcpyHostToDevice));
    StackConf stc;
    { int r = stack_conf(ctx, (n + 255) / 256 * 256, &stc); if (r != RL_OK) return r; }
    hipLaunchKernelGGL(k_trace_batch_two_level, dim3((unsigned)((n + 255) / 256)), dim3(256), traversal_lds_bytes(ctx, false, 256, false), ctx->stream, ctx->ds, stc, (unsigned)n,
                       b_o.as<float>(), b_d.as<float>(), b_t.as<float>(), b_u.as<float>(), b_v.as<float>(), b_m.as<int>(), b_tr.as<int>(), b_s.as<int>(), any_hit);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(ctx->stream));
    HIP_OK(hipMemcpy(t_inout, b_t.as<float>(), n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(steps_out, b_s.as<int>(), n * 4, hipMemcpyDeviceToHost));
    if (!any_hit) {
        HIP_OK(hipMemcpy(u_out, b_u.as<float>(), n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(v_out, b_v.as<float>(), n * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(mesh_out, b_m.as<int>(), n * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(tri_out, b_tr.as<int>(), n * 4, hipMemcpyDeviceToHost));
    }
    return RL_OK;
}

extern "C" int rl_debug_bvh_sizes(const rl_context* ctx, uint64_t* n_ref_nodes, uint64_t* n_prims, uint32_t* stack_depth, int* lds_scene) {
    if (!ctx) return RL_ERR_INVALID_ARGUMENT;
    *n_ref_nodes = ctx->bvh_dump.ref_info.size(); *n_prims = ctx->bvh_dump.ref_prim_mesh.size();
    *stack_depth = ctx->bvh_dump.stack_depth; *lds_scene = ctx->lds_scene ? 1 : 0;
    return RL_OK;
}

extern "C" const char* rl_build_info(void) { return "rustlight_amd wavefront path tracer; kernels: gfx950 (hipcc, -ffp-contract=off)"; }
