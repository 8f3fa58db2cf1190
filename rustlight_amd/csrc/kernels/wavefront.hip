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
    double draws_per_sample = 0.0;         // measured by the last path render of this context (0: none yet): k_stream_spec only pays when a pixel's samples are short next to spp
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
    BvhBuild bvh_dump;                // kept for rl_debug_bvh
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
        {   // the exact build's two-level records (traverse2): what every kernel that streams the BVH — and rl_trace_batch / rl_visible_batch on any scene — reads
            std::vector<BvhNode2> n2;
            two_level_nodes(bvh, &n2);
            if ((rc = upload(ctx, n2, &ds.nodes2)) != RL_OK) break;
        }
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
        ctx->area_lights_only = scene->ats_root < 0 && !getenv("RL_GENERIC_LIGHTS");
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
        if (getenv("RL_FORCE_STREAMING")) ctx->lds_scene = false;     // dev / test knob: small scenes through the kernels that stream the BVH (tests/parity_fuzz.py)
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
static size_t traversal_lds_bytes(const rl_context* ctx, bool lds_scene, unsigned block, bool with_list) {
    size_t stack = (size_t)2 * lds_levels_of(ctx) * block * sizeof(int);
    return (lds_scene ? ctx->scene_lds_bytes : 0) + (with_list ? 272 * sizeof(unsigned) : 0) + stack;   // [scene][compaction list][stacks]
}
// overflow levels beyond the LDS part, [2 * levels][n_threads] ints
static int stack_conf(rl_context* ctx, size_t n_threads, StackConf* out, bool second = false) {
    int*& d_overflow = second ? ctx->d_overflow2 : ctx->d_overflow;
    size_t& overflow_capacity = second ? ctx->overflow2_capacity : ctx->overflow_capacity;
    out->lds_levels = lds_levels_of(ctx);
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
static void trivial_pixel_masks(const TrivialInput& ti, const rl_path_params* params, const std::vector<unsigned>& owned, size_t nby, std::vector<unsigned>* out) {
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
    if (ti.medium || getenv("RL_SPEC_NO_TRIVIAL")) return;      // Edge::from_ray samples the medium on a miss too: no shortcut
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
    trivial_pixel_masks(ti, &pp, owned, nby, &masks);
    std::memset(out, 0, (size_t)ti.W * ti.H);
    for (size_t b = 0; b < owned.size(); b++) {
        const unsigned bx = (unsigned)(b / nby) * 16u, by = (unsigned)(b % nby) * 16u, bw = std::min(16u, ti.W - bx), bh = std::min(16u, ti.H - by);
        for (unsigned c = 0; c < bw * bh; c++) if ((masks[b * 8 + (c >> 5)] >> (c & 31u)) & 1u) out[(size_t)(by + c / bw) * ti.W + bx + c % bw] = 1;
    }
    return RL_OK;
}

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
    hipStream_t st = stream_arg ? (hipStream_t)stream_arg : ctx->stream;
    auto t_start = std::chrono::steady_clock::now();

    // ---- work decomposition: this shard's blocks, in creation order
    std::vector<unsigned> owned, item_base;
    unsigned n_pixels = 0;
    for (size_t b = 0; b < n_blocks; b++) {
        if (b % shard_count != params->shard_index) continue;
        unsigned bx = (unsigned)(b / nby) * 16u, by = (unsigned)(b % nby) * 16u;
        unsigned bw = std::min(16u, W - bx), bh = std::min(16u, H - by);
        owned.push_back((unsigned)b);
        item_base.push_back(n_pixels);
        n_pixels += bw * bh;
    }
    const bool per_sample = params->stream_mode == RL_STREAM_PER_SAMPLE;
    // pipeline: 1 = wavefront stage kernels, 2 = persistent fused kernel, 0 = auto = fused unless a pool size is forced (reference-order
    // streams at 1080p x 128 spp: wavefront 8.7 s, fused 2.9 s, fused with the items spread over the waves 1.7 s, fused in two passes: see
    // DESIGN.md; per-sample at 1080p x 32 spp, fused vs wavefront: 508 k-triangle / 6-BSDF scene 127 vs 202 ms, 4.9 k triangles 61 vs 153 ms,
    // Cornell box with mixed BSDFs 23 vs 70 ms, diffuse Cornell box 15 vs 35 ms)
    if (params->pipeline > 2) { rl_set_error("pipeline must be 0 (auto), 1 (wavefront) or 2 (fused)"); return RL_ERR_INVALID_ARGUMENT; }
    const bool fused = params->pipeline == 2 || (params->pipeline == 0 && params->pool_slots == 0);
    const bool fast_math = params->numerics == RL_NUMERICS_FAST;
    if (fast_math && !fused) { rl_set_error("numerics = fast exists for the persistent kernel only (pipeline 0 or 2, pool_slots 0)"); return RL_ERR_UNSUPPORTED; }
    // Reference-order streams through the persistent kernel run in TWO passes (chain.hip.h): k_stream_chain walks every block's stream with the
    // radiance half of the integrator left out and records the sampler state at the start of each camera sample, then the per-sample form of
    // k_path_fused evaluates all samples from those states with every lane busy.  Same image, same counters as the single-pass walk
    // (RL_REF_SINGLE_PASS=1 keeps that form: a test / measurement knob).
    bool two_pass = !per_sample && fused && !owned.empty() && !getenv("RL_REF_SINGLE_PASS");
    // the recorded states of ONE cursor position of every owned block must fit the budget (spp beyond ~90 000 at 1080p do not): else the single-pass walk
    size_t state_budget = (size_t)24 << 30;
    if (getenv("RL_STATE_BUDGET_MB")) state_budget = std::max<size_t>(1, (size_t)atoll(getenv("RL_STATE_BUDGET_MB"))) << 20;   // test knob: forces several chunks
    if (two_pass && (size_t)owned.size() * params->spp * 32 > state_budget) two_pass = false;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    // ---- how a set of work items is laid over the lanes.  per_pixel: pixel items (RL_STREAM_PER_SAMPLE, or the second pass of reference-order
    // streams), else one item per owned block.
    struct Plan { unsigned split, n_items, P, item_shift; };
    bool overlap_wanted = false;         // set once the chunks are known (below): the two-pass form's second pass will run beside the first
    auto plan_items = [&](bool per_pixel, unsigned n_pix, unsigned n_chains) -> Plan {
        // sample-parallel pixels: `split` lanes per pixel, per-sample radiances parked in HBM ([spp][pixel][3] floats) and folded
        // in order.  Auto: scenes that traverse out of L2 / HBM want ~8 M paths in flight per wavefront launch (measured on the
        // 508 k-triangle living-room scene at 1080p: 1 lane/pixel 935 ms, 4 lanes 787 ms, 8 lanes 836 ms); LDS-staged scenes in the
        // wavefront pipeline are VALU-bound and only pay for the extra state, so they stay at one lane per pixel.
        // The parking buffer is capped (kSampleBufBudget), beyond it one lane per pixel.
        Plan pl{1u, 0u, 0u, 0u};
        if (per_pixel && n_pix > 0) {
            // The persistent kernel keeps 4 x 256-lane workgroups per CU resident; a shard with few pixel tiles but many samples
            // per pixel (rank r of N at spp = 128 N: 1020 tiles at N = 8) leaves most of those slots empty once the tiles that
            // look past the scene have drained, so it is cut into >= ~16 k workgroups (measured, rank 0 of 8 at 1024 spp:
            // 1 lane / pixel 122 ms, 8 lanes 69 ms, 16 lanes 69 ms; a full 8160-tile frame is best left at 1 lane: 64 vs 67 ms).
            // Scenes that stream their BVH: all 64 lanes of a wave work on samples of ONE pixel (split = 64), so the camera rays of a wave are
            // nearly identical and fetch the same nodes (508 k triangles, 32 spp: 1 / 4 / 16 / 32 lanes per pixel = 108.9 / 104.0 / 100.8 / 97.4 ms).
            const unsigned fused_groups = (n_pix + 255u) / 256u;
            const unsigned fused_auto = fused_groups >= 6000u ? 1u : std::max(1u, 16384u / std::max(1u, fused_groups));
            unsigned want = params->sample_split ? params->sample_split
                                : (fused ? (ctx->lds_scene ? fused_auto : std::max(fused_auto, 64u))
                                         : (ctx->lds_scene ? 1u : std::max(1u, (8u << 20) / std::max(1u, n_pix))));
            // the evaluation pass beside the chain pass: its last launch — the blocks that completed last — is what is left to do when the chain pass ends, and a launch
            // lasts as long as its slowest pixel: several lanes per pixel cut that tail (1080p x 128 spp, 1 / 4 / 8 lanes: see profiles/NEGATIVES.md round 5)
            if (overlap_wanted && !params->sample_split) want = std::max(want, getenv("RL_EVAL_SPLIT") ? (unsigned)std::max(1, atoi(getenv("RL_EVAL_SPLIT"))) : 4u);
            pl.split = std::max(1u, std::min(want, params->spp));
            if ((size_t)n_pix * params->spp * 3 * sizeof(float) > kSampleBufBudget) pl.split = 1;
            while (pl.split > 1 && (size_t)n_pix * pl.split > (size_t)0x7fffff00u) pl.split--;
        }
        pl.n_items = per_pixel ? n_pix * pl.split : n_chains;
        // a pool never needs more slots than there are work items (and `pool_slots` is caller input: keep the rounding below from wrapping)
        unsigned P = params->pool_slots ? std::min(params->pool_slots, std::max(pl.n_items, 1u)) : std::min<unsigned>(pl.n_items, 16u << 20);
        P = std::max(256u, (unsigned)(((unsigned long long)P + 255ull) / 256ull * 256ull));
        if (fused) {
            // One lane per work item by default.  With a participating medium path lengths vary by orders of magnitude, and on scenes
            // that stream their BVH from L2 / HBM the cost per pixel varies as much, so there the grid is only what the chip keeps
            // resident (RL_FUSED_WAVES x 256-lane workgroups per CU) and lanes draw further items from the dispenser as they finish —
            // no workgroup idles behind its slowest pixel (cbox + medium, 32 spp: 185.5 -> 151.6 ms; 508 k triangles: 148.6 -> 126.9 ms;
            // LDS-staged scenes: plain cbox 63.5 vs 63.6 ms, mixed-BSDF cbox 23.3 vs 25.5 ms, so they keep the static tile order).
            const unsigned resident = (unsigned)cus * (unsigned)(ctx->lds_scene ? RL_FUSED_WAVES : RL_FUSED_WAVES_STREAMING) * 256u;
            const bool dynamic_items = getenv("RL_FUSED_DYNAMIC") ? atoi(getenv("RL_FUSED_DYNAMIC")) != 0 : (ctx->ds.medium.enabled != 0 || !ctx->lds_scene);
            P = std::max(256u, (std::min(pl.n_items, dynamic_items ? resident : pl.n_items) + 255u) / 256u * 256u);
            // sparse item sets (reference-order streams): one item per 2^item_shift lanes, all of them resident from the start
            const unsigned resident4 = (unsigned)cus * 4u * 256u;
            while (pl.item_shift < 6u && ((size_t)pl.n_items << (pl.item_shift + 1u)) <= resident4) pl.item_shift++;
            if (!per_pixel && getenv("RL_ITEM_SHIFT")) pl.item_shift = std::min(6u, (unsigned)atoi(getenv("RL_ITEM_SHIFT")));
            if (pl.item_shift) P = std::max(256u, (unsigned)((((size_t)pl.n_items << pl.item_shift) + 255u) / 256u * 256u));
        }
        pl.P = P;
        return pl;
    };
    // ---- chunks of the two-pass form: block cursors [c0, c1) of every owned block per chunk, sized so that the recorded sampler states
    // (32 B per camera sample) fit their budget; one chunk unless the render is very large (1080p x 128 spp = 8.5 GB).
    struct Chunk { unsigned c0, c1, n_pix; std::vector<unsigned> base; };
    std::vector<Chunk> chunks;
    // The state buffer is sized by what the device has free, not only by the fixed budget (several contexts or shards on one device, a smaller GPU):
    // the budget is cut to the buffer the context already holds + 60 % of the free memory, and if the allocation still fails it is halved until one
    // cursor position of every block no longer fits — then the single-pass walk, which needs no such buffer, renders the frame.
    if (two_pass && !getenv("RL_STATE_BUDGET_MB")) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) state_budget = std::min(state_budget, ctx->sample_states_capacity * sizeof(unsigned long long) + free_b / 10 * 6);
        else (void)hipGetLastError();
    }
    for (bool planned_states = false; two_pass && !planned_states;) {
        const size_t budget = state_budget;
        const size_t per_cursor = (size_t)owned.size() * params->spp * 32;       // bytes of states one cursor position of every block takes (upper bound)
        if (per_cursor > budget) { two_pass = false; chunks.clear(); break; }
        chunks.clear();
        const unsigned cursors_per_chunk = (unsigned)std::max<size_t>(1, std::min<size_t>(256, budget / std::max<size_t>(1, per_cursor)));
        for (unsigned c0 = 0; c0 < 256u; c0 += cursors_per_chunk) {
            Chunk ch; ch.c0 = c0; ch.c1 = std::min(256u, c0 + cursors_per_chunk); ch.n_pix = 0;
            for (size_t j = 0; j < owned.size(); j++) {
                const unsigned bidx = owned[j], bx = (unsigned)(bidx / nby) * 16u, by = (unsigned)(bidx % nby) * 16u;
                const unsigned npx = std::min(16u, W - bx) * std::min(16u, H - by);
                ch.base.push_back(ch.n_pix);
                ch.n_pix += std::min(ch.c1, npx) - std::min(ch.c0, npx);
            }
            if (ch.n_pix) chunks.push_back(std::move(ch));
        }
        size_t need = 0;
        for (const Chunk& ch : chunks) need = std::max(need, (size_t)ch.n_pix * params->spp * 4);
        if (ctx->sample_states_capacity >= need && ctx->d_sample_states) planned_states = true;
        else {
            if (ctx->d_sample_states) { hipFree(ctx->d_sample_states); ctx->d_sample_states = nullptr; ctx->sample_states_capacity = 0; }
            if (hipMalloc((void**)&ctx->d_sample_states, std::max<size_t>(need, 1) * sizeof(unsigned long long)) == hipSuccess) { ctx->sample_states_capacity = need; planned_states = true; }
            else { (void)hipGetLastError(); ctx->d_sample_states = nullptr; state_budget /= 2; }      // fewer cursors per chunk
        }
    }
    unsigned max_chunk_pix = 0;
    for (const Chunk& ch : chunks) max_chunk_pix = std::max(max_chunk_pix, ch.n_pix);
    overlap_wanted = two_pass && ctx->stream2 && chunks.size() == 1 && !fast_math && !getenv("RL_NO_OVERLAP");
    Plan plan = two_pass ? plan_items(true, max_chunk_pix, 0) : plan_items(per_sample, n_pixels, (unsigned)owned.size());   // two-pass: the largest second pass
    // (a SMALLER chunk can ask for MORE lanes per pixel, hence more slots and statistics rows than the largest one: uneven chunks at 1080p — 131 + 125 cursors — lost
    // 1 % of the counters and wrote past the rows; everything sized from `plan` below covers every chunk's own plan)
    if (two_pass) for (const Chunk& ch : chunks) { const Plan pc = plan_items(true, ch.n_pix, 0); plan.P = std::max(plan.P, pc.P); plan.split = std::max(plan.split, pc.split); }
    const Plan plan_chain = two_pass ? plan_items(false, 0, (unsigned)owned.size()) : Plan{1u, 0u, 0u, 0u};
    const unsigned split = plan.split, n_items = plan.n_items, item_shift = plan.item_shift;
    const unsigned P = std::max(plan.P, plan_chain.P);
    const unsigned n_item_pixels = two_pass ? max_chunk_pix : n_pixels;
    int rcode;
    if ((rcode = ensure(&ctx->d_owned, &ctx->owned_capacity, owned.size())) != RL_OK) return rcode;
    if ((rcode = ensure(&ctx->d_item_base, &ctx->item_base_capacity, owned.size())) != RL_OK) return rcode;
    if ((rcode = ensure(&ctx->d_block_seeds, &ctx->seeds_capacity, n_blocks)) != RL_OK) return rcode;
    if (per_sample || two_pass) {
        if (per_sample && (rcode = ensure(&ctx->d_item_seed, &ctx->item_capacity, n_item_pixels)) != RL_OK) return rcode;
        if ((rcode = ensure(&ctx->d_item_pixel, &ctx->item_pixel_capacity, n_item_pixels)) != RL_OK) return rcode;
    }
    if (two_pass) {
        if ((rcode = ensure(&ctx->d_sample_states, &ctx->sample_states_capacity, (size_t)max_chunk_pix * params->spp * 4)) != RL_OK) return rcode;
        if ((rcode = ensure(&ctx->d_chain_states, &ctx->chain_states_capacity, owned.size() * 4)) != RL_OK) return rcode;
    }
    if (!fused && ctx->pool_capacity < P) {
        if (ctx->pool.f) hipFree(ctx->pool.f);
        if (ctx->pool.u) hipFree(ctx->pool.u);
        if (ctx->pool.q) hipFree(ctx->pool.q);
        ctx->pool = Pool{};
        ctx->pool_capacity = 0;          // until all three planes exist: a failed allocation must not leave a half-built pool behind
        if (hipMalloc((void**)&ctx->pool.f, (size_t)F_COUNT * P * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&ctx->pool.u, (size_t)U_COUNT * P * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&ctx->pool.q, (size_t)Q_COUNT * P * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->pool.f) hipFree(ctx->pool.f);
            if (ctx->pool.u) hipFree(ctx->pool.u);
            if (ctx->pool.q) hipFree(ctx->pool.q);
            ctx->pool = Pool{};
            rl_set_error("out of device memory for a path-state pool of " + std::to_string(P) + " slots");
            return RL_ERR_HIP;
        }
        ctx->pool_capacity = P;
    }
    if (split > 1 && (rcode = ensure(&ctx->d_sample_buf, &ctx->sample_buf_capacity, (size_t)n_item_pixels * params->spp * 3)) != RL_OK) return rcode;
    Pool pool = ctx->pool;
    pool.P = P;
    const bool use_sort = !ctx->single_bsdf;
    float* d_out = out_rgb;
    if (!out_is_device) {
        if ((rcode = ensure(&ctx->d_out, &ctx->out_capacity, (size_t)3 * W * H)) != RL_OK) return rcode;
        d_out = ctx->d_out;
    }

    HIP_OK(hipMemcpyAsync(ctx->d_owned, owned.data(), owned.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    if (!two_pass) HIP_OK(hipMemcpyAsync(ctx->d_item_base, item_base.data(), item_base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(ctx->d_block_seeds, block_seeds, n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(d_out, 0, (size_t)3 * W * H * sizeof(float), st));
    Counters init{};
    init.active = std::min(plan.P, n_items);
    init.next_item = item_shift ? n_items : plan.P;
    if (!two_pass) HIP_OK(hipMemcpyAsync(ctx->d_counters, &init, sizeof(init), hipMemcpyHostToDevice, st));
    const size_t n_partial_rows = std::max<size_t>((P + 255) / 256, (size_t)cus * 8u * rl_context::kEvalStreams);      // (the queue-fed evaluation launches use a grid of the resident workgroups)
    if ((rcode = ensure(&ctx->d_partials, &ctx->partials_capacity, n_partial_rows * STAT_COUNT)) != RL_OK) return rcode;
    HIP_OK(hipMemsetAsync(ctx->d_partials, 0, n_partial_rows * STAT_COUNT * sizeof(unsigned long long), st));

    RenderConst rc{};
    rc.spp = params->spp;
    rc.has_min = params->has_min_depth; rc.min_depth = params->min_depth;
    rc.has_max = params->has_max_depth; rc.max_depth = params->max_depth;
    rc.has_rr = params->has_rr_depth; rc.rr_depth = params->rr_depth;
    rc.strategy = params->strategy; rc.single_scattering = params->single_scattering;
    rc.stream_mode = params->stream_mode; rc.seed_variant = params->seed_variant;
    rc.inv_spp = 1.0f / (float)params->spp;
    rc.W = W; rc.H = H; rc.nby = (unsigned)nby;
    rc.n_items = n_items;
    rc.item_shift = item_shift;
    rc.split = split; rc.sample_buf = ctx->d_sample_buf;
    rc.owned_blocks = ctx->d_owned; rc.block_item_base = ctx->d_item_base; rc.n_owned = (unsigned)owned.size();
    rc.block_seeds = ctx->d_block_seeds;
    rc.item_seed = ctx->d_item_seed; rc.item_pixel = ctx->d_item_pixel;
    rc.out = d_out;
    rc.counters = ctx->d_counters;
    rc.partials = ctx->d_partials;
    rc.sample_states = ctx->d_sample_states; rc.chain_states = ctx->d_chain_states;

    const DeviceScene& ds = ctx->ds;
    const dim3 block(256);
    const dim3 grid_all((plan.P + 255) / 256);
    // material-sort kernel: sparse pools (several lanes per pixel) are gathered four 256-slot chunks per workgroup
    const unsigned sort_chunks = split > 1 ? 4u : 1u;
    const dim3 grid_sort((P + 1023) / 1024);
    const dim3 grid_persistent(std::min<unsigned>((P + 255) / 256, (unsigned)cus * 8u));
    const bool medium = ds.medium.enabled != 0;
    const size_t lds_trav = traversal_lds_bytes(ctx, ctx->lds_scene, 256, true);
    StackConf stc;
    if ((rcode = stack_conf(ctx, (size_t)((P + 255) / 256) * 256, &stc)) != RL_OK) return rcode;

    if (per_sample && !owned.empty()) hipLaunchKernelGGL(k_seed_pixels, dim3(((unsigned)owned.size() + 63) / 64), dim3(64), 0, st, rc);
    if (!fused) hipLaunchKernelGGL(k_init, grid_all, block, 0, st, rc, pool);

    // events: 4 timed kernel classes per iteration
    const bool timing = stats != nullptr && !getenv("RL_NO_EVENTS");
    const size_t kEventsPerIter = 8;
    const unsigned poll_every = per_sample ? 8u : 32u;
    if (timing && ctx->events.size() < kEventsPerIter * poll_every) {
        size_t need = kEventsPerIter * poll_every;
        while (ctx->events.size() < need) { hipEvent_t ev; HIP_OK(hipEventCreate(&ev)); ctx->events.push_back(ev); }
    }
    double ms[4] = {0, 0, 0, 0};
    uint64_t iterations = 0, launches = 2, n_extend = 0;
    unsigned in_batch = 0;
    auto flush_events = [&](unsigned count) -> int {
        for (unsigned i = 0; i < count; i++)
            for (int k = 0; k < 4; k++) {
                float t = 0.0f;
                HIP_OK(hipEventElapsedTime(&t, ctx->events[kEventsPerIter * i + 2 * k], ctx->events[kEventsPerIter * i + 2 * k + 1]));
                ms[k] += t;
            }
        return RL_OK;
    };
    double ms_fused = 0.0, ms_chain = 0.0;
    unsigned long long spec_stat[3] = {0, 0, 0}; unsigned spec_group = 0;     // k_stream_spec's counters (rl_render_stats.reserved)
    // (LDS-staged scenes: k_path_fused stages the nodes as two-level records, 144 instead of 68 bytes each)
    const size_t lds_two_level_extra = (ctx->lds_scene && RL_LDS_TWO_LEVEL) ? (size_t)16 * (lds_scene2_float4s(ctx->ds.n_nodes, ctx->ds.n_prims) - lds_scene_float4s(ctx->ds.n_nodes, ctx->ds.n_prims)) : 0;
    const size_t lds_fused = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + lds_two_level_extra + kFusedColdBytes + ((ctx->lds_scene || RL_COOP_FETCH != 1) ? 0 : (size_t)4 * kCoopStageFloat4s * sizeof(float4));
    auto launch_fused = [&](const RenderConst& rcl, dim3 grid, hipStream_t on, const StackConf* stcl = nullptr) {
        if (rcl.queue_mode != 0u)       // (exact build only: `overlap` is off in the tolerance build)
            (ctx->lds_scene ? launch_fusedq_lds : launch_fusedq_stream)(ctx->single_bsdf ? ctx->bsdf_type : -1, medium, ctx->area_lights_only, grid, block, lds_fused, on, rcl, ds, stcl ? *stcl : stc);
        else
        (ctx->lds_scene ? (fast_math ? launch_fused_lds_fast : launch_fused_lds) : (fast_math ? launch_fused_stream_fast : launch_fused_stream))(ctx->single_bsdf ? ctx->bsdf_type : -1, medium, ctx->area_lights_only, grid, block, lds_fused, on, rcl, ds, stcl ? *stcl : stc);
    };
    if (two_pass) {
        // tiny LDS-staged scenes (the Cornell box: 19 nodes + 36 triangles): the lanes of a chain's group precompute its ray's node / triangle records
        // (trace.hip.h: precompute_records) — when a chain has at least 32 lanes to itself and the records fit a few passes
        StackConf stc_c = stc;
        // (not with a medium: most of its vertices are scattering events whose rays end in the volume, and the pass then costs more than it saves — cbox + medium,
        // 1080p x 16 spp: 404 vs 353 ms; a variant with ONE chain per wave, the records left in registers and a wave-uniform v_readlane walk measured no better
        // than the LDS records at the same chains per wave: 767.5 vs 768.1 ms, and 2 chains per wave beat both: 705 ms)
        if (ctx->lds_scene && !medium && ctx->ds.n_nodes <= 64u && ctx->ds.n_prims <= 64u && plan_chain.item_shift >= 5u && !getenv("RL_CHAIN_NO_PRE")) stc_c.pre_group = 1 << plan_chain.item_shift;
        // scenes that stream their BVH (exact build): the group fetches 16-node treelet blocks for its chain (traverse_treelet); 72 float4 of LDS per chain
        const bool treelets = !ctx->lds_scene && !fast_math && ctx->ds.nodes_t && ctx->ds.root_t >= 0 && plan_chain.item_shift >= 5u && ctx->ds.stack_depth <= 512u && !getenv("RL_CHAIN_NO_TREELETS");
        if (treelets) stc_c.pre_group = 1 << plan_chain.item_shift;
        // (both LDS-hungry forms are only taken when their workgroup fits: a very deep BVH or a large record set falls back to the plain per-node walk — ADVICE r3)
        {
            const size_t want = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + (treelets ? (size_t)(256 / stc_c.pre_group) * (72 + (ctx->ds.stack_depth + 1) / 2) * 16
                                : (stc_c.pre_group ? (size_t)(256 / stc_c.pre_group) * ((size_t)ctx->ds.n_nodes * 32 + (size_t)ctx->ds.n_prims * 8) : 0));
            if (want > ctx->lds_limit) stc_c.pre_group = 0;
        }
        const bool treelets_on = treelets && stc_c.pre_group != 0;
        const size_t lds_chain = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + (treelets_on ? (size_t)(256 / stc_c.pre_group) * (72 + (ctx->ds.stack_depth + 1) / 2) * 16
                                 : (stc_c.pre_group ? (size_t)(256 / stc_c.pre_group) * ((size_t)ctx->ds.n_nodes * 32 + (size_t)ctx->ds.n_prims * 8) : 0));
        // ---- k_stream_spec (spec.hip.h): the chains with every lane busy — exact build; RL_CHAIN_SERIAL=1 keeps the one-lane-per-block walk (the cross-check)
        bool spec = !fast_math && !getenv("RL_CHAIN_SERIAL");
        // Two walks of a pixel fall in with each other after about as many samples as a sample takes draws, so the speculative pass only pays when a
        // pixel has several times that many samples (spec.hip.h): decided from the draws per sample the context's last render measured, else from
        // what the scene is (a participating medium: ~150 draws per sample on the Cornell box — 1080p x 128 spp: 3085 vs 2837 ms; without one ~10).
        // RL_SPEC_FORCE=1 (tests): always, also inside the kernel.
        const bool spec_force = getenv("RL_SPEC_FORCE") != nullptr;
        if (spec && !spec_force) {
            const double nbar = ctx->draws_per_sample > 0.0 ? ctx->draws_per_sample : (medium ? 150.0 : 12.0);
            if ((double)params->spp < 4.0 * nbar) spec = false;
        }
        // its workgroup parks 30 words of state per lane + the groups' scratch in LDS on top of the scene and the stacks: where that exceeds what a workgroup may ask for on
        // this device the launch would be refused — the serial chain (k_stream_chain), whose workgroup is the plain traversal one, renders those scenes instead
        const size_t lds_spec = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + (size_t)kSpecColdWords * 256 * 4 + kSpecGroupLdsBytes;
        const size_t spec_lds_limit = getenv("RL_SPEC_LDS_LIMIT_TEST") ? (size_t)atoll(getenv("RL_SPEC_LDS_LIMIT_TEST")) : ctx->lds_limit;      // (test knob: a device with a smaller limit)
        if (lds_spec > spec_lds_limit) spec = false;
        SpecConf spc{};
        unsigned spec_threads = 0;
        unsigned long long spec_totals[32] = {0};
        if (spec) {
            // lanes per block: the fewest that still give every SIMD about two waves (a wider batch looks further ahead, so its windows are wider)
            unsigned group = 16u;
            while (group < 64u && (size_t)owned.size() * group < (size_t)cus * 4u * 2u * 64u) group <<= 1;
            // scenes that stream their BVH are bound by the latency of a wave's dependent fetches: one block per wave, four lanes per pixel
            // (508 k triangles, 1080p x 128 spp: 32 x 1 / 64 x 4 = 4377 / 2667 ms; the serial chain 5689 ms); LDS-staged ones: two lanes per pixel from 32 lanes per block on
            if (!ctx->lds_scene) group = 64u;
            if (getenv("RL_SPEC_GROUP")) { const int g = atoi(getenv("RL_SPEC_GROUP")); if (g == 16 || g == 32 || g == 64 || g == 256) group = (unsigned)g; }
            spc.group = group;
            // sixteen pixels per batch whatever the group: a longer look-ahead widens every window and misses more often (shard 0 of 8 at 1024 spp, 64 lanes per block,
            // 1 / 2 / 4 lanes per pixel: 2621 / 1235 / 731 ms; full frame at 128 spp, 32 lanes: 2 lanes per pixel 265, 4: 705)
            spc.sub = std::max(1u, group / 16u);
            spc.serial_ratio = spec_force ? 0.0f : (getenv("RL_SPEC_SERIAL_RATIO") ? (float)atof(getenv("RL_SPEC_SERIAL_RATIO")) : 3.0f);
            if (getenv("RL_SPEC_SUB")) { const int v = atoi(getenv("RL_SPEC_SUB")); if ((v == 1 || v == 2 || v == 4 || v == 8 || v == 16) && (unsigned)v <= group) spc.sub = (unsigned)v; }
            spc.cap = std::max(96u, std::min(3u * params->spp + 64u, 1u << 20));
            if (getenv("RL_SPEC_CAP")) spc.cap = std::max(4u, (unsigned)atoi(getenv("RL_SPEC_CAP")));
            spc.probe = getenv("RL_SPEC_PROBE") ? (unsigned)atoi(getenv("RL_SPEC_PROBE")) : std::min(32u, std::max(4u, params->spp));
            spc.lead = getenv("RL_SPEC_LEAD") ? (unsigned)atoi(getenv("RL_SPEC_LEAD")) : 24u;
            spc.lead_max = getenv("RL_SPEC_LEAD_MAX") ? (unsigned)atoi(getenv("RL_SPEC_LEAD_MAX")) : 128u;
            spc.lead_var = getenv("RL_SPEC_LEAD_VAR") ? (float)atof(getenv("RL_SPEC_LEAD_VAR")) : 100.0f;      // (cbox 1080p x 128 spp: 265.5 -> 260.2 ms; probing every batch: 319.6 ms)
            spc.extra = getenv("RL_SPEC_EXTRA") ? (unsigned)atoi(getenv("RL_SPEC_EXTRA")) : 0u;      // (cbox 1080p x 128 spp: 2.84 M instead of 3.57 M serial samples, 383 M instead of 306 M walked: 261 vs 259 ms — a wash, off)
            // (cbox 1080p x 128 spp, three workgroups per CU: 244 -> 198 ms; 8 / 16 / 32 lanes alike.  Scenes that stream their BVH: 2452 -> 2517 ms on the 508 k-triangle scene — a helper's sample is a
            // chain of dependent fetches like any other there — so off)
            spc.dense = getenv("RL_SPEC_DENSE") ? std::min(64u, (unsigned)atoi(getenv("RL_SPEC_DENSE"))) : (ctx->lds_scene ? 16u : 0u);
            spc.dense_frac = getenv("RL_SPEC_DENSE_FRAC") ? (float)atof(getenv("RL_SPEC_DENSE_FRAC")) : 0.6f;
            spc.probe_every = getenv("RL_SPEC_PROBE_EVERY") ? (unsigned)atoi(getenv("RL_SPEC_PROBE_EVERY")) : 0u;
            // window margins in standard deviations of the predicted offsets: with one block per wave a pixel the chain has to be walked through stalls the whole wave, so wider
            // (shard 0 of 8, 1024 spp: 1.65 / 2.5 sigma = 714 / 688 ms; full frame, two blocks per wave: 281 / 292)
            spc.ks = getenv("RL_SPEC_KS") ? (float)atof(getenv("RL_SPEC_KS")) : (group >= 64u ? 2.5f : 1.65f);
            spc.ke = getenv("RL_SPEC_KE") ? (float)atof(getenv("RL_SPEC_KE")) : (group >= 64u ? 2.5f : 1.65f);
            spec_threads = (unsigned)((((size_t)owned.size() * group) + 255u) / 256u * 256u);
            // the tracks: 36 B per entry; when they do not fit what the device has free the serial walk runs instead
            const size_t need = (size_t)spec_threads * spc.cap * 36u;
            size_t free_b = 0, total_b = 0;
            const size_t have = ctx->trk_off_capacity * 4u + ctx->trk_st_capacity * 16u;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            if (need > have && need - have > free_b / 2u) spec = false;
            if (spec && (ensure(&ctx->d_trk_off, &ctx->trk_off_capacity, (size_t)spec_threads * spc.cap) != RL_OK ||
                         ensure(&ctx->d_trk_st, &ctx->trk_st_capacity, (size_t)spec_threads * spc.cap * 2u) != RL_OK)) { (void)hipGetLastError(); spec = false; }
        }
        if (spec) {
            // the masks depend on the camera, the scene bounds and the shard only: computed once per context and shard
            const bool expand = !params->has_max_depth || 1u < params->max_depth;
            const uint64_t key = ((uint64_t)params->shard_index << 33) | ((uint64_t)shard_count << 1) | (expand ? 1u : 0u);
            if (key != ctx->trivial_key || ctx->trivial_capacity < owned.size() * 8 || getenv("RL_SPEC_NO_TRIVIAL")) {
                std::vector<unsigned> masks;
                trivial_pixel_masks(trivial_input(ctx), params, owned, nby, &masks);
                if ((rcode = ensure(&ctx->d_trivial, &ctx->trivial_capacity, masks.size())) != RL_OK) return rcode;
                HIP_OK(hipMemcpyAsync(ctx->d_trivial, masks.data(), masks.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
                HIP_OK(hipStreamSynchronize(st));        // (`masks` is a local)
                ctx->trivial_key = getenv("RL_SPEC_NO_TRIVIAL") ? ~0ull : key;
            }
            if ((rcode = ensure(&ctx->d_spec_stats, &ctx->spec_stats_capacity, 32 + 8 * (size_t)(spec_threads / 64u))) != RL_OK) return rcode;
            HIP_OK(hipMemsetAsync(ctx->d_spec_stats, 0, 32 * sizeof(unsigned long long), st));
            spc.trk_off = ctx->d_trk_off; spc.trk_st = ctx->d_trk_st; spc.trivial = ctx->d_trivial;
            spc.stats = (stats || getenv("RL_SPEC_STATS")) ? ctx->d_spec_stats : nullptr;
        }
        StackConf stc_s = stc;
        if (spec && (rcode = stack_conf(ctx, std::max<size_t>((size_t)((P + 255) / 256) * 256, spec_threads), &stc_s)) != RL_OK) return rcode;
        if (spec) stc = stc_s;      // (one overflow buffer serves both passes: the stride is the larger launch)
        // ---- the evaluation pass overlapped with the chain pass: one chunk, exact build, a group of the speculative pass inside one wave (the wave that pushes a block
        // is the wave that wrote its states)
        const bool overlap = overlap_wanted && !(spec && spc.group > 64u);
        for (const Chunk& ch : chunks) {
            HIP_OK(hipMemcpyAsync(ctx->d_item_base, ch.base.data(), ch.base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
            RenderConst ra = rc;
            ra.stream_mode = RL_STREAM_REFERENCE_ORDER;
            ra.n_items = plan_chain.n_items; ra.item_shift = plan_chain.item_shift; ra.split = 1;
            ra.n_state_pixels = ch.n_pix; ra.cursor_begin = ch.c0; ra.cursor_end = ch.c1;
            const unsigned chain_grid = spec ? spec_threads / 256u : (plan_chain.P + 255u) / 256u;
            constexpr unsigned kMaxEvalLaunches = 256u;
            if (overlap) {
                // device: [0] started workgroups, [16 + k] the claim counter of the k-th evaluation launch, [16 + kMaxEvalLaunches + i] the block lists (completion order);
                // mapped host memory: [0] "every chain workgroup runs", [16 + j] "block j is complete"; pinned staging of the lists
                const size_t n_dev = 16 + (size_t)kMaxEvalLaunches + owned.size(), n_flags = 16 + owned.size();
                if ((rcode = ensure(&ctx->d_queue, &ctx->done_queue_capacity, n_dev)) != RL_OK) return rcode;
                if (ctx->flags_capacity < n_flags) {
                    if (ctx->h_flags) hipHostFree(ctx->h_flags);
                    ctx->h_flags = nullptr; ctx->d_flags = nullptr; ctx->flags_capacity = 0;
                    HIP_OK(hipHostMalloc((void**)&ctx->h_flags, n_flags * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
                    HIP_OK(hipHostGetDevicePointer((void**)&ctx->d_flags, ctx->h_flags, 0));
                    std::memset(ctx->h_flags, 0, n_flags * sizeof(unsigned));
                    ctx->flags_capacity = n_flags;
                }
                if (ctx->list_capacity < owned.size()) {
                    if (ctx->h_list) hipHostFree(ctx->h_list);
                    ctx->h_list = nullptr; ctx->list_capacity = 0;
                    HIP_OK(hipHostMalloc((void**)&ctx->h_list, owned.size() * sizeof(unsigned), hipHostMallocDefault));
                    ctx->list_capacity = owned.size();
                }
                HIP_OK(hipMemsetAsync(ctx->d_queue, 0, (16 + (size_t)kMaxEvalLaunches) * sizeof(unsigned), st));
                ra.queue = ctx->d_queue; ra.chain_grid = chain_grid;
                ra.done_flags = ctx->d_flags + 16; ra.started_flag = ctx->d_flags;
                ra.queue_seq = ++ctx->queue_seq;
                if (ra.queue_seq == 0u) { std::memset(ctx->h_flags, 0, ctx->flags_capacity * sizeof(unsigned)); ra.queue_seq = ++ctx->queue_seq; }      // (the tag wrapped: 0 is the words' idle value)
            }
            hipLaunchKernelGGL(k_chunk_pixels, dim3(((unsigned)owned.size() + 63) / 64), dim3(64), 0, st, ra);
            // ---- pass 2 (planned before pass 1 is launched: overlapped, it starts beside it): every camera sample of the chunk from its recorded state, per-pixel work items
            const Plan pb = plan_items(true, ch.n_pix, 0);
            RenderConst rb = ra;
            rb.stream_mode = kStreamGivenStates;
            rb.n_items = pb.n_items; rb.item_shift = pb.item_shift; rb.split = pb.split;
            rb.queue = nullptr; rb.queue_mode = 0u; rb.done_flags = nullptr; rb.started_flag = nullptr;
            Counters cinit{};
            cinit.active = std::min(pb.P, pb.n_items);
            cinit.next_item = pb.item_shift ? pb.n_items : pb.P;
            if (overlap) cinit.next_item = pb.n_items;          // (queue-fed: the dispenser has nothing to hand out)
            HIP_OK(hipMemcpyAsync(ctx->d_counters, &cinit, sizeof(cinit), hipMemcpyHostToDevice, st));
            // (overlapped: a grid of the workgroups the chip keeps resident; its overflow stack levels are its own — the chain kernel beside it spills into the context's
            // first buffer under the same thread indices; allocated before anything is launched: an allocation may wait for the device)
            const dim3 grid_q((unsigned)cus * (unsigned)(ctx->lds_scene ? RL_FUSED_WAVES : RL_FUSED_WAVES_STREAMING));
            StackConf stc_q = stc;
            if (overlap && (rcode = stack_conf(ctx, (size_t)grid_q.x * 256 * rl_context::kEvalStreams, &stc_q, true)) != RL_OK) return rcode;      // (one column range per evaluation stream)
            // ---- pass 1: the chains
            if (timing) hipEventRecord(ctx->events[0], st);
            if (spec) (ctx->lds_scene ? launch_spec_lds : launch_spec_stream)(ctx->single_bsdf ? ctx->bsdf_type : -1, medium, dim3(spec_threads / 256u), block, lds_spec, st, ra, ds, stc, spc);
            else
            (ctx->lds_scene ? (fast_math ? launch_chain_lds_fast : launch_chain_lds) : (fast_math ? launch_chain_stream_fast : launch_chain_stream))(ctx->single_bsdf ? ctx->bsdf_type : -1, medium, dim3((plan_chain.P + 255) / 256), block, lds_chain, st, ra, ds, stc_c);
            if (timing) hipEventRecord(ctx->events[1], st);
            if (overlap) {
                // Overlapped: k_path_fused<.., QUEUE = true> on the context's low-priority streams WHILE the chain pass runs on `st`.  Nothing on the device waits for anything:
                // the chain kernels flag completed blocks in mapped host memory, THIS host thread (the call is synchronous anyway) collects them and launches the evaluation
                // kernel over explicit lists of complete blocks — not before the chain kernel has reported that every one of its workgroups runs (so the launches beside it
                // only take resources it has no further use for), then whenever a fifth of the blocks still to come have come in (at least 64: a launch lasts as long as its
                // slowest pixel and should fill a good part of the chip), the rest when the chain pass has ended.  Same samples from the same states, folded per pixel in
                // sample order: same bits (RL_NO_OVERLAP=1 keeps the two passes back to back: the cross-check).
                rb.queue = ctx->d_queue; rb.queue_mode = 1u; rb.chain_grid = chain_grid;
                HIP_OK(hipEventRecord(ctx->ev_chain_done, st));
                const unsigned seq = ra.queue_seq, n_blocks_owned = (unsigned)owned.size();
                volatile unsigned* hf = ctx->h_flags;
                std::vector<unsigned> waiting(n_blocks_owned);          // owned blocks not listed yet (the scan below only looks at these; it shrinks as blocks complete)
                for (unsigned j = 0; j < n_blocks_owned; j++) waiting[j] = j;
                unsigned n_listed = 0, n_launches = 0, scan_from = 0;
                // an error in here must not leave kernels running on the context's streams behind the caller's back
                auto drain = [&]() { (void)hipStreamSynchronize(st); for (int k = 0; k < rl_context::kEvalStreams; k++) (void)hipStreamSynchronize(ctx->eval_streams[k]); (void)hipGetLastError(); };
                // the poll interval grows with the number of flags a poll reads (50 us for a 1080p frame's 8160 blocks, ~1 ms per 100 k blocks)
                const auto poll_us = std::chrono::microseconds(50 + n_blocks_owned / 100u);
                bool chain_over = false, started = false;
                const unsigned resident = grid_q.x;
                // (a launch when a 1 / batch_div of the blocks still to come — at least batch_min — have come in; RL_EVAL_MIN / RL_EVAL_DIV: dev knobs of the sweep in NEGATIVES round 5)
                const unsigned batch_min = getenv("RL_EVAL_MIN") ? (unsigned)std::max(1, atoi(getenv("RL_EVAL_MIN"))) : 64u, batch_div = getenv("RL_EVAL_DIV") ? (unsigned)std::max(1, atoi(getenv("RL_EVAL_DIV"))) : 5u;
                auto launch_batch = [&](unsigned first, unsigned count) -> int {
                    const unsigned k = n_launches % (unsigned)rl_context::kEvalStreams;
                    hipStream_t on = ctx->eval_streams[k];
                    HIP_OK(hipMemcpyAsync(ctx->d_queue + 16 + kMaxEvalLaunches + first, ctx->h_list + first, count * sizeof(unsigned), hipMemcpyHostToDevice, on));
                    RenderConst rq = rb;
                    rq.q_list = ctx->d_queue + 16 + kMaxEvalLaunches + first; rq.q_n = count; rq.q_ctr = ctx->d_queue + 16 + n_launches;
                    // launches on different streams run side by side: each stream has its own statistics rows and its own columns of the overflow stack levels
                    rq.partials = rb.partials + (size_t)k * resident * STAT_COUNT;
                    StackConf stc_k = stc_q;
                    if (stc_k.overflow) stc_k.overflow += (size_t)k * resident * 256u * 2u;      // ([level][thread] pairs of ints)
                    const unsigned wgs = std::min<unsigned>(resident, (unsigned)(((size_t)count * 256u * pb.split + 255u) / 256u));
                    launch_fused(rq, dim3(std::max(1u, wgs)), on, &stc_k);
                    HIP_OK(hipGetLastError());
                    n_launches++; launches++;
                    return RL_OK;
                };
                while (n_listed < n_blocks_owned) {
                    if (!chain_over) {
                        const hipError_t qe = hipEventQuery(ctx->ev_chain_done);
                        if (qe == hipSuccess) chain_over = true;
                        else if (qe != hipErrorNotReady) { rl_set_error(std::string("hipEventQuery(chain pass): ") + hipGetErrorString(qe)); (void)hipGetLastError(); drain(); return RL_ERR_HIP; }
                    }
                    if (!started && hf[0] == seq) started = true;
                    if (started || chain_over) {    // newly flagged blocks, appended in the order found (once the chain pass is over every block is complete)
                        size_t keep = 0;
                        for (size_t w = 0; w < waiting.size(); w++) {
                            const unsigned j = waiting[w];
                            if (chain_over || hf[16 + j] == seq) ctx->h_list[n_listed++] = j; else waiting[keep++] = j;
                        }
                        waiting.resize(keep);
                    }
                    const unsigned fresh = n_listed - scan_from, to_come = n_blocks_owned - scan_from;
                    if (fresh > 0 && (chain_over || (fresh >= std::max(batch_min, to_come / batch_div) && n_launches + 2u < kMaxEvalLaunches))) {
                        if ((rcode = launch_batch(scan_from, fresh)) != RL_OK) { drain(); return rcode; }
                        scan_from = n_listed;
                    }
                    if (n_listed < n_blocks_owned && !chain_over) std::this_thread::sleep_for(poll_us);
                }
                if (scan_from < n_listed && (rcode = launch_batch(scan_from, n_listed - scan_from)) != RL_OK) { drain(); return rcode; }      // the blocks listed last
                if (getenv("RL_QUEUE_DEBUG")) std::fprintf(stderr, "[queue] %u evaluation launches beside / after the chain pass, %u blocks, started flag %s\n", n_launches, n_listed, started ? "seen" : "not seen");
                HIP_OK(hipStreamSynchronize(st));                      // the chain pass (over already: every block was flagged or its event had fired)
                for (int k = 0; k < rl_context::kEvalStreams; k++) HIP_OK(hipStreamSynchronize(ctx->eval_streams[k]));      // the last evaluation launches
                if (timing) hipEventRecord(ctx->events[3], st);
            } else {
                if (timing) hipEventRecord(ctx->events[2], st);
                launch_fused(rb, dim3((pb.P + 255) / 256), st);
                if (timing) hipEventRecord(ctx->events[3], st);
            }
            if (pb.split > 1) hipLaunchKernelGGL(k_fold_samples, dim3((ch.n_pix + 255) / 256), block, 0, st, rb);
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(st));       // (the chunk's host arrays and the counters block are reused by the next chunk)
            if (timing) {
                float t = 0.0f;
                HIP_OK(hipEventElapsedTime(&t, ctx->events[0], ctx->events[1])); ms_chain += t;
                // overlapped: what the evaluation pass still takes AFTER the chain pass has ended (the part of it that is not hidden)
                HIP_OK(hipEventElapsedTime(&t, ctx->events[overlap ? 1 : 2], ctx->events[3])); ms_fused += t;
            }
            launches += 3 + (pb.split > 1 ? 1 : 0);
        }
        dump_stage_timers(ctx->lds_scene);
        if (ctx->lds_scene) dump_chain_timers_lds(); else dump_chain_timers_stream();     // dev-only build
        if (spec && spc.stats) {
            HIP_OK(hipMemcpy(spec_totals, ctx->d_spec_stats, sizeof(spec_totals), hipMemcpyDeviceToHost));
            spec_group = spc.group; spec_stat[0] = spec_totals[0]; spec_stat[1] = spec_totals[1]; spec_stat[2] = spec_totals[2];
            if (getenv("RL_SPEC_STATS")) std::fprintf(stderr, "[spec] LDS per workgroup %zu bytes (scene %zu, %d stack levels)\n", traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + (size_t)kSpecColdWords * 256 * 4 + kSpecGroupLdsBytes, (size_t)(ctx->lds_scene ? ctx->scene_lds_bytes : 0), lds_levels_of(ctx));
            if (getenv("RL_SPEC_STATS")) std::fprintf(stderr, "[spec] group %u x sub %u cap %u: %llu speculative + %llu serial + %llu probe samples for %llu camera samples (%.2f x, %.2f serial per pixel), %llu wave iterations\n",
                spc.group, spc.sub, spc.cap, spec_totals[0], spec_totals[1], spec_totals[2], (unsigned long long)n_pixels * params->spp,
                (double)(spec_totals[0] + spec_totals[1] + spec_totals[2]) / std::max(1.0, (double)n_pixels * params->spp), (double)spec_totals[1] / std::max(1u, n_pixels), spec_totals[3]);
            if (getenv("RL_SPEC_WAVE_TIMES") && spec_totals[8]) {     // dev build: lifetime of every wave (100 MHz clock)
                std::vector<unsigned long long> wt(8 * (size_t)(spec_threads / 64u));
                HIP_OK(hipMemcpy(wt.data(), ctx->d_spec_stats + 32, wt.size() * 8, hipMemcpyDeviceToHost));
                FILE* f = std::fopen(getenv("RL_SPEC_WAVE_TIMES"), "w");
                if (f) { unsigned long long t0 = ~0ull; for (size_t w = 0; w < wt.size() / 8; w++) if (wt[8 * w]) t0 = std::min(t0, wt[8 * w]);
                         for (size_t w = 0; w < wt.size() / 8; w++) std::fprintf(f, "%zu %.3f %.3f %llu %llu %.3f %.3f %llu %llu\n", w, (wt[8 * w] - t0) * 1e-5, (wt[8 * w + 1] - t0) * 1e-5, wt[8 * w + 2], wt[8 * w + 3], wt[8 * w + 4] * 1e-5, wt[8 * w + 5] * 1e-5, wt[8 * w + 6], wt[8 * w + 7]); std::fclose(f); }
            }
            if (getenv("RL_SPEC_STATS") && spec_totals[8]) {     // dev build (-DRL_SPEC_TIMERS)
                const double tot = (double)(spec_totals[4] + spec_totals[5] + spec_totals[6] + spec_totals[7] + spec_totals[8]);
                std::fprintf(stderr, "[spec] cycles: bookkeeping %.1f %%, plan %.1f %%, thread %.1f %%, copy-out %.1f %%, extend+shade %.1f %%; %.1f lanes per traced iteration, %.1f %% of the traced iterations serial only, %.0f cycles per wave iteration\n",
                    100.0 * spec_totals[4] / tot, 100.0 * spec_totals[5] / tot, 100.0 * spec_totals[6] / tot, 100.0 * spec_totals[7] / tot, 100.0 * spec_totals[8] / tot,
                    (double)spec_totals[9] / std::max<double>(1.0, (double)spec_totals[10]), 100.0 * spec_totals[11] / std::max<double>(1.0, (double)spec_totals[10]), tot / std::max<double>(1.0, (double)spec_totals[3]));
                { const double ts = (double)(spec_totals[16] + spec_totals[17] + spec_totals[18] + spec_totals[19] + spec_totals[20]);
                  std::fprintf(stderr, "[spec] iterations after a serial-only one (%.1f %% of all cycles): bookkeeping %.1f %%, plan %.1f %%, thread %.1f %%, copy-out %.1f %%, extend+shade %.1f %%\n", 100.0 * ts / tot,
                    100.0 * spec_totals[16] / ts, 100.0 * spec_totals[17] / ts, 100.0 * spec_totals[18] / ts, 100.0 * spec_totals[19] / ts, 100.0 * spec_totals[20] / ts); }
                std::fprintf(stderr, "[spec] slow walks: %llu from a pixel's start, %llu across a missing link, %llu past the last track\n", spec_totals[12], spec_totals[13], spec_totals[14]);
                std::fprintf(stderr, "[spec] serial walks on helpers: %llu (one lane: %llu), %llu rounds, %llu samples taken from them\n", spec_totals[26], spec_totals[27], spec_totals[24], spec_totals[25]);
            }
        }
        iterations = chunks.size();
    } else if (fused) {
        if (timing) hipEventRecord(ctx->events[0], st);
        launch_fused(rc, grid_all, st);
        if (timing) hipEventRecord(ctx->events[1], st);
        HIP_OK(hipGetLastError());          // a refused launch configuration is not sticky: without this the sync below would "succeed"
        HIP_OK(hipStreamSynchronize(st));
        if (timing) { float t = 0.0f; HIP_OK(hipEventElapsedTime(&t, ctx->events[0], ctx->events[1])); ms_fused = t; }
        dump_stage_timers(ctx->lds_scene);     // dev-only build (-DRL_STAGE_TIMERS): per-stage cycle shares of the fused loop
        launches += 1;
        iterations = 1;
    } else for (;;) {
        hipEvent_t* ev = timing ? &ctx->events[kEventsPerIter * in_batch] : nullptr;
        if (timing) hipEventRecord(ev[0], st);
        hipLaunchKernelGGL(k_raygen, grid_persistent, block, 0, st, rc, ds, pool);
        if (timing) { hipEventRecord(ev[1], st); hipEventRecord(ev[2], st); }
        if (ctx->lds_scene) hipLaunchKernelGGL((k_extend<true>), grid_all, block, lds_trav, st, rc, ds, pool, stc);
        else hipLaunchKernelGGL((k_extend<false>), grid_all, block, lds_trav, st, rc, ds, pool, stc);
        if (timing) { hipEventRecord(ev[3], st); hipEventRecord(ev[4], st); }
        if (use_sort) launch_shade_sorted(medium, sort_chunks, sort_chunks == 4u ? grid_sort : grid_all, block, st, rc, ds, pool);
        else launch_shade_type(ctx->bsdf_type, medium, grid_all, block, st, rc, ds, pool);
        launches += 1;
        if (timing) { hipEventRecord(ev[5], st); hipEventRecord(ev[6], st); }
        if (ctx->lds_scene) hipLaunchKernelGGL((k_shadow<true>), grid_all, block, lds_trav, st, rc, ds, pool, stc);
        else hipLaunchKernelGGL((k_shadow<false>), grid_all, block, lds_trav, st, rc, ds, pool, stc);
        if (timing) hipEventRecord(ev[7], st);
        launches += 3;
        n_extend++;
        iterations++;
        in_batch++;
        if (in_batch == poll_every) {
            HIP_OK(hipGetLastError());      // launch-configuration errors of the batch (non-sticky): never spin on a counter no kernel updates
            HIP_OK(hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(Counters), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            if (timing) { int r = flush_events(in_batch); if (r != RL_OK) return r; }
            in_batch = 0;
            if (ctx->h_counters->active == 0) break;
        }
        if (iterations > (uint64_t)1 << 28) { rl_set_error("render did not terminate"); return RL_ERR_HIP; }
    }
#ifdef RL_TRAV_STATS
    if (!fused) {
        unsigned long long h[8];
        HIP_OK(hipStreamSynchronize(st));
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trav_stats), sizeof(h));
        std::fprintf(stderr, "[trav] rays %llu (zero-step %.1f %%)  steps/ray %.1f  tris/ray %.1f  waves %llu  lanes/wave %.1f  paid steps/ray %.1f  step utilisation %.1f %%\n",
                     h[0], 100.0 * h[5] / (double)h[0], (double)h[1] / h[0], (double)h[2] / h[0], h[4], (double)h[0] / h[4], (double)h[3] / h[0], 100.0 * h[1] / (double)h[3]);
        std::memset(h, 0, sizeof(h)); hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), h, sizeof(h));
    }
#endif
    // one more raygen pass is never needed: `active` reaches 0 inside k_raygen after the last fold.
    if (split > 1 && !two_pass) { hipLaunchKernelGGL(k_fold_samples, dim3((n_pixels + 255) / 256), block, 0, st, rc); launches += 1; }
    if (!out_is_device) HIP_OK(hipMemcpyAsync(out_rgb, d_out, (size_t)3 * W * H * sizeof(float), hipMemcpyDeviceToHost, st));
    std::vector<unsigned long long> partials(n_partial_rows * STAT_COUNT);
    HIP_OK(hipMemcpyAsync(partials.data(), ctx->d_partials, partials.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    unsigned long long totals[STAT_COUNT] = {0};
    for (size_t r = 0; r < n_partial_rows; r++) for (int k = 0; k < STAT_COUNT; k++) totals[k] += partials[r * STAT_COUNT + k];
    HIP_OK(hipGetLastError());
    auto t_end = std::chrono::steady_clock::now();
    if (totals[STAT_SAMPLES]) ctx->draws_per_sample = (double)totals[STAT_DRAWS] / (double)totals[STAT_SAMPLES];
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->camera_samples = totals[STAT_SAMPLES];
        stats->vertices = totals[STAT_VERTICES];
        stats->extension_rays = totals[STAT_EXT_RAYS];
        stats->shadow_rays = totals[STAT_SHADOW_RAYS];
        stats->rng_draws = totals[STAT_DRAWS];
        stats->iterations = iterations;
        stats->kernel_launches = launches;
        stats->render_ms = std::chrono::duration<double, std::milli>(t_end - t_start).count();
        stats->ms_raygen = ms[0]; stats->ms_extend = ms[1]; stats->ms_shade = ms[2]; stats->ms_shadow = ms[3];
        stats->ms_other = ms_fused;   // the persistent fused kernel (pipeline 2)
        stats->ms_prepass = ms_chain;  // k_stream_chain (reference-order streams, first pass)
        stats->n_extend_launches = n_extend;
        stats->reserved[0] = spec_stat[0]; stats->reserved[1] = spec_stat[1]; stats->reserved[2] = spec_stat[2]; stats->reserved[3] = spec_group;   // speculative / serial / probe samples of k_stream_spec, its lanes per block (0: the serial chain ran)
    }
    return RL_OK;
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
    size_t state_budget = (size_t)24 << 30;
    if (getenv("RL_STATE_BUDGET_MB")) state_budget = std::max<size_t>(1, (size_t)atoll(getenv("RL_STATE_BUDGET_MB"))) << 20;
    const bool two_pass = !per_sample && !owned.empty() && !getenv("RL_REF_SINGLE_PASS") && (size_t)n_pixels * params->spp * 32 <= state_budget;
    const bool per_pixel = per_sample || two_pass;
    const unsigned n_items = per_pixel ? n_pixels : (unsigned)owned.size();
    unsigned chain_shift = 0;
    if (two_pass) {
        int cus = 256;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        while (chain_shift < 6u && (owned.size() << (chain_shift + 1u)) <= (size_t)cus * 4u * 256u) chain_shift++;
        if (getenv("RL_ITEM_SHIFT")) chain_shift = std::min(6u, (unsigned)atoi(getenv("RL_ITEM_SHIFT")));
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
    const bool timing = stats != nullptr && !getenv("RL_NO_EVENTS");
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
