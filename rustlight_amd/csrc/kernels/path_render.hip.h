// path_render.hip.h — the host driver behind rl_render_path: one PathRender object per call (included by wavefront.hip after rl_context and the
// launch helpers; host code only).  Integrator::compute for IntegratorPathTracing (src/integrators/mod.rs:219-233 -> compute_mc, 403-450).
//
// The call in steps — each a method, none of them reads the environment (the context's options were copied when the object was built: knobs.h):
//   decompose()        this shard's blocks in creation order (mod.rs:351-374: `b % shard_count`)
//   choose_form()      wavefront stage kernels | persistent kernel; reference-order streams in two passes (chain.hip.h) or one
//   plan_chunks()      two passes: block cursors per chunk so that the recorded sampler states fit their buffer
//   plan_lanes()       how pixel items / block chains are laid over the lanes (lanes per pixel, pool slots, item spread), buffers
//   upload()           the shard's tables, the zeroed framebuffer, RenderConst
//   run_two_pass()     per chunk: plan_chain_pass() -> chain pass (k_stream_spec | k_stream_chain) with the evaluation pass beside it
//                      (overlap_loop(): host-driven launches over lists of complete blocks) or after it, then the fold of parked samples
//   run_fused() / run_wavefront()
//   finish()           framebuffer download, statistics rows, rl_render_stats
#pragma once

namespace {

using namespace rl;

struct Plan { unsigned split, n_items, P, item_shift; };
struct Chunk { unsigned c0, c1, n_pix; std::vector<unsigned> base; };

class PathRender {
public:
    PathRender(rl_context* c, const rl_path_params* p, const uint64_t* seeds, size_t nb, float* out, int out_dev, hipStream_t stream, rl_render_stats* s)
        : ctx(c), params(p), block_seeds(seeds), n_blocks(nb), out_rgb(out), out_is_device(out_dev), st(stream), stats(s), knobs(c->knobs),
          W(c->width), H(c->height), nby((c->height + 15) / 16), shard_count(p->shard_count ? p->shard_count : 1) {}

    int run() {
        t_start = std::chrono::steady_clock::now();
        decompose();
        int r;
        if ((r = choose_form()) != RL_OK) return r;
        plan_chunks();
        if ((r = plan_lanes()) != RL_OK) return r;
        if ((r = upload()) != RL_OK) return r;
        if (two_pass) r = run_two_pass();
        else if (fused) r = run_fused();
        else r = run_wavefront();
        if (r != RL_OK) return r;
        return finish();
    }

private:
    // ---- the call
    rl_context* const ctx;
    const rl_path_params* const params;
    const uint64_t* const block_seeds;
    const size_t n_blocks;
    float* const out_rgb;
    const int out_is_device;
    const hipStream_t st;
    rl_render_stats* const stats;
    const Knobs knobs;                 // the context's options as they stood when the render started
    const uint32_t W, H;
    const size_t nby;
    const uint32_t shard_count;
    std::chrono::steady_clock::time_point t_start;
    // ---- work decomposition and form
    std::vector<unsigned> owned, item_base;
    unsigned n_pixels = 0;
    bool per_sample = false, fused = false, fast_math = false, two_pass = false, medium = false;
    bool overlap_wanted = false;       // two passes: the evaluation pass runs beside the chain pass (every chunk)
    bool one_lane_per_pixel = false;   // the per-sample parking buffer could not be allocated: one lane per pixel, no overlap
    bool overlapped = false;           // ... and did
    int cus = 256;
    size_t state_budget = (size_t)24 << 30;
    std::vector<Chunk> chunks;
    unsigned max_chunk_pix = 0;
    Plan plan{1u, 0u, 0u, 0u}, plan_chain{1u, 0u, 0u, 0u};
    unsigned split = 1, n_items = 0, item_shift = 0, P = 0, n_item_pixels = 0;
    Pool pool{};
    float* d_out = nullptr;
    size_t n_partial_rows = 0;
    RenderConst rc{};
    StackConf stc{};
    size_t lds_trav = 0, lds_fused = 0;
    bool timing = false;
    // ---- results
    double ms[4] = {0, 0, 0, 0}, ms_fused = 0.0, ms_chain = 0.0, ms_eval_span = 0.0;
    uint64_t iterations = 0, launches = 2, n_extend = 0;
    unsigned long long spec_stat[3] = {0, 0, 0};
    unsigned spec_group = 0;
    // ---- the chain pass of the two-pass form (plan_chain_pass)
    bool spec = false;
    SpecConf spc{};
    unsigned spec_threads = 0;
    StackConf stc_c{}, stc_s{};        // the stacks of k_stream_chain / of k_stream_spec (its own number of LDS levels on scenes that stream their BVH)
    size_t lds_chain = 0, lds_spec = 0;
    int spec_levels = -1;

    static constexpr unsigned kMaxEvalLaunches = 256u;
    static constexpr size_t kEventsPerIter = 8;

    // this shard's blocks, in creation order
    void decompose() {
        for (size_t b = 0; b < n_blocks; b++) {
            if (b % shard_count != params->shard_index) continue;
            const unsigned bx = (unsigned)(b / nby) * 16u, by = (unsigned)(b % nby) * 16u;
            const unsigned bw = std::min(16u, W - bx), bh = std::min(16u, H - by);
            owned.push_back((unsigned)b);
            item_base.push_back(n_pixels);
            n_pixels += bw * bh;
        }
    }

    int choose_form() {
        per_sample = params->stream_mode == RL_STREAM_PER_SAMPLE;
        // pipeline: 1 = wavefront stage kernels, 2 = persistent fused kernel, 0 = auto = fused unless a pool size is forced (reference-order
        // streams at 1080p x 128 spp: wavefront 8.7 s, fused 2.9 s, fused with the items spread over the waves 1.7 s, fused in two passes: see
        // DESIGN.md; per-sample at 1080p x 32 spp, fused vs wavefront: 508 k-triangle / 6-BSDF scene 127 vs 202 ms, 4.9 k triangles 61 vs 153 ms,
        // Cornell box with mixed BSDFs 23 vs 70 ms, diffuse Cornell box 15 vs 35 ms)
        if (params->pipeline > 2) { rl_set_error("pipeline must be 0 (auto), 1 (wavefront) or 2 (fused)"); return RL_ERR_INVALID_ARGUMENT; }
        fused = params->pipeline == 2 || (params->pipeline == 0 && params->pool_slots == 0);
        fast_math = params->numerics == RL_NUMERICS_FAST;
        if (fast_math && !fused) { rl_set_error("numerics = fast exists for the persistent kernel only (pipeline 0 or 2, pool_slots 0)"); return RL_ERR_UNSUPPORTED; }
        medium = ctx->ds.medium.enabled != 0;
        // Reference-order streams through the persistent kernel run in TWO passes (chain.hip.h): k_stream_chain walks every block's stream with the
        // radiance half of the integrator left out and records the sampler state at the start of each camera sample, then the per-sample form of
        // k_path_fused evaluates all samples from those states with every lane busy.  Same image, same counters as the single-pass walk
        // (option ref_single_pass keeps that form: a test / measurement knob).
        two_pass = !per_sample && fused && !owned.empty() && !knobs.has(K_REF_SINGLE_PASS);
        // the recorded states of ONE cursor position of every owned block must fit the budget (spp beyond ~90 000 at 1080p do not): else the single-pass walk
        if (knobs.has(K_STATE_BUDGET_MB)) state_budget = std::max<size_t>(1, (size_t)knobs.i(K_STATE_BUDGET_MB, 0)) << 20;   // test knob: forces several chunks
        if (two_pass && (size_t)owned.size() * params->spp * 32 > state_budget) two_pass = false;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        return RL_OK;
    }

    // ---- how a set of work items is laid over the lanes.  per_pixel: pixel items (RL_STREAM_PER_SAMPLE, or the second pass of reference-order
    // streams), else one item per owned block.
    Plan plan_items(bool per_pixel, unsigned n_pix, unsigned n_chains) const {
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
            if (overlap_wanted && !params->sample_split) want = std::max(want, (unsigned)std::max<long long>(1, knobs.i(K_EVAL_SPLIT, 4)));
            pl.split = std::max(1u, std::min(want, params->spp));
            if ((size_t)n_pix * params->spp * 3 * sizeof(float) > kSampleBufBudget || one_lane_per_pixel) pl.split = 1;
            while (pl.split > 1 && (size_t)n_pix * pl.split > (size_t)0x7fffff00u) pl.split--;
        }
        pl.n_items = per_pixel ? n_pix * pl.split : n_chains;
        // a pool never needs more slots than there are work items (and `pool_slots` is caller input: keep the rounding below from wrapping)
        unsigned Pp = params->pool_slots ? std::min(params->pool_slots, std::max(pl.n_items, 1u)) : std::min<unsigned>(pl.n_items, 16u << 20);
        Pp = std::max(256u, (unsigned)(((unsigned long long)Pp + 255ull) / 256ull * 256ull));
        if (fused) {
            // One lane per work item by default.  With a participating medium path lengths vary by orders of magnitude, and on scenes
            // that stream their BVH from L2 / HBM the cost per pixel varies as much, so there the grid is only what the chip keeps
            // resident (RL_FUSED_WAVES x 256-lane workgroups per CU) and lanes draw further items from the dispenser as they finish —
            // no workgroup idles behind its slowest pixel (cbox + medium, 32 spp: 185.5 -> 151.6 ms; 508 k triangles: 148.6 -> 126.9 ms;
            // LDS-staged scenes: plain cbox 63.5 vs 63.6 ms, mixed-BSDF cbox 23.3 vs 25.5 ms, so they keep the static tile order).
            const unsigned resident = (unsigned)cus * (unsigned)(ctx->lds_scene ? RL_FUSED_WAVES : RL_FUSED_WAVES_STREAMING) * 256u;
            const bool dynamic_items = knobs.has(K_FUSED_DYNAMIC) ? knobs.i(K_FUSED_DYNAMIC, 0) != 0 : (medium || !ctx->lds_scene);
            Pp = std::max(256u, (std::min(pl.n_items, dynamic_items ? resident : pl.n_items) + 255u) / 256u * 256u);
            // sparse item sets (reference-order streams): one item per 2^item_shift lanes, all of them resident from the start
            // (block chains of a scene with a participating medium: their vertices are scattering events, nearly all shading — ~600 wave instructions of exact f64
            // recipes and sampler steps each, whatever the lane count — and four waves of two chains per SIMD saturate its issue slots: four chains per wave on two waves
            // per SIMD; cbox + medium 1080p x 128 spp, one chain per 4 / 8 / 16 / 32 / 64 lanes: 4405 / 5404 / 2366 / 2474 / 4001 ms, profiles/NEGATIVES.md round 6)
            const unsigned resident4 = (unsigned)cus * 4u * 256u / ((medium && !per_pixel) ? 2u : 1u);
            while (pl.item_shift < 6u && ((size_t)pl.n_items << (pl.item_shift + 1u)) <= resident4) pl.item_shift++;
            if (!per_pixel && knobs.has(K_ITEM_SHIFT)) pl.item_shift = std::min(6u, (unsigned)std::max<long long>(0, knobs.i(K_ITEM_SHIFT, 0)));
            if (pl.item_shift) Pp = std::max(256u, (unsigned)((((size_t)pl.n_items << pl.item_shift) + 255u) / 256u * 256u));
        }
        pl.P = Pp;
        return pl;
    }

    // ---- chunks of the two-pass form: block cursors [c0, c1) of every owned block per chunk, sized so that the recorded sampler states
    // (32 B per camera sample) fit their budget; one chunk unless the render is very large (1080p x 128 spp = 8.5 GB, x 1024 spp = 68 GB: three chunks).
    // The state buffer is sized by what the device has free, not only by the fixed budget (several contexts or shards on one device, a smaller GPU):
    // the budget is cut to the buffer the context already holds + 60 % of the free memory, and if the allocation still fails it is halved until one
    // cursor position of every block no longer fits — then the single-pass walk, which needs no such buffer, renders the frame.
    void plan_chunks() {
        if (two_pass && !knobs.has(K_STATE_BUDGET_MB)) {
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
        max_chunk_pix = 0;
        for (const Chunk& ch : chunks) max_chunk_pix = std::max(max_chunk_pix, ch.n_pix);
    }

    // the plans of every pass and the buffers they need
    int plan_lanes() {
        // The evaluation pass runs beside the chain pass (chunk by chunk) in the exact build when the context has its second stream — unless a group of the speculative
        // pass spans a workgroup (option spec_group = 256: the wave that flags a block is then not the only one that wrote its states).  Decided BEFORE the lanes are
        // planned: it asks for several lanes per pixel, i.e. for the per-sample parking buffer.
        overlap_wanted = two_pass && ctx->stream2 && !fast_math && !knobs.has(K_NO_OVERLAP) && knobs.i(K_SPEC_GROUP, 0) != 256;
        int rcode;
        for (int attempt = 0; attempt < 2; attempt++) {
            plan = two_pass ? plan_items(true, max_chunk_pix, 0) : plan_items(per_sample, n_pixels, (unsigned)owned.size());   // two-pass: the largest second pass
            // (a SMALLER chunk can ask for MORE lanes per pixel, hence more slots and statistics rows than the largest one: uneven chunks at 1080p — 131 + 125 cursors — lost
            // 1 % of the counters and wrote past the rows; everything sized from `plan` below covers every chunk's own plan)
            if (two_pass) for (const Chunk& ch : chunks) { const Plan pc = plan_items(true, ch.n_pix, 0); plan.P = std::max(plan.P, pc.P); plan.split = std::max(plan.split, pc.split); }
            plan_chain = two_pass ? plan_items(false, 0, (unsigned)owned.size()) : Plan{1u, 0u, 0u, 0u};
            split = plan.split; n_items = plan.n_items; item_shift = plan.item_shift;
            P = std::max(plan.P, plan_chain.P);
            n_item_pixels = two_pass ? max_chunk_pix : n_pixels;
            if (split <= 1) break;
            // several lanes per pixel park their samples in HBM ([spp][pixel][3] floats: 3.2 GB at 1080p x 128 spp).  Where that buffer cannot be had (a full device, several
            // contexts in flight) the frame still renders: one lane per pixel, the evaluation pass after the chain pass — unless the caller asked for the lanes himself
            if (ensure(&ctx->d_sample_buf, &ctx->sample_buf_capacity, (size_t)n_item_pixels * params->spp * 3) == RL_OK) break;
            if (params->sample_split || attempt == 1) return RL_ERR_HIP;
            one_lane_per_pixel = true; overlap_wanted = false;
        }
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
        pool = ctx->pool;
        pool.P = P;
        d_out = out_rgb;
        if (!out_is_device) {
            if ((rcode = ensure(&ctx->d_out, &ctx->out_capacity, (size_t)3 * W * H)) != RL_OK) return rcode;
            d_out = ctx->d_out;
        }
        n_partial_rows = std::max<size_t>((P + 255) / 256, (size_t)cus * 8u * rl_context::kEvalStreams);      // (the queue-fed evaluation launches use a grid of the resident workgroups)
        if ((rcode = ensure(&ctx->d_partials, &ctx->partials_capacity, n_partial_rows * STAT_COUNT)) != RL_OK) return rcode;
        return RL_OK;
    }

    int upload() {
        HIP_OK(hipMemcpyAsync(ctx->d_owned, owned.data(), owned.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
        if (!two_pass) HIP_OK(hipMemcpyAsync(ctx->d_item_base, item_base.data(), item_base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(ctx->d_block_seeds, block_seeds, n_blocks * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(d_out, 0, (size_t)3 * W * H * sizeof(float), st));
        Counters init{};
        init.active = std::min(plan.P, n_items);
        init.next_item = item_shift ? n_items : plan.P;
        if (!two_pass) HIP_OK(hipMemcpyAsync(ctx->d_counters, &init, sizeof(init), hipMemcpyHostToDevice, st));
        HIP_OK(hipMemsetAsync(ctx->d_partials, 0, n_partial_rows * STAT_COUNT * sizeof(unsigned long long), st));

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

        lds_trav = traversal_lds_bytes(ctx, ctx->lds_scene, 256, true);
        int rcode;
        if ((rcode = stack_conf(ctx, (size_t)((P + 255) / 256) * 256, &stc)) != RL_OK) return rcode;
        // (LDS-staged scenes: k_path_fused stages the nodes as two-level records, 144 instead of 68 bytes each)
        const size_t lds_two_level_extra = (ctx->lds_scene && RL_LDS_TWO_LEVEL) ? (size_t)16 * (lds_scene2_float4s(ctx->ds.n_nodes, ctx->ds.n_prims) - lds_scene_float4s(ctx->ds.n_nodes, ctx->ds.n_prims)) : 0;
        lds_fused = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + lds_two_level_extra + ((!ctx->lds_scene && RL_FUSED_COLD_SCRATCH) ? 0 : kFusedColdBytes) + ((ctx->lds_scene || RL_COOP_FETCH != 1) ? 0 : (size_t)4 * kCoopStageFloat4s * sizeof(float4));

        const dim3 block(256);
        if (per_sample && !owned.empty()) hipLaunchKernelGGL(k_seed_pixels, dim3(((unsigned)owned.size() + 63) / 64), dim3(64), 0, st, rc);
        if (!fused) hipLaunchKernelGGL(k_init, dim3((plan.P + 255) / 256), block, 0, st, rc, pool);

        // events: 4 timed kernel classes per iteration of the wavefront pipeline, 4 per chunk of the two-pass form
        timing = stats != nullptr && !knobs.has(K_NO_EVENTS);
        const unsigned poll_every = per_sample ? 8u : 32u;
        if (timing) while (ctx->events.size() < kEventsPerIter * poll_every) { hipEvent_t ev; HIP_OK(hipEventCreate(&ev)); ctx->events.push_back(ev); }
        return RL_OK;
    }

    void launch_fused(const RenderConst& rcl, dim3 grid, hipStream_t on, const StackConf* stcl = nullptr) const {
        const int mat = ctx->single_bsdf ? ctx->bsdf_type : -1;
        const StackConf& s = stcl ? *stcl : stc;
        if (rcl.queue_mode != 0u)       // (exact build only: the overlap is off in the tolerance build)
            (ctx->lds_scene ? launch_fusedq_lds : launch_fusedq_stream)(mat, medium, ctx->area_lights_only, grid, dim3(256), lds_fused, on, rcl, ctx->ds, s);
        else
            (ctx->lds_scene ? (fast_math ? launch_fused_lds_fast : launch_fused_lds) : (fast_math ? launch_fused_stream_fast : launch_fused_stream))(mat, medium, ctx->area_lights_only, grid, dim3(256), lds_fused, on, rcl, ctx->ds, s);
    }

    // an error must not leave kernels running on the context's streams behind the caller's back (they write ctx buffers and the caller's framebuffer)
    void drain() const {
        (void)hipStreamSynchronize(st);
        for (int k = 0; k < rl_context::kEvalStreams; k++) if (ctx->eval_streams[k]) (void)hipStreamSynchronize(ctx->eval_streams[k]);
        (void)hipGetLastError();
    }
    struct DrainOnError {       // armed while launches of a chunk may be in flight: every non-OK exit of the enclosing scope drains first
        const PathRender* r; bool armed = true;
        ~DrainOnError() { if (armed) r->drain(); }
    };

    // ---- which kernel walks the chains and in what shape: k_stream_spec (spec.hip.h) with every lane busy — exact build; option chain_serial keeps the
    // one-lane-per-block walk of k_stream_chain (the cross-check).  A pure function of the scene, the parameters and the context's options: the same
    // frame takes the same kernels whether it is the context's first or its fifth (the choice used to follow the draws per sample
    // the context's previous render had measured).
    int plan_chain_pass() {
        // tiny LDS-staged scenes (the Cornell box: 19 nodes + 36 triangles): the lanes of a chain's group precompute its ray's node / triangle records
        // (trace.hip.h: precompute_records) — when a chain has at least 32 lanes to itself and the records fit a few passes
        stc_c = stc;
        // (not with a medium: most of its vertices are scattering events whose rays end in the volume, and the pass then costs more than it saves — cbox + medium,
        // 1080p x 16 spp: 404 vs 353 ms; a variant with ONE chain per wave, the records left in registers and a wave-uniform v_readlane walk measured no better
        // than the LDS records at the same chains per wave: 767.5 vs 768.1 ms, and 2 chains per wave beat both: 705 ms)
        if (ctx->lds_scene && !medium && ctx->ds.n_nodes <= 64u && ctx->ds.n_prims <= 64u && plan_chain.item_shift >= 5u && !knobs.has(K_CHAIN_NO_PRE)) stc_c.pre_group = 1 << plan_chain.item_shift;
        // scenes that stream their BVH (exact build): the group fetches 16-node treelet blocks for its chain (traverse_treelet); 72 float4 of LDS per chain
        const bool treelets = !ctx->lds_scene && !fast_math && ctx->ds.nodes_t && ctx->ds.root_t >= 0 && plan_chain.item_shift >= 5u && ctx->ds.stack_depth <= 512u && !knobs.has(K_CHAIN_NO_TREELETS);
        if (treelets) stc_c.pre_group = 1 << plan_chain.item_shift;
        // (both LDS-hungry forms are only taken when their workgroup fits: a very deep BVH or a large record set falls back to the plain per-node walk — ADVICE r3)
        auto chain_lds_of = [&](bool tl) {
            return traversal_lds_bytes(ctx, ctx->lds_scene, 256, false) + (tl ? (size_t)(256 / stc_c.pre_group) * (72 + (ctx->ds.stack_depth + 1) / 2) * 16
                   : (stc_c.pre_group ? (size_t)(256 / stc_c.pre_group) * ((size_t)ctx->ds.n_nodes * 32 + (size_t)ctx->ds.n_prims * 8) : 0));
        };
        if (chain_lds_of(treelets) > ctx->lds_limit) stc_c.pre_group = 0;
        lds_chain = chain_lds_of(treelets && stc_c.pre_group != 0);

        spec = !fast_math && !knobs.has(K_CHAIN_SERIAL);
        // Two walks of a pixel fall in with each other after about as many samples as a sample takes draws, so the speculative pass only pays when a
        // pixel has several times that many samples (spec.hip.h): decided from what the scene is (a participating medium: ~150 draws per sample on the
        // Cornell box — 1080p x 128 spp: 3085 vs 2837 ms; without one ~10-12) unless the host says what its scene takes (option spec_draws_per_sample:
        // rl_render_stats.rng_draws / camera_samples of an earlier render is the measured figure).  Option spec_force (tests): always, also inside the kernel.
        const bool spec_force = knobs.has(K_SPEC_FORCE);
        if (spec && !spec_force) {
            const double nbar = knobs.f(K_SPEC_DRAWS_PER_SAMPLE, medium ? 150.0 : 12.0);
            if ((double)params->spp < 4.0 * nbar) spec = false;
        }
        // its workgroup parks 30 words of state per lane + the groups' scratch in LDS on top of the scene and the stacks: where that exceeds what a workgroup may ask for on
        // this device the launch would be refused — the serial chain (k_stream_chain), whose workgroup is the plain traversal one, renders those scenes instead
        // (scenes that stream their BVH: the kernel's occupancy is set by its LDS — 30 KB of parked state per workgroup + the stacks — so the stack levels kept there are its own
        // choice: three, which lets a fourth workgroup fit a CU (spec.hip.h: RL_SPEC_WAVES_STREAMING); option spec_lds_levels overrides)
        spec_levels = ctx->lds_scene ? -1 : (int)std::max<long long>(0, std::min<long long>(knobs.i(K_SPEC_LDS_LEVELS, kSpecLdsLevelsStreaming), lds_levels_of(ctx)));
        lds_spec = traversal_lds_bytes(ctx, ctx->lds_scene, 256, false, spec_levels) + (size_t)(kSpecColdWords - (ctx->lds_scene ? 0 : kSpecHelperWords)) * 256 * 4 + kSpecGroupLdsBytes;
        const size_t spec_lds_limit = knobs.has(K_SPEC_LDS_LIMIT_TEST) ? (size_t)knobs.i(K_SPEC_LDS_LIMIT_TEST, 0) : ctx->lds_limit;      // (test knob: a device with a smaller limit)
        if (lds_spec > spec_lds_limit) spec = false;
        spc = SpecConf{};
        spec_threads = 0;
        if (spec) {
            // lanes per block: the fewest that still give every SIMD about two waves (a wider batch looks further ahead, so its windows are wider)
            unsigned group = 16u;
            while (group < 64u && (size_t)owned.size() * group < (size_t)cus * 4u * 2u * 64u) group <<= 1;
            // scenes that stream their BVH are bound by the latency of a wave's dependent fetches: one block per wave, four lanes per pixel
            // (508 k triangles, 1080p x 128 spp: 32 x 1 / 64 x 4 = 4377 / 2667 ms; the serial chain 5689 ms); LDS-staged ones: two lanes per pixel from 32 lanes per block on
            if (!ctx->lds_scene) group = 64u;
            if (knobs.has(K_SPEC_GROUP)) { const long long g = knobs.i(K_SPEC_GROUP, 0); if (g == 16 || g == 32 || g == 64 || g == 256) group = (unsigned)g; }
            spc.group = group;
            // sixteen pixels per batch whatever the group: a longer look-ahead widens every window and misses more often (shard 0 of 8 at 1024 spp, 64 lanes per block,
            // 1 / 2 / 4 lanes per pixel: 2621 / 1235 / 731 ms; full frame at 128 spp, 32 lanes: 2 lanes per pixel 265, 4: 705)
            spc.sub = std::max(1u, group / 16u);
            spc.serial_ratio = spec_force ? 0.0f : (float)knobs.f(K_SPEC_SERIAL_RATIO, 3.0);
            if (knobs.has(K_SPEC_SUB)) { const long long v = knobs.i(K_SPEC_SUB, 0); if ((v == 1 || v == 2 || v == 4 || v == 8 || v == 16) && (unsigned)v <= group) spc.sub = (unsigned)v; }
            spc.cap = std::max(96u, std::min(3u * params->spp + 64u, 1u << 20));
            if (knobs.has(K_SPEC_CAP)) spc.cap = std::max(4u, (unsigned)knobs.i(K_SPEC_CAP, 0));
            spc.probe = knobs.has(K_SPEC_PROBE) ? (unsigned)knobs.i(K_SPEC_PROBE, 0) : std::min(32u, std::max(4u, params->spp));
            spc.lead = (unsigned)knobs.i(K_SPEC_LEAD, 24);
            spc.lead_max = (unsigned)knobs.i(K_SPEC_LEAD_MAX, 128);
            spc.lead_var = (float)knobs.f(K_SPEC_LEAD_VAR, 100.0);      // (cbox 1080p x 128 spp: 265.5 -> 260.2 ms; probing every batch: 319.6 ms)
            spc.extra = (unsigned)knobs.i(K_SPEC_EXTRA, 0);            // (cbox 1080p x 128 spp: 2.84 M instead of 3.57 M serial samples, 383 M instead of 306 M walked: 261 vs 259 ms — a wash, off)
            // (cbox 1080p x 128 spp, three workgroups per CU: 244 -> 198 ms; 8 / 16 / 32 lanes alike.  Scenes that stream their BVH: 2452 -> 2517 ms on the 508 k-triangle scene — a helper's sample is a
            // chain of dependent fetches like any other there — so off)
            spc.dense = !ctx->lds_scene ? 0u : (knobs.has(K_SPEC_DENSE) ? std::min(64u, (unsigned)knobs.i(K_SPEC_DENSE, 0)) : 16u);
            spc.dense_frac = (float)knobs.f(K_SPEC_DENSE_FRAC, 0.6);
            spc.probe_every = (unsigned)knobs.i(K_SPEC_PROBE_EVERY, 0);
            // window margins in standard deviations of the predicted offsets: with one block per wave a pixel the chain has to be walked through stalls the whole wave, so wider
            // (shard 0 of 8, 1024 spp: 1.65 / 2.5 sigma = 714 / 688 ms; full frame, two blocks per wave: 281 / 292)
            spc.ks = (float)knobs.f(K_SPEC_KS, group >= 64u ? 2.5 : 1.65);
            spc.ke = (float)knobs.f(K_SPEC_KE, group >= 64u ? 2.5 : 1.65);
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
        int rcode;
        if (spec) {
            // the masks depend on the camera, the scene bounds and the shard only: computed once per context and shard
            const bool expand = !params->has_max_depth || 1u < params->max_depth;
            const bool no_trivial = knobs.has(K_SPEC_NO_TRIVIAL);
            const uint64_t key = ((uint64_t)params->shard_index << 33) | ((uint64_t)shard_count << 1) | (expand ? 1u : 0u);
            if (key != ctx->trivial_key || ctx->trivial_capacity < owned.size() * 8 || no_trivial) {
                std::vector<unsigned> masks;
                trivial_pixel_masks(trivial_input(ctx), params, owned, nby, no_trivial, &masks);
                if ((rcode = ensure(&ctx->d_trivial, &ctx->trivial_capacity, masks.size())) != RL_OK) return rcode;
                HIP_OK(hipMemcpyAsync(ctx->d_trivial, masks.data(), masks.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
                HIP_OK(hipStreamSynchronize(st));        // (`masks` is a local)
                ctx->trivial_key = no_trivial ? ~0ull : key;
            }
            if ((rcode = ensure(&ctx->d_spec_stats, &ctx->spec_stats_capacity, 32 + 8 * (size_t)(spec_threads / 64u))) != RL_OK) return rcode;
            HIP_OK(hipMemsetAsync(ctx->d_spec_stats, 0, 32 * sizeof(unsigned long long), st));
            spc.trk_off = ctx->d_trk_off; spc.trk_st = ctx->d_trk_st; spc.trivial = ctx->d_trivial;
            spc.stats = (stats || knobs.has(K_SPEC_STATS)) ? ctx->d_spec_stats : nullptr;
            // (one overflow buffer serves both passes: the stride is the larger launch; the pass with fewer LDS levels — more overflow levels — sizes it first)
            const size_t n_thr = std::max<size_t>((size_t)((P + 255) / 256) * 256, spec_threads);
            if ((rcode = stack_conf(ctx, n_thr, &stc_s, false, spec_levels)) != RL_OK) return rcode;
            if ((rcode = stack_conf(ctx, n_thr, &stc)) != RL_OK) return rcode;
            stc_c.overflow = stc.overflow; stc_c.overflow_stride = stc.overflow_stride;
        }
        return RL_OK;
    }

    void launch_chain_pass(const RenderConst& ra) const {
        const int mat = ctx->single_bsdf ? ctx->bsdf_type : -1;
        if (spec) (ctx->lds_scene ? launch_spec_lds : launch_spec_stream)(mat, medium, dim3(spec_threads / 256u), dim3(256), lds_spec, st, ra, ctx->ds, stc_s, spc);
        else (ctx->lds_scene ? (fast_math ? launch_chain_lds_fast : launch_chain_lds) : (fast_math ? launch_chain_stream_fast : launch_chain_stream))(mat, medium, dim3((plan_chain.P + 255) / 256), dim3(256), lds_chain, st, ra, ctx->ds, stc_c);
    }

    // the completion flags (mapped host memory), the block lists and the claim counters of the evaluation launches beside the chain pass
    int prepare_queue(RenderConst& ra, unsigned chain_grid) {
        // device: [0] started workgroups, [16 + k] the claim counter of the k-th evaluation launch, [16 + kMaxEvalLaunches + i] the block lists (completion order);
        // mapped host memory: [0] "every chain workgroup runs", [16 + j] "block j is complete"; pinned staging of the lists
        const size_t n_dev = 16 + (size_t)kMaxEvalLaunches + owned.size(), n_flags = 16 + owned.size();
        int rcode;
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
        return RL_OK;
    }

    // Overlapped: k_path_fused<.., QUEUE = true> on the context's low-priority streams WHILE the chain pass runs on `st`.  Nothing on the device waits for anything:
    // the chain kernels flag completed blocks in mapped host memory, THIS host thread (the call is synchronous anyway) collects them and launches the evaluation
    // kernel over explicit lists of complete blocks — not before the chain kernel has reported that every one of its workgroups runs (so the launches beside it
    // only take resources it has no further use for), then whenever a fifth of the blocks still to come have come in (at least 64: a launch lasts as long as its
    // slowest pixel and should fill a good part of the chip), the rest when the chain pass has ended.  Same samples from the same states, folded per pixel in
    // sample order: same bits (option no_overlap keeps the two passes back to back: the cross-check).
    int overlap_loop(const RenderConst& rb, const Plan& pb, unsigned seq, dim3 grid_q, const StackConf& stc_q) {
        HIP_OK(hipEventRecord(ctx->ev_chain_done, st));
        const unsigned n_blocks_owned = (unsigned)owned.size();
        volatile unsigned* hf = ctx->h_flags;
        std::vector<unsigned> waiting(n_blocks_owned);          // owned blocks not listed yet (the scan below only looks at these; it shrinks as blocks complete)
        for (unsigned j = 0; j < n_blocks_owned; j++) waiting[j] = j;
        unsigned n_listed = 0, n_launches = 0, scan_from = 0;
        // the poll interval grows with the number of flags a poll reads (50 us for a 1080p frame's 8160 blocks, ~1 ms per 100 k blocks)
        const auto poll_us = std::chrono::microseconds(50 + n_blocks_owned / 100u);
        bool chain_over = false, started = false;
        const unsigned resident = grid_q.x;
        // (a launch when a 1 / batch_div of the blocks still to come — at least batch_min — have come in: the sweep in NEGATIVES round 5)
        const unsigned batch_min = (unsigned)std::max<long long>(1, knobs.i(K_EVAL_MIN, 64)), batch_div = (unsigned)std::max<long long>(1, knobs.i(K_EVAL_DIV, 5));
        std::chrono::steady_clock::time_point t_first{};
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
            if (n_launches == 0) t_first = std::chrono::steady_clock::now();
            launch_fused(rq, dim3(std::max(1u, wgs)), on, &stc_k);
            HIP_OK(hipGetLastError());
            n_launches++; launches++;
            return RL_OK;
        };
        int rcode;
        while (n_listed < n_blocks_owned) {
            if (!chain_over) {
                const hipError_t qe = hipEventQuery(ctx->ev_chain_done);
                if (qe == hipSuccess) chain_over = true;
                else if (qe != hipErrorNotReady) { rl_set_error(std::string("hipEventQuery(chain pass): ") + hipGetErrorString(qe)); (void)hipGetLastError(); return RL_ERR_HIP; }
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
                if ((rcode = launch_batch(scan_from, fresh)) != RL_OK) return rcode;
                scan_from = n_listed;
            }
            if (n_listed < n_blocks_owned && !chain_over) std::this_thread::sleep_for(poll_us);
        }
        if (scan_from < n_listed && (rcode = launch_batch(scan_from, n_listed - scan_from)) != RL_OK) return rcode;      // the blocks listed last
        if (knobs.has(K_QUEUE_DEBUG)) std::fprintf(stderr, "[queue] %u evaluation launches beside / after the chain pass, %u blocks, started flag %s\n", n_launches, n_listed, started ? "seen" : "not seen");
        HIP_OK(hipStreamSynchronize(st));                      // the chain pass (over already: every block was flagged or its event had fired)
        for (int k = 0; k < rl_context::kEvalStreams; k++) HIP_OK(hipStreamSynchronize(ctx->eval_streams[k]));      // the last evaluation launches
        // the evaluation pass's own span on the host clock, first launch to last completion (its launches run on several streams beside the chain pass: no pair of events brackets them)
        if (n_launches) ms_eval_span += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count();
        return RL_OK;
    }

    // one chunk of the two-pass form: chain pass, evaluation pass (beside or after it), fold
    int run_chunk(const Chunk& ch) {
        DrainOnError guard{this};
        const dim3 block(256);
        HIP_OK(hipMemcpyAsync(ctx->d_item_base, ch.base.data(), ch.base.size() * sizeof(unsigned), hipMemcpyHostToDevice, st));
        RenderConst ra = rc;
        ra.stream_mode = RL_STREAM_REFERENCE_ORDER;
        ra.n_items = plan_chain.n_items; ra.item_shift = plan_chain.item_shift; ra.split = 1;
        ra.n_state_pixels = ch.n_pix; ra.cursor_begin = ch.c0; ra.cursor_end = ch.c1;
        const unsigned chain_grid = spec ? spec_threads / 256u : (plan_chain.P + 255u) / 256u;
        const bool overlap = overlap_wanted;
        int rcode;
        if (overlap && (rcode = prepare_queue(ra, chain_grid)) != RL_OK) return rcode;
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
        launch_chain_pass(ra);
        if (timing) hipEventRecord(ctx->events[1], st);
        HIP_OK(hipGetLastError());
        if (overlap) {
            rb.queue = ctx->d_queue; rb.queue_mode = 1u; rb.chain_grid = chain_grid;
            if ((rcode = overlap_loop(rb, pb, ra.queue_seq, grid_q, stc_q)) != RL_OK) return rcode;
            overlapped = true;
            if (timing) hipEventRecord(ctx->events[3], st);
        } else {
            if (timing) hipEventRecord(ctx->events[2], st);
            launch_fused(rb, dim3((pb.P + 255) / 256), st);
            if (timing) hipEventRecord(ctx->events[3], st);
        }
        if (pb.split > 1) hipLaunchKernelGGL(k_fold_samples, dim3((ch.n_pix + 255) / 256), block, 0, st, rb);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(st));       // (the chunk's host arrays and the counters block are reused by the next chunk)
        guard.armed = false;
        if (timing) {
            float t = 0.0f;
            HIP_OK(hipEventElapsedTime(&t, ctx->events[0], ctx->events[1])); ms_chain += t;
            // overlapped: what the evaluation pass still takes AFTER the chain pass has ended (the part of it that is not hidden); its whole span: ms_eval_span
            HIP_OK(hipEventElapsedTime(&t, ctx->events[overlap ? 1 : 2], ctx->events[3])); ms_fused += t;
            if (!overlap) ms_eval_span += t;
        }
        launches += 3 + (pb.split > 1 ? 1 : 0);
        return RL_OK;
    }

    int run_two_pass() {
        int rcode;
        if ((rcode = plan_chain_pass()) != RL_OK) return rcode;
        for (const Chunk& ch : chunks) if ((rcode = run_chunk(ch)) != RL_OK) return rcode;
        dump_stage_timers(ctx->lds_scene);
        if (ctx->lds_scene) dump_chain_timers_lds(); else dump_chain_timers_stream();     // dev-only build
        if (spec && spc.stats && (rcode = read_spec_stats()) != RL_OK) return rcode;
        iterations = chunks.size();
        return RL_OK;
    }

    // k_stream_spec's counters (rl_render_stats.reserved) and, in the dev builds, its cycle shares and wave lifetimes
    int read_spec_stats() {
        unsigned long long spec_totals[32] = {0};
        HIP_OK(hipMemcpy(spec_totals, ctx->d_spec_stats, sizeof(spec_totals), hipMemcpyDeviceToHost));
        spec_group = spc.group; spec_stat[0] = spec_totals[0]; spec_stat[1] = spec_totals[1]; spec_stat[2] = spec_totals[2];
        const bool verbose = knobs.has(K_SPEC_STATS);
        if (verbose) std::fprintf(stderr, "[spec] LDS per workgroup %zu bytes (scene %zu, %d stack levels)\n", lds_spec, (size_t)(ctx->lds_scene ? ctx->scene_lds_bytes : 0), stc_s.lds_levels);
        if (verbose) std::fprintf(stderr, "[spec] group %u x sub %u cap %u: %llu speculative + %llu serial + %llu probe samples for %llu camera samples (%.2f x, %.2f serial per pixel), %llu wave iterations\n",
            spc.group, spc.sub, spc.cap, spec_totals[0], spec_totals[1], spec_totals[2], (unsigned long long)n_pixels * params->spp,
            (double)(spec_totals[0] + spec_totals[1] + spec_totals[2]) / std::max(1.0, (double)n_pixels * params->spp), (double)spec_totals[1] / std::max(1u, n_pixels), spec_totals[3]);
        if (knobs.has(K_SPEC_WAVE_TIMES) && spec_totals[8]) {     // dev build: lifetime of every wave (100 MHz clock)
            std::vector<unsigned long long> wt(8 * (size_t)(spec_threads / 64u));
            HIP_OK(hipMemcpy(wt.data(), ctx->d_spec_stats + 32, wt.size() * 8, hipMemcpyDeviceToHost));
            FILE* f = std::fopen(knobs.str(K_SPEC_WAVE_TIMES), "w");
            if (f) { unsigned long long t0 = ~0ull; for (size_t w = 0; w < wt.size() / 8; w++) if (wt[8 * w]) t0 = std::min(t0, wt[8 * w]);
                     for (size_t w = 0; w < wt.size() / 8; w++) std::fprintf(f, "%zu %.3f %.3f %llu %llu %.3f %.3f %llu %llu\n", w, (wt[8 * w] - t0) * 1e-5, (wt[8 * w + 1] - t0) * 1e-5, wt[8 * w + 2], wt[8 * w + 3], wt[8 * w + 4] * 1e-5, wt[8 * w + 5] * 1e-5, wt[8 * w + 6], wt[8 * w + 7]); std::fclose(f); }
        }
        if (verbose && spec_totals[8]) {     // dev build (-DRL_SPEC_TIMERS)
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
        return RL_OK;
    }

    // one launch of the persistent kernel
    int run_fused() {
        if (timing) hipEventRecord(ctx->events[0], st);
        launch_fused(rc, dim3((plan.P + 255) / 256), st);
        if (timing) hipEventRecord(ctx->events[1], st);
        HIP_OK(hipGetLastError());          // a refused launch configuration is not sticky: without this the sync below would "succeed"
        HIP_OK(hipStreamSynchronize(st));
        if (timing) { float t = 0.0f; HIP_OK(hipEventElapsedTime(&t, ctx->events[0], ctx->events[1])); ms_fused = t; ms_eval_span = t; }
        dump_stage_timers(ctx->lds_scene);     // dev-only build (-DRL_STAGE_TIMERS): per-stage cycle shares of the fused loop
        launches += 1;
        iterations = 1;
        return RL_OK;
    }

    // the wavefront stage kernels: raygen -> extend -> shade -> shadow per iteration until no slot is active
    int run_wavefront() {
        const dim3 block(256);
        const dim3 grid_all((plan.P + 255) / 256);
        // material-sort kernel: sparse pools (several lanes per pixel) are gathered four 256-slot chunks per workgroup
        const bool use_sort = !ctx->single_bsdf;
        const unsigned sort_chunks = split > 1 ? 4u : 1u;
        const dim3 grid_sort((P + 1023) / 1024);
        const dim3 grid_persistent(std::min<unsigned>((P + 255) / 256, (unsigned)cus * 8u));
        const unsigned poll_every = per_sample ? 8u : 32u;
        const DeviceScene& ds = ctx->ds;
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
        for (;;) {
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
        {
            unsigned long long h[8];
            HIP_OK(hipStreamSynchronize(st));
            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trav_stats), sizeof(h));
            std::fprintf(stderr, "[trav] rays %llu (zero-step %.1f %%)  steps/ray %.1f  tris/ray %.1f  waves %llu  lanes/wave %.1f  paid steps/ray %.1f  step utilisation %.1f %%\n",
                         h[0], 100.0 * h[5] / (double)h[0], (double)h[1] / h[0], (double)h[2] / h[0], h[4], (double)h[0] / h[4], (double)h[3] / h[0], 100.0 * h[1] / (double)h[3]);
            std::memset(h, 0, sizeof(h)); hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), h, sizeof(h));
        }
#endif
        return RL_OK;
    }

    int finish() {
        // one more raygen pass is never needed: `active` reaches 0 inside k_raygen after the last fold.
        if (split > 1 && !two_pass) { hipLaunchKernelGGL(k_fold_samples, dim3((n_pixels + 255) / 256), dim3(256), 0, st, rc); launches += 1; }
        if (!out_is_device) HIP_OK(hipMemcpyAsync(out_rgb, d_out, (size_t)3 * W * H * sizeof(float), hipMemcpyDeviceToHost, st));
        std::vector<unsigned long long> partials(n_partial_rows * STAT_COUNT);
        HIP_OK(hipMemcpyAsync(partials.data(), ctx->d_partials, partials.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        unsigned long long totals[STAT_COUNT] = {0};
        for (size_t r = 0; r < n_partial_rows; r++) for (int k = 0; k < STAT_COUNT; k++) totals[k] += partials[r * STAT_COUNT + k];
        HIP_OK(hipGetLastError());
        const auto t_end = std::chrono::steady_clock::now();
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
            stats->ms_other = ms_fused;   // the persistent fused kernel (pipeline 2); overlapped: the part of the evaluation pass left after the chain pass had ended
            stats->ms_prepass = ms_chain;  // k_stream_chain / k_stream_spec (reference-order streams, first pass)
            stats->n_extend_launches = n_extend;
            stats->reserved[0] = spec_stat[0]; stats->reserved[1] = spec_stat[1]; stats->reserved[2] = spec_stat[2]; stats->reserved[3] = spec_group;   // speculative / serial / probe samples of k_stream_spec, its lanes per block (0: the serial chain ran)
            stats->chunks = two_pass ? (uint32_t)chunks.size() : 0u;
            stats->overlapped = overlapped ? 1u : 0u;
            stats->ms_eval_span = ms_eval_span;
        }
        return RL_OK;
    }
};

}  // namespace
