// stages.hip.h — the four per-vertex stage functions shared by the wavefront kernels and the persistent fused kernel:
// raygen_slot, extend_slot, shade_slot<MAT, MEDIUM>, shadow_slot (reference: integrators/mod.rs:403-450, paths/strategies/*.rs,
// integrators/explicit/path.rs:37-184).
// Part of the single translation unit wavefront.hip (included once, after devmath / shading / trace).
#pragma once

namespace rl {

// ------------------------------------------------------------------------------------------
// Camera::generate (src/camera.rs:81-91): direction of the ray through image position (u, v)
RL_DEV V3 camera_direction(const DeviceScene& sc, float u, float v) {
    const float* m = sc.camera.sample_to_camera;
    float sx = div_rn(u, (float)sc.camera.width), sy = div_rn(v, (float)sc.camera.height), sz = 0.0f;
    float hx = ((m[0] * sx + m[4] * sy) + m[8] * sz) + m[12] * 1.0f;
    float hy = ((m[1] * sx + m[5] * sy) + m[9] * sz) + m[13] * 1.0f;
    float hz = ((m[2] * sx + m[6] * sy) + m[10] * sz) + m[14] * 1.0f;
    float hw = ((m[3] * sx + m[7] * sy) + m[11] * sz) + m[15] * 1.0f;
    float inv_w = div_rn(1.0f, hw);
    V3 near_p = mk3(hx * inv_w, hy * inv_w, hz * inv_w);
    V3 dl = normalize(near_p);
    const float* tw = sc.camera.to_world;
    return mk3(((tw[0] * dl.x + tw[4] * dl.y) + tw[8] * dl.z) + tw[12] * 0.0f,
               ((tw[1] * dl.x + tw[5] * dl.y) + tw[9] * dl.z) + tw[13] * 0.0f,
               ((tw[2] * dl.x + tw[6] * dl.y) + tw[10] * dl.z) + tw[14] * 0.0f);
}

// Reference-order streams in two passes: sampler state at the start of camera sample `s` of chunk pixel `p` ([sample][pixel][4] u64)
RL_DEV Rng load_sample_state(const RenderConst& rc, unsigned s, unsigned p) {
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(rc.sample_states + 4 * ((size_t)s * rc.n_state_pixels + p));
    const ulonglong2 a = q[0], b = q[1];
    Rng r; r.s0 = a.x; r.s1 = a.y; r.s2 = b.x; r.s3 = b.y;
    return r;
}
RL_DEV void store_sample_state(const RenderConst& rc, unsigned s, unsigned p, const Rng& r) {
    ulonglong2* q = reinterpret_cast<ulonglong2*>(rc.sample_states + 4 * ((size_t)s * rc.n_state_pixels + p));
    q[0] = make_ulonglong2(r.s0, r.s1); q[1] = make_ulonglong2(r.s2, r.s3);
}

// raygen_chain_slot — first pass of reference-order streams (k_stream_chain): the slot owns one 16x16 block and walks (iy, ix, sample) in
// compute_mc's order (integrators/mod.rs:420-435) on the block's own sampler, like raygen_slot does in RL_STREAM_REFERENCE_ORDER — but all it
// keeps of a camera sample is where in the stream it STARTS (sample_states); no radiance, no fold.  How many draws a sample takes is decided by
// shade_slot<.., DRAWS_ONLY = true> from the same geometry, BSDF samples and Russian-roulette tests as the full evaluation.
template <class PS>
RL_DEV void raygen_chain_slot(const RenderConst& rc, const DeviceScene& sc, PS& ps) {
    const unsigned flags = PU(U_FLAGS);
    const unsigned item = PU(U_ITEM);
    unsigned s = PU(U_SAMPLE), cursor = PU(U_CURSOR);
    unsigned bx, by, bw, bh;
    const unsigned b = rc.owned_blocks[item];
    block_geometry(rc, b, &bx, &by, &bw, &bh);
    const unsigned c_end = min(rc.cursor_end, bw * bh);
    Rng rng;
    if (flags & ST_FRESH) {
        cursor = rc.cursor_begin; s = 0;
        if (cursor >= c_end) { if (rc.queue) queue_push(rc, item); PU(U_FLAGS) = ST_FINISHED; return; }
        if (rc.cursor_begin == 0u) rng = rng_seed(rc.block_seeds[b], rc.seed_variant);      // the block's own sampler (mod.rs:371)
        else { const unsigned long long* q = rc.chain_states + 4 * (size_t)item; rng.s0 = q[0]; rng.s1 = q[1]; rng.s2 = q[2]; rng.s3 = q[3]; }
    } else {
        rng = load_rng(ps, Q_R0);
        s++;
        if (s == rc.spp) { s = 0; cursor++; }
        if (cursor == c_end) {
            unsigned long long* q = rc.chain_states + 4 * (size_t)item;                     // the next chunk resumes here
            q[0] = rng.s0; q[1] = rng.s1; q[2] = rng.s2; q[3] = rng.s3;
            if (rc.queue) queue_push(rc, item);           // every sample state of the block is recorded: the evaluation pass may start on it
            PU(U_FLAGS) = ST_FINISHED;
            return;
        }
    }
    store_sample_state(rc, s, rc.block_item_base[item] + (cursor - rc.cursor_begin), rng);
    const unsigned px = bx + cursor % bw, py = by + cursor / bw;
    const float u = (float)px + rng_next_f32(rng);          // Path::from_sensor: uv = (ix + next(), iy + next())
    const float v = (float)py + rng_next_f32(rng);
    PU(U_SAMPLE) = s;
    PU(U_CURSOR) = cursor;
    const bool expand = (!rc.has_max || 1u < rc.max_depth);
    if (!expand) { store_rng(ps, Q_R0, rng); PU(U_FLAGS) = ST_REGEN; return; }
    store3(ps, F_DX, camera_direction(sc, u, v));
    if (sc.medium.enabled) PF(F_XI) = rng_next_f32(rng);
    store_rng(ps, Q_R0, rng);
    PU(U_DEPTH) = 1u;
    PU(U_FLAGS) = ST_RAY | (PREV_SENSOR << ST_PREV_SHIFT) | ST_PDF_SA;
}

// raygen_slot — sample completion, work-item hand-out, sampler forking, Path::from_sensor (2 draws) and
// Camera::generate for one slot that asked for regeneration.  DYNAMIC: work items come from the global
// dispenser (wavefront pool); otherwise the slot owns exactly one item (persistent fused kernel).
template <bool DYNAMIC, class PS>
RL_DEV void raygen_slot(const RenderConst& rc, const DeviceScene& sc, PS& ps, unsigned& n_samples, unsigned& n_draws) {
    unsigned flags = PU(U_FLAGS);
    if (!(flags & ST_REGEN)) return;
    const bool fresh = (flags & ST_FRESH) != 0u;
    unsigned item = PU(U_ITEM), s = PU(U_SAMPLE), cursor = PU(U_CURSOR);
    Col acc = loadc(ps, F_AR);
    bool need_item = fresh;
    unsigned bx = 0, by = 0, bw = 1, bh = 1;
    if (rc.stream_mode == RL_STREAM_REFERENCE_ORDER && item < rc.n_items) block_geometry(rc, rc.owned_blocks[item], &bx, &by, &bw, &bh);
    const bool per_pixel = rc.stream_mode != RL_STREAM_REFERENCE_ORDER;    // per-pixel work items: RL_STREAM_PER_SAMPLE, or the second pass of reference-order streams
    const bool given = rc.stream_mode == kStreamGivenStates;
    const unsigned split = per_pixel ? rc.split : 1u;
    const unsigned pitem = split > 1u ? item / split : item;         // pixel item of this lane
    if (!fresh && split > 1u) {
        // sample-parallel pixels: this lane owns samples s, s + split, ...; each sample's radiance is parked in
        // sample_buf[s][pixel item] and k_fold_samples adds them up in sample order, as mod.rs:431 does
        const Col L = loadc(ps, F_LR);
        float* dst = rc.sample_buf + 3 * ((size_t)s * (rc.n_items / split) + pitem);
        dst[0] = L.r; dst[1] = L.g; dst[2] = L.b;
        s += split;
        if (s >= rc.spp) need_item = true;
    } else if (!fresh) {
        // im_block.accumulate(.., c, "primal") in sample order (mod.rs:431)
        acc = acc + loadc(ps, F_LR);
        s++;
        if (s == rc.spp) {
            unsigned pix = per_pixel ? rc.item_pixel[item] : (by + cursor / bw) * rc.W + (bx + cursor % bw);
            Col px = scale_unguarded(acc, rc.inv_spp);            // im_block.scale(1 / spp) (mod.rs:436)
            rc.out[3 * (size_t)pix] = px.r; rc.out[3 * (size_t)pix + 1] = px.g; rc.out[3 * (size_t)pix + 2] = px.b;
            acc = czero();
            s = 0;
            if (per_pixel) need_item = true;
            else { cursor++; if (cursor == bw * bh) need_item = true; }
        }
    }
    Rng rng;
    if (need_item) {
        if (!fresh) item = DYNAMIC ? atomicAdd(&rc.counters->next_item, 1u) : rc.n_items;   // fused kernel: one item per thread
        if (item >= rc.n_items) {
            PU(U_FLAGS) = ST_FINISHED;
            if (DYNAMIC) atomicSub(&rc.counters->active, 1u);
            return;
        }
        cursor = 0;
        if (given) {
            const unsigned pi = split > 1u ? item / split : item;
            if (split > 1u) s = item % split;
            rng = load_sample_state(rc, s, pi);                             // the block sampler as compute_mc would hold it here (k_stream_chain)
        } else if (per_pixel) {
            const unsigned pi = split > 1u ? item / split : item;
            Rng item_rng = rng_seed(rc.item_seed[pi], rc.seed_variant);     // pixel sampler = block_sampler.clone_box()
            if (split > 1u) { s = item % split; for (unsigned k = 0; k < s; k++) rng_next_u64(item_rng); }   // forks of the samples before ours
            rng = rng_seed(rng_next_u64(item_rng), rc.seed_variant);        // sample sampler = pixel_sampler.clone_box()
            store_rng(ps, Q_I0, item_rng);
        } else {
            unsigned b = rc.owned_blocks[item];
            block_geometry(rc, b, &bx, &by, &bw, &bh);
            rng = rng_seed(rc.block_seeds[b], rc.seed_variant);            // the block's own sampler (mod.rs:371)
        }
        PU(U_ITEM) = item;
    } else if (given) {
        rng = load_sample_state(rc, s, pitem);
    } else if (per_pixel) {
        Rng item_rng = load_rng(ps, Q_I0);
        for (unsigned k = 1; k < split; k++) rng_next_u64(item_rng);        // the forks taken by the other lanes of this pixel
        rng = rng_seed(rng_next_u64(item_rng), rc.seed_variant);
        store_rng(ps, Q_I0, item_rng);
    } else {
        rng = load_rng(ps, Q_R0);
    }
    unsigned px, py;
    if (per_pixel) { unsigned pix = rc.item_pixel[split > 1u ? item / split : item]; px = pix % rc.W; py = pix / rc.W; }
    else { px = bx + cursor % bw; py = by + cursor / bw; }
    // Path::from_sensor: uv = (ix + next(), iy + next())
    float u = (float)px + rng_next_f32(rng);
    float v = (float)py + rng_next_f32(rng);
    n_draws += 2;
    n_samples++;
    storec(ps, F_AR, acc);
    PU(U_SAMPLE) = s;
    PU(U_CURSOR) = cursor;
    const bool expand = (!rc.has_max || 1u < rc.max_depth);   // TechniquePathTracing::expand at depth 1
    if (!expand) {   // sensor not expanded: the sample is 0 (next raygen pass folds it)
        storec(ps, F_LR, czero());
        store_rng(ps, Q_R0, rng);
        PU(U_FLAGS) = ST_REGEN;
        return;
    }
    const V3 d = camera_direction(sc, u, v);   // Camera::generate (camera.rs:81-91)
    // the sensor edge's state is implied by PREV_SENSOR and never stored: origin = Camera::position(),
    // weight 1, rr_weight 1, PDF::SolidAngle(1), beta = thr = 1 (strategies/directional.rs:27-41)
    store3(ps, F_DX, d);
    if (sc.medium.enabled) { PF(F_XI) = rng_next_f32(rng); n_draws++; }   // Edge::from_ray's medium.sample(ray, next())
    store_rng(ps, Q_R0, rng);
    PU(U_DEPTH) = 1u;
    PU(U_FLAGS) = ST_RAY | (PREV_SENSOR << ST_PREV_SHIFT) | ST_PDF_SA;
}

// ------------------------------------------------------------------------------------------
// extend_slot / shadow_slot — Acceleration::trace and Acceleration::visible for one slot.
template <class PS, class Stack>
RL_DEV void extend_slot(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, PS& ps, int* dbg = nullptr) {
    const unsigned flags = PU(U_FLAGS);
    const bool primary = ((flags >> ST_PREV_SHIFT) & 3u) == PREV_SENSOR;    // camera rays start at Camera::position()
    V3 o = primary ? mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]) : load3(ps, F_OX);
    V3 d = load3(ps, F_DX);
    Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    traverse<false>(recs, Stack::kBvh4 ? sc.root4 : sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                    o, d, kEps, kF32Max, hit, stack);
    PF(F_T) = hit.t; PF(F_U) = hit.u; PF(F_V) = hit.v;
    PU(U_PRIM) = (unsigned)hit.prim;
    if (dbg) { dbg[0] = hit.steps; dbg[1] = hit.tris; }
}
// Acceleration::visible(p0, p1) (accel.rs:316-343)
template <class Stack>
RL_DEV bool shadow_visible(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, V3 p0, V3 p1) {
    V3 d = p1 - p0;
    float len = length(d);
    d = d / len;
    float tfar = len * (1.0f - 0.00001f);
    Hit hit; hit.t = tfar; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float te;
    if (!slab(mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]), p0, inv_d, kEps, tfar, &te))
        return false;   // root box missed => "occluded" (accel.rs:338-340)
    return !traverse<true>(recs, Stack::kBvh4 ? sc.root4 : sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                           p0, d, kEps, tfar, hit, stack);
}
template <class PS, class Stack>
RL_DEV void shadow_slot(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, PS& ps) {
    if (shadow_visible(sc, recs, stack, load3(ps, F_OX), load3(ps, F_SX))) storec(ps, F_LR, loadc(ps, F_LR) + loadc(ps, F_CR));
}

// Wave-cooperative forms (trace.hip.h: traverse_coop): called by EVERY lane of the wave in step; `has_ray` / `has_shadow` say whether this
// lane carries one, the others only help fetching.  Same state reads and writes as the per-lane forms for the lanes that do.
template <class PS, class Stack>
RL_DEV void extend_slot_coop(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, PS& ps, bool has_ray, float4* stage) {
    const unsigned flags = PU(U_FLAGS);
    const bool primary = ((flags >> ST_PREV_SHIFT) & 3u) == PREV_SENSOR;
    V3 o = primary ? mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]) : load3(ps, F_OX);
    V3 d = load3(ps, F_DX);
    Hit hit; hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    traverse_coop<false>(recs, sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                         o, d, kEps, kF32Max, hit, stack, has_ray, stage);
    if (has_ray) {
        PF(F_T) = hit.t; PF(F_U) = hit.u; PF(F_V) = hit.v;
        PU(U_PRIM) = (unsigned)hit.prim;
    }
}
template <class PS, class Stack>
RL_DEV void shadow_slot_coop(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, PS& ps, bool has_shadow, float4* stage) {
    // Acceleration::visible(p0, p1) (accel.rs:316-343), as shadow_visible
    const V3 p0 = load3(ps, F_OX), p1 = load3(ps, F_SX);
    V3 d = p1 - p0;
    const float len = length(d);
    d = d / len;
    const float tfar = len * (1.0f - 0.00001f);
    Hit hit; hit.t = tfar; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    const V3 lo = mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), hi = mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]);
    const V3 inv_d = mk3(div_rn(1.0f, d.x), div_rn(1.0f, d.y), div_rn(1.0f, d.z));
    float te;
    const bool root_hit = has_shadow && slab(lo, hi, p0, inv_d, kEps, tfar, &te);   // root box missed => "occluded" (accel.rs:338-340)
    const bool occluded = traverse_coop<true>(recs, sc.root, lo, hi, p0, d, kEps, tfar, hit, stack, root_hit, stage);
    if (root_hit && !occluded) storec(ps, F_LR, loadc(ps, F_LR) + loadc(ps, F_CR));
}

// shade_slot<MAT, MEDIUM> — one path vertex of one slot.  MAT >= 0: the hit material is known to have that
// BSDF type (per-BSDF code path, uniform over the calling lanes); MAT = -1: generic (run-time switch).
// DRAWS_ONLY (k_stream_chain, the first pass of reference-order streams): only what decides how the path goes on and how many
// random numbers it takes — medium distance, surface point, BSDF / phase sample, Russian roulette — is evaluated, through the very
// same statements as the full form; emission, MIS, light sampling (its four draws are skipped over) and the radiance fields are left out.
template <int MAT, bool MEDIUM, int LIGHTS = LIGHTS_ANY, bool DRAWS_ONLY = false, class PS>
RL_DEV void shade_slot(const RenderConst& rc, const DeviceScene& sc, PS& ps, unsigned flags,
                       unsigned& n_vertices, unsigned& n_draws, unsigned& n_shadow, unsigned& n_ext) {
    n_ext += 1;      // every shaded slot carried exactly one extension ray through k_extend
    const unsigned prev = (flags >> ST_PREV_SHIFT) & 3u;
    const unsigned depth = PU(U_DEPTH);          // generate()'s depth at which the edge's origin vertex was expanded
    const int prim = (int)PU(U_PRIM);
    const bool primary = prev == PREV_SENSOR;     // sensor edge: implied state, see k_raygen
    const V3 ro = primary ? mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]) : load3(ps, F_OX);
    const V3 rd = load3(ps, F_DX);
    const float t_hit = PF(F_T);
    Col w_edge = (primary || DRAWS_ONLY) ? cone() : loadc(ps, F_WR);
    const float rr = (primary || DRAWS_ONLY) ? 1.0f : PF(F_RR);
    const float pdf_edge = (primary || DRAWS_ONLY) ? 1.0f : PF(F_PDF);
    Col beta = (primary || DRAWS_ONLY) ? cone() : loadc(ps, F_BR);
    Col L = (primary || DRAWS_ONLY) ? czero() : loadc(ps, F_LR);
    bool zeroed = (flags & ST_ZEROED) != 0u;
    const bool hit = prim >= 0;
    bool is_volume = false;
    V3 vpos = mk3(0.0f, 0.0f, 0.0f);
    if (MEDIUM) {
        // Edge::from_ray (paths/edge.rs:93-162): distance sampling up to the surface (or infinity on a miss)
        MediumSample ms = medium_sample(sc.medium, hit ? t_hit : kF32Max, PF(F_XI));
        w_edge = w_edge * ms.w;
        is_volume = !hit || !ms.exited;
        if (is_volume) vpos = ro + rd * ms.t;
    }
    bool ended = false;
    unsigned new_flags = ST_REGEN;
    if (!MEDIUM && !hit) {
        // edge without a next vertex: Edge::contribution = weight * rr * scene.enviroment_luminance(d) (edge.rs:201-210)
        ended = true;
        if constexpr (!DRAWS_ONLY) {
        // A zero term is still added as beta * 0: with a NaN / infinite throughput (hostile textures) the reference's recursion turns the
        // whole sample into NaN through `weight * evaluate(next)`, and the forward form keeps that by not skipping the product.
        Col contrib = (w_edge * rr) * (LIGHTS != LIGHTS_AREA_ONLY && sc.env_emitter >= 0 ? env_eval(sc, rd) : czero());      // enviroment_luminance is black without an environment
        const bool add_contrib = rc.has_min ? (depth - 1u) >= rc.min_depth : true;
        if (prev == PREV_SENSOR) {
            if (!is_zero(contrib) && add_contrib) L = L + contrib;
        } else if (!zeroed) {
            if (rc.strategy == RL_STRATEGY_EMITTER) contrib = czero();
            Col term = czero();
            if (!is_zero(contrib) && add_contrib) {
                float wmis = 1.0f;
                if (rc.strategy == RL_STRATEGY_ALL && (flags & ST_PDF_SA)) {
                    // pdf_emitter, `None` next vertex: direct_pdf of the environment (emitters.rs:18-46)
                    float p2 = (LIGHTS != LIGHTS_AREA_ONLY && (prev == PREV_SURFACE || prev == PREV_VOLUME)) ? env_direct_pdf(sc, rd) : 0.0f;
                    float total = (0.0f + pdf_edge) + p2;
                    wmis = div_rn(pdf_edge, total);
                }
                term = contrib * wmis;
            }
            L = L + beta * term;
        }
        storec(ps, F_LR, L);
        }
    }
    if (!ended) {
        const Col W = w_edge * rr;                // edge.weight * edge.rr_weight (Color * f32, guarded)
        SurfacePoint sp;
        const Material* mat = nullptr;
        MeshRecord mr;
        if (!is_volume) {
            sp = fill_intersection(sc, prim, PF(F_U), PF(F_V), ro, rd, t_hit);
            mr = sc.meshes[sp.mesh];
            mat = &sc.materials[mr.material];
        }
        // ---- contribution carried by the arriving edge (Edge::contribution -> Vertex::contribution)
        if constexpr (!DRAWS_ONLY) {
        Col emit = czero();
        if (!is_volume && (mr.flags & MESH_IS_LIGHT) && dot(sp.n_s, -rd) >= 0.0f) emit = mesh_emit<LIGHTS>(sc, mr, sp.has_uv, sp.uv);      // its.mesh.emit(&its.uv) (vertex.rs:69-82)
        Col contrib = W * emit;
        const unsigned cur = depth - 1u;          // evaluate()'s curr_depth of the origin vertex
        const bool add_contrib = rc.has_min ? cur >= rc.min_depth : true;
        if (prev == PREV_SENSOR) {
            if (!is_zero(contrib) && add_contrib) L = L + contrib;              // path.rs:152-166 (no MIS)
        } else if (!zeroed) {
            if (rc.strategy == RL_STRATEGY_EMITTER) contrib = czero();          // id_sampling 0 != 1
            Col term = czero();
            if (!is_zero(contrib) && add_contrib) {
                float wmis = 1.0f;
                if (rc.strategy == RL_STRATEGY_ALL && (flags & ST_PDF_SA)) {
                    // LightSamplingStrategy::pdf -> pdf_emitter (strategies/emitters.rs:10-92,250-282)
                    float p2 = 0.0f;
                    if (!is_volume && (mr.flags & MESH_IS_LIGHT) && (prev == PREV_SURFACE || prev == PREV_VOLUME))
                        p2 = light_direct_pdf<LIGHTS>(sc, mr, sc.tris[prim].tri, ro, sp.p, sp.n_g, rd, false, mk3(0.0f, 0.0f, 0.0f));   // n = None (emitters.rs:52-57)
                    float total = (0.0f + pdf_edge) + p2;
                    wmis = div_rn(pdf_edge, total);                             // balance heuristic (path.rs:80-98)
                }
                term = contrib * wmis;
            }
            L = L + beta * term;                                                // a zero term too: see the miss branch
        }
        beta = beta * W;
        if (rc.single_scattering && !is_volume) zeroed = true;                  // evaluate(): surface vertex => subtree is 0
        }

        // ---- expand the new vertex (generate(), strategies/mod.rs:35-80)
        const unsigned gen = depth + 1u;
        const bool expand = (rc.has_max ? gen < rc.max_depth : true) && gen < kDepthCap;
        if (expand) {
            n_vertices += 1;
            Rng rng = load_rng(ps, Q_R0);
            Col thr = primary ? cone() : loadc(ps, F_TR);
            const V3 vp = is_volume ? vpos : sp.p;
            const V3 d_in = -rd;
            // strategy 0: DirectionalSamplingStrategy::bounce (strategies/directional.rs:44-153)
            V2 s2; s2.x = rng_next_f32(rng); s2.y = rng_next_f32(rng);
            n_draws += 2;
            bool has_edge = false;
            bool sampled = false;
            Col sw = czero(); V3 sd_world = mk3(0.0f, 0.0f, 0.0f); float spdf = 0.0f; int spdf_kind = PDF_SOLID_ANGLE;
            if (is_volume) {
                phase_sample(sc.medium, d_in, s2, &sd_world, &sw, &spdf);
                sampled = true;
            } else {
                BsdfSample bs;
                if (bsdf_sample<MAT>(sc, *mat, sp.has_uv, sp.uv, sp.wi, s2, &bs)) {
                    sampled = true;
                    sw = bs.weight; spdf = bs.pdf; spdf_kind = bs.pdf_kind;
                    sd_world = to_world(sp.frame, bs.d);
                }
            }
            float rr_new = 1.0f;
            if (sampled) {
                thr = thr * sw;
                if (!is_zero(thr)) {
                    const bool do_rr = rc.has_rr ? rc.rr_depth <= gen : true;
                    bool alive = true;
                    if (do_rr) {
                        float q = rmin(channel_max(thr), 0.95f);
                        float x = rng_next_f32(rng);
                        n_draws++;
                        if (q < x) alive = false; else rr_new = div_rn(1.0f, q);
                    }
                    if (alive) {
                        thr = scale_unguarded(thr, rr_new);
                        has_edge = true;
                        if (MEDIUM) { PF(F_XI) = rng_next_f32(rng); n_draws++; }   // the new edge's medium.sample draw
                    }
                }
            }
            // strategy 1: LightSamplingStrategy::sample (strategies/emitters.rs:95-248)
            const bool use_light = rc.strategy != RL_STRATEGY_BSDF;
            const bool smooth = !is_volume && mat->smooth;
            bool shadow = false;
            if (DRAWS_ONLY) {
                if (use_light && !smooth) { rng_next_u64(rng); rng_next_u64(rng); rng_next_u64(rng); rng_next_u64(rng); n_draws += 4; }   // the light sample's four draws
            } else
            if (use_light && !smooth) {
                float a = rng_next_f32(rng);
                float b = rng_next_f32(rng);
                V2 c; c.x = rng_next_f32(rng); c.y = rng_next_f32(rng);
                n_draws += 4;
                n_shadow += 1;     // the reference always traces the shadow ray (emitters.rs:125-126)
                LightSample ls = sample_light<LIGHTS>(sc, vp, !is_volume, is_volume ? mk3(0.0f, 0.0f, 0.0f) : sp.n_s, a, b, c);   // Some(&its.n_s) | None
                if (ls.pdf != 0.0f) {
                    Col wl;
                    float p_dir;
                    if (is_volume) { wl = phase_eval(sc.medium, d_in, ls.d); p_dir = phase_pdf(sc.medium, d_in, ls.d); }   // (no environment with a medium)
                    else {
                        V3 wo = to_local(sp.frame, ls.d);
                        wl = bsdf_eval<MAT>(sc, *mat, sp.has_uv, sp.uv, sp.wi, wo, false);
                        // the MIS pdf is asked along Edge::from_vertex's own direction (p_light - p) / |..| (edge.rs:37-39):
                        // bitwise equal to ls.d for mesh lights, recomputed for the environment
                        V3 wo_edge = wo;
                        if (LIGHTS != LIGHTS_AREA_ONLY && ls.kind == EMITTER_ENV) { V3 ed = ls.p - vp; ed = ed / length(ed); wo_edge = to_local(sp.frame, ed); }
                        p_dir = bsdf_pdf<MAT>(sc, *mat, sp.has_uv, sp.uv, sp.wi, wo_edge, false);
                    }
                    if (MEDIUM) {
                        V3 dd = ls.p - vp;
                        wl = wl * medium_transmittance(sc.medium, dot(dd, ls.d));
                    }
                    Col c_l = ls.weight * wl * 1.0f;                       // contrib * weight * rr_weight (edge.rs:204)
                    const bool add_l = rc.has_min ? (gen - 1u) >= rc.min_depth : true;
                    if (!zeroed) {
                        Col term = czero();
                        if (!is_zero(c_l) && add_l) {
                            float wmis = 1.0f;
                            if (rc.strategy == RL_STRATEGY_ALL && (LIGHTS == LIGHTS_AREA_ONLY || ls.pdf_kind == PDF_SOLID_ANGLE)) {   // Discrete (point / directional): no MIS
                                float total = (0.0f + p_dir) + ls.pdf;
                                wmis = div_rn(ls.pdf, total);
                            }
                            term = c_l * wmis;
                        }
                        Col pending = beta * term;
                        // a zero contribution needs no visibility test: the image cannot change (beta * 0 is NaN for a NaN / infinite
                        // throughput — then the edge matters and is traced)
                        if (!is_zero(pending)) {
                            shadow = true;
                            store3(ps, F_SX, ls.p);
                            storec(ps, F_CR, pending);
                        }
                    }
                }
            }
            store_rng(ps, Q_R0, rng);
            store3(ps, F_OX, vp);
            new_flags = shadow ? ST_SHADOW : 0u;
            if (has_edge) {
                store3(ps, F_DX, sd_world);
                storec(ps, F_TR, thr);
                if constexpr (!DRAWS_ONLY) {
                    storec(ps, F_WR, sw);
                    PF(F_RR) = rr_new;
                    PF(F_PDF) = spdf;
                }
                PU(U_DEPTH) = gen;
                const unsigned kind = is_volume ? PREV_VOLUME : (smooth ? PREV_SURFACE_SMOOTH : PREV_SURFACE);
                new_flags |= ST_RAY | (kind << ST_PREV_SHIFT) | (spdf_kind == PDF_SOLID_ANGLE ? ST_PDF_SA : 0u) | (zeroed ? ST_ZEROED : 0u);
            } else new_flags |= ST_REGEN;
        }
        if constexpr (!DRAWS_ONLY) {
            storec(ps, F_BR, beta);
            storec(ps, F_LR, L);
        }
    }
    PU(U_FLAGS) = new_flags;
}

}  // namespace rl
