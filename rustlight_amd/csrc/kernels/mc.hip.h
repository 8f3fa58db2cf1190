// mc.hip.h — device side of the `ao` and `direct` integrators (ao.rs:20-70, direct.rs:21-233).
// Included by mc.hip after common.hip.h (McConst: launch.h).
#pragma once

namespace rl {

// ------------------------------------------------------------------------------------------
// k_pixel_mc<KIND> — the other two `compute_mc` integrators, one lane per work item
// (pixel, or block in reference-order mode), samples folded in order:
//   KIND 0  IntegratorAO::compute_pixel      src/integrators/ao.rs:20-70
//   KIND 1  IntegratorDirect::compute_pixel  src/integrators/direct.rs:21-233 (power heuristic, mod.rs:462-478)
RL_DEV float mis_weight_power(float pdf_a, float pdf_b) {
    if (pdf_a == 0.0f) return 0.0f;
    if (!finite_f(pdf_a) || !finite_f(pdf_b)) return 0.0f;
    float w = div_rn(pdf_a * pdf_a, pdf_a * pdf_a + pdf_b * pdf_b);
    return finite_f(w) ? w : 0.0f;
}
template <class Stack>
RL_DEV bool trace_closest(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, V3 o, V3 d, Hit& hit) {
    hit.t = kF32Max; hit.u = 0.0f; hit.v = 0.0f; hit.prim = -1;
    traverse<false>(recs, sc.root, mk3(sc.root_min[0], sc.root_min[1], sc.root_min[2]), mk3(sc.root_max[0], sc.root_max[1], sc.root_max[2]),
                    o, d, kEps, kF32Max, hit, stack);
    return hit.prim >= 0;
}
template <class Stack>
RL_DEV bool trace_visible(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, V3 p0, V3 p1) { return shadow_visible(sc, recs, stack, p0, p1); }

template <int KIND, class Stack>
RL_DEV Col mc_compute_pixel(const DeviceScene& sc, const SceneRecs& recs, const Stack& stack, const McConst& mp, unsigned px, unsigned py, Rng& rng,
                            unsigned& n_draws, unsigned& n_ext, unsigned& n_shadow, unsigned& n_vertices) {
    float u = (float)px + rng_next_f32(rng);
    float v = (float)py + rng_next_f32(rng);
    n_draws += 2;
    const V3 o = mk3(sc.camera.position[0], sc.camera.position[1], sc.camera.position[2]);
    const V3 d = camera_direction(sc, u, v);
    Hit hit;
    n_ext++;
    if (!trace_closest(sc, recs, stack, o, d, hit)) {
        if (KIND == 1 && sc.env_emitter >= 0) return env_eval(sc, d);   // scene.enviroment_luminance(ray.d)
        return czero();
    }
    const SurfacePoint sp = fill_intersection(sc, hit.prim, hit.u, hit.v, o, d, hit.t);
    if (KIND == 0) {
        if (!mp.normal_correction && sp.wi.z <= 0.0f) return czero();
        const bool flipped = mp.normal_correction && sp.wi.z <= 0.0f;
        V2 s2; s2.x = rng_next_f32(rng); s2.y = rng_next_f32(rng);
        n_draws += 2;
        V3 d_local = cosine_sample_hemisphere(s2);
        V3 d_world = flipped ? to_world(sp.frame, -d_local) : to_world(sp.frame, d_local);
        Hit h2;
        n_ext++;
        if (!trace_closest(sc, recs, stack, sp.p, d_world, h2)) return cone();
        if (!mp.has_max_distance) return czero();
        return h2.t > mp.max_distance ? cone() : czero();
    }
    // ---- direct
    Col l_i = czero();
    if (sp.wi.z <= 0.0f) return l_i;
    const MeshRecord mr = sc.meshes[sp.mesh];
    const Material& mat = sc.materials[mr.material];
    l_i = l_i + ((mr.flags & MESH_IS_LIGHT) ? mesh_emit(sc, mr, sp.has_uv, sp.uv) : czero());
    const float w_nb_bsdf = mp.nb_bsdf_samples == 0u ? 0.0f : div_rn(1.0f, (float)mp.nb_bsdf_samples);
    const float w_nb_light = mp.nb_light_samples == 0u ? 0.0f : div_rn(1.0f, (float)mp.nb_light_samples);
    n_vertices++;
    for (unsigned k = 0; k < mp.nb_light_samples; k++) {
        float a = rng_next_f32(rng);
        float b = rng_next_f32(rng);
        V2 c; c.x = rng_next_f32(rng); c.y = rng_next_f32(rng);
        n_draws += 4;
        LightSample ls = sample_light(sc, sp.p, true, sp.n_s, a, b, c);   // Some(&its.n_s) (direct.rs:64-70)
        V3 d_out_local = to_local(sp.frame, ls.d);
        if (ls.pdf == 0.0f) continue;
        n_shadow++;
        if (!trace_visible(sc, recs, stack, sp.p, ls.p)) continue;
        if (mat.smooth) continue;
        float pdf_bsdf = bsdf_pdf<-1>(sc, mat, sp.has_uv, sp.uv, sp.wi, d_out_local, false);
        float weight_light = ls.pdf_kind == PDF_SOLID_ANGLE ? mis_weight_power(ls.pdf * w_nb_light, pdf_bsdf * w_nb_bsdf) : 1.0f;
        l_i = l_i + weight_light * bsdf_eval<-1>(sc, mat, sp.has_uv, sp.uv, sp.wi, d_out_local, false) * w_nb_light * ls.weight;
    }
    for (unsigned k = 0; k < mp.nb_bsdf_samples; k++) {
        V2 s2; s2.x = rng_next_f32(rng); s2.y = rng_next_f32(rng);
        n_draws += 2;
        BsdfSample bs;
        if (!bsdf_sample<-1>(sc, mat, sp.has_uv, sp.uv, sp.wi, s2, &bs)) continue;
        V3 d_out_world = to_world(sp.frame, bs.d);
        Hit h2;
        n_ext++;
        if (trace_closest(sc, recs, stack, sp.p, d_out_world, h2)) {
            const SurfacePoint nx = fill_intersection(sc, h2.prim, h2.u, h2.v, sp.p, d_out_world, h2.t);
            const MeshRecord nm = sc.meshes[nx.mesh];
            if ((nm.flags & MESH_IS_LIGHT) && dot(nx.n_g, -d_out_world) > 0.0f) {
                float weight_bsdf = 1.0f;
                if (bs.pdf_kind == PDF_SOLID_ANGLE) {
                    float light_pdf = light_direct_pdf(sc, nm, sc.tris[h2.prim].tri, sp.p, nx.p, nx.n_g, d_out_world, true, sp.n_s);   // direct.rs:156-164
                    weight_bsdf = mis_weight_power(bs.pdf * w_nb_bsdf, light_pdf * w_nb_light);
                }
                l_i = l_i + weight_bsdf * bs.weight * mesh_emit(sc, nm, nx.has_uv, nx.uv) * w_nb_bsdf;
            }
        } else if (sc.env_emitter >= 0) {
            float weight_bsdf = bs.pdf_kind == PDF_SOLID_ANGLE ? mis_weight_power(bs.pdf * w_nb_bsdf, env_direct_pdf(sc, d_out_world) * w_nb_light) : 1.0f;
            l_i = l_i + weight_bsdf * bs.weight * env_eval(sc, d_out_world) * w_nb_bsdf;
        }
    }
    return l_i;
}

}  // namespace rl
