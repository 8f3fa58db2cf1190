// wavefront.h — test/debug hooks exported next to the public C-ABI (not part of the drop-in surface).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>

struct rl_context;
struct rl_scene;
void rl_set_error(const std::string& s);

extern "C" {
// runs div/sqrt/mul-add and the deterministic transcendentals on the device: out8 = 8 arrays of n
int rl_debug_numerics(int device, size_t n, const float* a, const float* b, float* out8);
// closest hits of a batch of rays through the tolerance build's traversal of a streaming scene (quantised BVH4, numerics = fast); steps = node trips per ray.
// RL_ERR_UNSUPPORTED for scenes staged in LDS (they have no BVH4).
int rl_debug_trace_batch_fast(rl_context* ctx, size_t n, const float* origins, const float* directions, float* t_out, int32_t* mesh_out, int32_t* tri_out, int32_t* steps_out);
// rl_trace_batch through the two-level node records (trace.hip.h: traverse2; off by default, RL_TWO_LEVEL) + node trips per ray; any_hit: t_inout = segment lengths in, 1 / 0 out
int rl_debug_trace_batch_two_level(rl_context* ctx, size_t n, const float* origins, const float* directions, float* t_inout, float* u_out, float* v_out,
                                   int32_t* mesh_out, int32_t* tri_out, int32_t* steps_out, int any_hit);
// host only: the two-level records against the BVH2 they are derived from (bvh.cpp)
int rl_debug_check_two_level(const rl_scene* scene, uint64_t* out6);
// rng_advance (kernels/rngjump.h) on the device: states_out[i] = sampler states_in[i] (4 x u64) after counts[i] more draws
int rl_debug_rng_advance(int device, size_t n, const uint64_t* states_in, const uint32_t* counts, uint64_t* states_out);
// host only: out[y * W + x] = 1 where every camera sample of the pixel takes exactly two draws (its rays cannot reach the scene's bounding box; k_stream_spec's shortcut)
int rl_debug_trivial_pixels(const rl_scene* scene, int has_max_depth, uint32_t max_depth, uint8_t* out);
int rl_debug_bvh_sizes(const rl_context* ctx, uint64_t* n_ref_nodes, uint64_t* n_prims, uint32_t* stack_depth, int* lds_scene);
// host-only: builds the BVH of `scene` and returns it in the reference's node shape (no GPU needed)
int rl_debug_bvh(const rl_scene* scene, uint64_t* n_nodes, uint64_t* n_prims, float* boxes, uint64_t* info, uint64_t* count,
                 int32_t* prim_mesh, int32_t* prim_tri);
// host-only: Camera::generate for one pixel position
int rl_debug_emitters_cdf(const rl_scene* scene, uint64_t* n_entries, float* cdf);
int rl_debug_ats(const rl_scene* scene, uint64_t* n_nodes, float* nodes16, uint64_t* n_lights, int32_t* light_emitter, int32_t* light_prim);
int rl_debug_camera_ray(const rl_scene* scene, float px, float py, float* origin, float* direction);
}
