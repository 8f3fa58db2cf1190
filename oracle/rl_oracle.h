/* oracle/rl_oracle.h — C API of the CPU oracle (TEST INFRASTRUCTURE, see rl_oracle.cpp header). */
#ifndef RL_ORACLE_H
#define RL_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/rustlight_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_path_params {
    uint32_t spp;
    int32_t has_min_depth; uint32_t min_depth;
    int32_t has_max_depth; uint32_t max_depth;
    int32_t has_rr_depth; uint32_t rr_depth;
    int32_t strategy;           /* rl_path_strategy */
    int32_t single_scattering;
    int32_t stream_mode;        /* rl_stream_mode */
    int32_t seed_variant;       /* 0 = PCG32 fill, 1 = SplitMix64 */
    uint32_t shard_index, shard_count;
    int32_t eval_order;         /* 0 = reference recursion (inner-first), 1 = forward (GPU accumulation order) */
} orc_path_params;

typedef struct orc_mc_params {   /* ao / direct integrators */
    uint32_t spp;
    int32_t stream_mode, seed_variant;
    uint32_t shard_index, shard_count;
    int32_t has_max_distance; float max_distance; int32_t normal_correction;   /* IntegratorAO */
    uint32_t nb_bsdf_samples, nb_light_samples;                                /* IntegratorDirect */
} orc_mc_params;

typedef struct orc_stats {
    uint64_t camera_samples, vertices, extension_rays, shadow_rays, rng_draws;
    uint32_t threads;
} orc_stats;

orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene* s);
int orc_scene_set_camera(orc_scene* sc, uint32_t w, uint32_t h, float fov, int fov_axis, const float* to_world, int flip);
int orc_scene_add_bitmap(orc_scene* sc, uint32_t w, uint32_t h, const float* rgb);
int orc_scene_add_mesh(orc_scene* sc, const float* vertices, size_t nv, const uint32_t* indices, size_t ntri,
                       const float* normals, const float* uv, const rl_bsdf_desc* bsdf, const float* emission);
int orc_scene_set_mesh_emission(orc_scene* sc, int mesh, int type, float scale, int bitmap_id);   /* 1 = EmissionType::HSV { scale }, 2 = Texture { scale, img } (geometry.rs:99-104, cli.rs:410-429) */
int orc_scene_set_medium(orc_scene* sc, const float* sigma_a, const float* sigma_s, int phase, float g);
int orc_scene_add_point_light(orc_scene* sc, const float* position, const float* intensity);
int orc_scene_add_directional_light(orc_scene* sc, const float* direction, const float* intensity);
int orc_scene_set_environment(orc_scene* sc, const float* rgb);
int orc_scene_set_environment_map(orc_scene* sc, uint32_t w, uint32_t h, const float* rgb);   /* lat-long texture, emitter.rs:300-425 */
int orc_scene_set_ats(orc_scene* sc, int build_ats);   /* Scene::build_emitters(build_ats), `-x ats` */
int orc_ats_probe(const orc_scene* sc, int kind, const float* in, float* out);
int orc_ats_dump(const orc_scene* sc, uint64_t* n_nodes, float* nodes16, uint64_t* n_lights, int32_t* light_emitter, int32_t* light_prim);
int orc_env_probe(const orc_scene* sc, int kind, const float* in, float* out);
int orc_scene_build(orc_scene* sc);

void orc_rng_seed(uint64_t seed, int variant, uint64_t* state_out);
uint64_t orc_rng_next_u64(uint64_t* state);
float orc_rng_next_f32(uint64_t* state);
size_t orc_block_count(uint32_t w, uint32_t h);
void orc_generate_block_seeds(uint64_t* master_state, uint32_t w, uint32_t h, uint64_t* seeds);

float orc_sinf(float x); float orc_cosf(float x); float orc_expf(float x); float orc_logf(float x);
float orc_powf(float x, float y); float orc_acosf(float x); float orc_atan2f(float y, float x);
void orc_math_batch(int fn, size_t n, const float* a, const float* b, float* out);

void orc_camera_generate(const orc_scene* sc, float px, float py, float* o, float* d);
void orc_scene_info(const orc_scene* sc, uint64_t* n_nodes, uint64_t* n_prims, uint64_t* n_emitters, float* bsphere);
void orc_bvh_dump(const orc_scene* sc, float* boxes, uint64_t* info, uint64_t* count, int32_t* prim_mesh, int32_t* prim_tri);
int orc_trace_batch(const orc_scene* sc, size_t n, const float* o, const float* d, int brute,
                    float* t_out, float* u_out, float* v_out, int32_t* mesh_out, int32_t* tri_out);
int orc_visible_batch(const orc_scene* sc, size_t n, const float* p0, const float* p1, uint8_t* out);
int orc_trace_full(const orc_scene* sc, const float* o, const float* d, float* out);
int orc_bsdf_probe(const orc_scene* sc, int mesh, int op, const float* wi, const float* wo_or_sample, float* out);
int orc_sample_light(const orc_scene* sc, const float* p, float r_sel, float r, float ux, float uy, float* out);
uint64_t orc_compute_pixel(const orc_scene* sc, const orc_path_params* pp, uint32_t ix, uint32_t iy, uint64_t* rng_state,
                           float* rgb, uint64_t* n_vertices, uint64_t* n_shadow);
int orc_render_mc(const orc_scene* sc, int kind, const orc_mc_params* mp, const uint64_t* block_seeds, size_t n_blocks,
                  float* out_rgb, int n_threads, orc_stats* stats);
int orc_render_path(const orc_scene* sc, const orc_path_params* pp, const uint64_t* block_seeds, size_t n_blocks,
                    float* out_rgb, int n_threads, orc_stats* stats);
#ifdef __cplusplus
}
#endif
#endif
