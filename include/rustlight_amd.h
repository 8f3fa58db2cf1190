/*
 * rustlight_amd.h — C-ABI of the MI355X-native drop-in for rustlight's `path` integrator.
 *
 * Everything here is plain C: POD structs, pointers and sizes.  No torch / HIP / C++ types
 * cross this boundary.  Each entry point cites the rustlight interface it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the Rust `extern "C"` block
 * and the `impl Integrator for ...` a rustlight maintainer would add on top of this header.
 *
 * Conventions
 *   - all functions return RL_OK (0) or a negative rl_status; they never throw or abort
 *     (the reference panics instead: src/scene_loader.rs:34, examples/cli.rs:36,348);
 *   - vectors are tightly packed f32 triples, matrices are column-major 4x4 f32 (cgmath);
 *   - colours are linear RGB f32 triples (src/structure.rs:105-110);
 *   - images are row-major, origin top-left, W*H*3 f32 (src/structure.rs:383-402).
 *
 * Execution options.  None of them changes a result; they exist for the tests and for measurements and are NOT part of the
 * drop-in surface.  rl_context_create copies them ONCE from the process environment (RL_<NAME>) into the context; afterwards
 * rl_context_set_option(ctx, "<name>", value) changes them per context, and no render entry point ever reads the environment
 * (a render works on the copy of the table it took when it started).  The full list is rustlight_amd/csrc/kernels/knobs.h; the ones the tests use:
 *   force_streaming = 1       (creation time: environment RL_FORCE_STREAMING only) keep small scenes out of LDS — the kernels that stream the BVH
 *                             from L2 / HBM on scenes the oracle finishes in seconds
 *   generic_lights = 1        (creation time: RL_GENERIC_LIGHTS) do not specialise the NEE code for area-light-only scenes
 *   item_shift = k            reference-order streams: one block chain per 2^k lanes
 *   ref_single_pass = 1       reference-order streams through the persistent kernel in ONE pass (the form of rounds 1-2) instead of
 *                             chain pass + per-sample evaluation
 *   chain_serial = 1          the chain pass by k_stream_chain (one lane per block) instead of k_stream_spec; spec_force = 1: the opposite
 *   spec_draws_per_sample = x the draws a camera sample takes on this scene (the choice between the two chain kernels; default 150 with a medium, 12 without —
 *                             rl_render_stats.rng_draws / camera_samples of an earlier render is the measured figure)
 *   chain_no_pre = 1, chain_no_treelets = 1     k_stream_chain without its lane-parallel records / treelet blocks
 *   no_overlap = 1            the evaluation pass after the chain pass instead of beside it
 *   state_budget_mb = n       MB the recorded sampler states of the two-pass form may take (default 24 GB): small values force several chunks
 *   fused_dynamic = 0|1       persistent kernel: static tile order / work items from the atomic dispenser
 *   no_events = 1             no HIP events around the kernels (rl_render_stats.ms_* stay 0)
 * rl_multi_* reads RL_MULTI_FORCE_HOST_MERGE=1, RL_MULTI_NO_FALLBACK=1, RL_MULTI_REDUCE_TIMEOUT_S=s when the communicator is built / a reduce runs: see rl_multi_describe.
 */
#ifndef RUSTLIGHT_AMD_H
#define RUSTLIGHT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rl_status {
    RL_OK = 0,
    RL_ERR_INVALID_ARGUMENT = -1,
    RL_ERR_NO_DEVICE = -2,       /* HIP runtime / GPU missing: the product path never falls back to CPU */
    RL_ERR_HIP = -3,             /* a HIP call failed; see rl_last_error() */
    RL_ERR_NOT_BUILT = -4,       /* scene used before rl_scene_build_emitters() */
    RL_ERR_IO = -5,
    RL_ERR_PARSE = -6,
    RL_ERR_UNSUPPORTED = -7,
    RL_ERR_NO_EMITTER = -8       /* NEE requested on a scene without emitters (reference: warn + crash, src/scene.rs:96-99) */
} rl_status;

/* ---------------------------------------------------------------- materials ------------- */

/* BSDFColor (src/bsdfs/mod.rs:11-29) */
typedef enum rl_tex_type {
    RL_TEX_CONSTANT = 0,
    RL_TEX_CHECKERBOARD = 1,
    RL_TEX_GRID = 2,
    RL_TEX_BITMAP = 3
} rl_tex_type;

typedef struct rl_color_desc {
    int32_t type;          /* rl_tex_type */
    float color0[3];       /* Constant colour / checker+grid colour 0 */
    float color1[3];
    float offset[2];
    float scale[2];
    float line_width;      /* grid only */
    int32_t bitmap_id;     /* RL_TEX_BITMAP: id returned by rl_scene_add_bitmap */
} rl_color_desc;

/* concrete `impl BSDF` types (src/bsdfs/{diffuse,phong,metal,glass,substrate}.rs) */
typedef enum rl_bsdf_type {
    RL_BSDF_DIFFUSE = 0,
    RL_BSDF_PHONG = 1,
    RL_BSDF_METAL = 2,
    RL_BSDF_GLASS = 3,
    RL_BSDF_SUBSTRATE = 4
} rl_bsdf_type;

/* MicrofacetType (src/bsdfs/distribution.rs:13-17); 0 = `distribution: None` */
typedef enum rl_microfacet_type {
    RL_MICROFACET_NONE = 0,
    RL_MICROFACET_BECKMANN = 1,
    RL_MICROFACET_GGX = 2
} rl_microfacet_type;

typedef struct rl_bsdf_desc {
    int32_t type;                 /* rl_bsdf_type */
    rl_color_desc diffuse;        /* Diffuse.diffuse, Phong.diffuse, Substrate.diffuse */
    rl_color_desc specular;       /* Phong.specular, Metal.specular, Substrate.specular, Glass.specular_reflectance */
    rl_color_desc transmittance;  /* Glass.specular_transmittance */
    rl_color_desc eta;            /* Metal.eta */
    rl_color_desc k;              /* Metal.k */
    float exponent;               /* Phong.exponent */
    float weight_specular;        /* Phong.weight_specular */
    int32_t distribution;         /* one of rl_microfacet_type; Metal and Substrate only */
    float alpha_u, alpha_v;
    float glass_eta;              /* BSDFGlass.eta = int_ior / ext_ior (src/bsdfs/glass.rs:44-49) */
} rl_bsdf_desc;

/* PhaseFunction (src/volume.rs:12-16) */
typedef enum rl_phase_type { RL_PHASE_ISOTROPIC = 0, RL_PHASE_HG = 1 } rl_phase_type;

/* ---------------------------------------------------------------- scene ----------------- */

/* Opaque host-side scene: the flattened counterpart of `struct Scene` (src/scene.rs:16-30). */
typedef struct rl_scene rl_scene;

/* Scene { .. } literal (src/scene_loader.rs:302-312): empty scene, nb_samples = 1. */
int rl_scene_create(rl_scene** out);
void rl_scene_destroy(rl_scene* scene);

/* Camera::new(img, fov, mat, flip) (src/camera.rs:31-67).
 * fov_axis: 0 = Fov::X, 1 = Fov::Y (the reference multiplies Fov::Y by the aspect ratio, camera.rs:41-44).
 * to_world: column-major camera-to-world matrix.  flip: true for the Mitsuba loader, false for PBRT. */
int rl_scene_set_camera(rl_scene* scene, uint32_t width, uint32_t height, float fov_degrees,
                        int fov_axis, const float to_world[16], int flip);

/* The camera as rustlight ALREADY HOLDS it (SURVEY.md 8(b) SceneDesc: `sample_to_camera[16], to_world[16]`): the two matrices Camera::generate reads
 * (src/camera.rs:81-91: `sample_to_camera.transform_point(px / img)`, `to_world.transform_vector(d)`, position = to_world * origin, camera.rs:140-142),
 * column-major as cgmath stores them (`let m: &[f32; 16] = camera.sample_to_camera.as_ref()`).  This is the entry a Rust host should use: nothing of
 * Camera::new (cgmath's `perspective`, `inverse_transform`, the Fov::Y x aspect rule — camera.rs:31-67) is re-derived on this side, so the rays are
 * rustlight's own by construction.  The two fields are private in rustlight: the host-side patch adds two accessors.
 * rl_scene_set_camera (below, fov / flip: what the scene FILES carry) stays for the loaders and derives the same two matrices itself. */
int rl_scene_set_camera_matrices(rl_scene* scene, uint32_t width, uint32_t height, const float sample_to_camera[16], const float to_world[16]);
/* The matrices the scene's camera uses (whichever call set it): sample_to_camera, to_world (column-major), and Camera::position. */
int rl_scene_get_camera_matrices(const rl_scene* scene, float sample_to_camera[16], float to_world[16], float position[3]);

/* EmissionType (src/geometry.rs:99-104) of a light mesh: the constant colour rl_scene_add_mesh gave it, or one of the two uv-dependent kinds
 * Mesh::emit evaluates (geometry.rs:184-206): HSV { scale } = scale * (x, 1 - x, 0) with x = |uv.x| % 1, Texture { scale, img } = scale * img.pixel_uv(uv).
 * The mesh must be a light and — for the two uv-dependent kinds — carry uv coordinates (the reference's `uv.unwrap()` panics otherwise).
 * Emitter::flux takes Color::value(scale) for them (emitter.rs:591-599).  Call before rl_scene_build_emitters. */
typedef enum rl_emission_type { RL_EMISSION_COLOR = 0, RL_EMISSION_HSV = 1, RL_EMISSION_TEXTURE = 2 } rl_emission_type;
int rl_scene_set_mesh_emission(rl_scene* scene, uint32_t mesh, int type, float scale, int bitmap_id);
/* The CLI's `-x hvs-light` / `-x texture-light` (examples/cli.rs:410-429): EVERY light mesh becomes HSV { scale } / Texture { scale, img = bitmap_id } with
 * scale = Color::luminance of its colour (1 if it already was uv-dependent).  `type`: RL_EMISSION_HSV or RL_EMISSION_TEXTURE. */
int rl_scene_override_light_emission(rl_scene* scene, int type, int bitmap_id);

/* Camera::scale_image (src/camera.rs:73-78), the CLI's -s flag. */
int rl_scene_scale_image(rl_scene* scene, float scale);

/* Mesh::new(name, vertices, indices, normals, uv) + `.bsdf = ...` + `.emission = EmissionType::Color`
 * (src/geometry.rs:122-182, src/scene_loader.rs:134-150).  normals / uv / emission may be NULL.
 * Returns the mesh index (>= 0) or a negative rl_status.  Meshes keep insertion order
 * (it fixes the emitter order, src/scene.rs:64-69, and the BVH primitive order, src/accel.rs:206-219). */
int rl_scene_add_mesh(rl_scene* scene, const float* vertices, size_t n_vertices,
                      const uint32_t* indices, size_t n_triangles, const float* normals,
                      const float* uv, const rl_bsdf_desc* bsdf, const float* emission_rgb);

/* Bitmap used by BSDFColor::Bitmap (src/bsdfs/mod.rs:13-15): row-major W*H*3 f32.  Returns its id. */
int rl_scene_add_bitmap(rl_scene* scene, uint32_t width, uint32_t height, const float* rgb);

/* scene.volume = Some(HomogenousVolume{..}) — the CLI's `-m sigma_s[:sigma_a[:g]]` (examples/cli.rs:355-399).
 * sigma_t = sigma_a + sigma_s is formed here exactly as cli.rs:383-385 does. */
int rl_scene_set_medium(rl_scene* scene, const float sigma_a[3], const float sigma_s[3],
                        int phase_type, float g);

/* Non-mesh emitters (`EmittersState::Unbuild`, src/scene_loader.rs:177-204): PointEmitter { intensity, position }
 * (src/emitter.rs:183-250) and DirectionalLight { direction, intensity } (src/emitter.rs:96-181), kept in
 * insertion order after the emissive meshes and the environment. */
int rl_scene_add_point_light(rl_scene* scene, const float position[3], const float intensity[3]);
int rl_scene_add_directional_light(rl_scene* scene, const float direction[3], const float intensity[3]);
/* scene.emitter_environment = EnvironmentLight { luminance: EnvironmentLightColor::Constant(rgb) }
 * (src/emitter.rs:300-568, src/scene_loader.rs:205-224).  Not combinable with a medium (the reference
 * asserts, src/paths/edge.rs:94). */
int rl_scene_set_environment(rl_scene* scene, const float rgb[3]);
/* EnvironmentLightColor::Texture { image, image_cdf } = EnvironmentLightColor::new_texture(image) (src/emitter.rs:340-353,
 * src/scene_loader.rs:259-271): lat-long image, z up, w x h RGB f32 row-major (row 0 = theta 0), importance sampled by
 * luminance x sin(theta) through a Distribution2D (src/math.rs:489-532); nearest-texel lookups as in the reference. */
int rl_scene_set_environment_map(rl_scene* scene, uint32_t w, uint32_t h, const float* rgb);

/* Scene::build_emitters(false) (src/scene.rs:53-123): scene bounding sphere, emitter list
 * (emissive meshes in mesh order), CDF over flux().channel_max().  The ATS light tree
 * (`-x ats`) is out of scope. */
int rl_scene_build_emitters(rl_scene* scene);
/* The `build_ats` argument of Scene::build_emitters (src/scene.rs:53,118-120; CLI `-x ats`): when set, the next
 * rl_scene_build_emitters() also builds the LightSamplerATS light tree over the emissive triangles (src/emitter.rs:1117-1292)
 * and light sampling / its MIS pdf descend that tree (importance_point, emitter.rs:1024-1086) instead of the flux cdf.
 * Every emitter must then be an emissive mesh (the reference asserts is_surface()). */
int rl_scene_enable_ats(rl_scene* scene, int build_ats);

/* SceneLoaderManager::load(path, use_shading_normals) for the `.pbrt` subset the reference's
 * PBRT loader consumes (src/scene_loader.rs:77-315): Transform/LookAt/Camera perspective/Film,
 * MakeNamedMaterial matte|..., NamedMaterial, Shape trianglemesh, AreaLightSource diffuse. */
int rl_scene_load_pbrt(const char* path, int use_shading_normals, rl_scene** out);
/* MTSSceneLoader::load (src/scene_loader.rs:318-795): the Mitsuba 0.5/0.6 XML subset the reference consumes — one perspective
 * sensor (flip = true), shapes obj | ply | serialized | rectangle | sphere with bsdf / area emitter / toWorld, bsdf_mts
 * materials (src/bsdfs/mod.rs:499-612), point emitters, the first homogeneous medium.  Needs rl_scene_build_emitters(). */
int rl_scene_load_mitsuba(const char* path, int use_shading_normals, rl_scene** out);
/* SceneLoaderManager::load (src/scene_loader.rs:27-58): picks the loader by extension (.pbrt | .xml). */
int rl_scene_load(const char* path, int use_shading_normals, rl_scene** out);

int rl_scene_image_size(const rl_scene* scene, uint32_t* width, uint32_t* height);
int rl_scene_counts(const rl_scene* scene, uint64_t* n_meshes, uint64_t* n_triangles,
                    uint64_t* n_emitters);

/* ---------------------------------------------------------------- sampler --------------- */

/* IndependentSampler (src/samplers/independent.rs:5-34) over rand 0.8.5 SmallRng = Xoshiro256++.
 * The reference trait object hides the raw u64 stream, so the drop-in owns the concrete sampler. */
typedef struct rl_sampler {
    uint64_t s[4];
} rl_sampler;

/* SmallRng::seed_from_u64(seed) as the CLI's `-r independent:SEED` does (examples/cli.rs:886-890).
 * variant 0 = rand_core 0.6.4 default (PCG32 fill), 1 = SplitMix64 (see oracle header for why both). */
void rl_sampler_seed(rl_sampler* sampler, uint64_t seed, int variant);
uint64_t rl_sampler_next_u64(rl_sampler* sampler);
/* Sampler::next() (samplers/independent.rs:9-11): rng.gen::<f32>() */
float rl_sampler_next_f32(rl_sampler* sampler);

/* ---------------------------------------------------------------- integrator ------------ */

/* IntegratorPathTracingStrategies (src/integrators/explicit/path.rs:9-13) */
typedef enum rl_path_strategy { RL_STRATEGY_ALL = 0, RL_STRATEGY_BSDF = 1, RL_STRATEGY_EMITTER = 2 } rl_path_strategy;

/* How the per-block random stream is mapped onto GPU lanes:
 *  RL_STREAM_REFERENCE_ORDER: one serial stream per 16x16 block, consumed over (iy, ix, sample)
 *      exactly as compute_mc does (src/integrators/mod.rs:420-435) — equals rustlight proper.  Runs in two passes on the
 *      GPU: a draw-count walk of every block's stream records the sampler state at the start of each camera sample (what is
 *      serial is only how many numbers a sample takes; since round 4 that walk is speculative: per-pixel windows of the block's stream walked by
 *      every lane, the true chain threaded through them), then all samples are evaluated in parallel from those states; 5x
 *      slower than RL_STREAM_PER_SAMPLE on the Cornell box at 1080p x 128 spp, same image and counters as the one-lane-per-block walk.
 *  RL_STREAM_PER_SAMPLE: the block stream is forked with the reference's own clone_box rule
 *      (samplers/independent.rs:18-22) once per pixel and once per sample — throughput mode. */
typedef enum rl_stream_mode { RL_STREAM_REFERENCE_ORDER = 0, RL_STREAM_PER_SAMPLE = 1 } rl_stream_mode;
typedef enum rl_numerics { RL_NUMERICS_EXACT = 0, RL_NUMERICS_FAST = 1 } rl_numerics;

/* struct IntegratorPathTracing (explicit/path.rs:14-20) + scene.nb_samples + sharding. */
typedef struct rl_path_params {
    uint32_t spp;              /* scene.nb_samples (CLI global -n) */
    int32_t has_min_depth;     /* Option<u32>: 0 = None */
    uint32_t min_depth;
    int32_t has_max_depth;
    uint32_t max_depth;
    int32_t has_rr_depth;
    uint32_t rr_depth;
    int32_t strategy;          /* rl_path_strategy */
    int32_t single_scattering;
    int32_t stream_mode;       /* rl_stream_mode */
    int32_t seed_variant;      /* 0 = PCG32 fill, 1 = SplitMix64; used when forking child streams */
    /* multi-GPU sharding of the 16x16 blocks: this context renders blocks with
     * b % shard_count == shard_index, b = (ix/16)*ceil(H/16) + iy/16 (creation order, mod.rs:357-358). */
    uint32_t shard_index;
    uint32_t shard_count;
    /* tuning: number of path slots resident on the device (0 = auto). Does not change results. */
    uint32_t pool_slots;
    /* 0 = auto, 1 = wavefront stage kernels (raygen / extend / shade / shadow per iteration, state in HBM),
     * 2 = persistent fused kernel (same stages in one launch, state in registers; the BSDF code is specialised when the scene has
     * one BSDF type and switches per vertex otherwise).  Auto takes the fused kernel whenever pool_slots = 0 (reference-order
     * streams too: k_stream_chain walks the block streams — one per 16x16 block, dealt to every 32nd lane so that all SIMDs have waves — and
     * the fused kernel evaluates the samples from the recorded states) and the wavefront kernels otherwise.  Does not change results. */
    uint32_t pipeline;
    /* per-sample stream mode: lanes working on one pixel at a time (sample s of a pixel runs on lane s % sample_split; the
     * per-sample radiances are parked in HBM and added up in sample order afterwards, so the sum keeps the reference's
     * association). 0 = auto, 1 = one lane per pixel. Does not change results. */
    uint32_t sample_split;
    /* rl_numerics: 0 = exact (default; every f32 operation as the reference performs it — the only mode the parity tests bless),
     * 1 = fast (opt-in: FMA contraction, a * v_rcp(b) / raw v_sqrt / v_rsq (1 ulp, no refinement step) instead of the IEEE divide / sqrt
     * sequences, hardware sin / cos / exp2 / log2; the RNG sequence stays bit-exact, pixels agree with the exact mode within
     * BASELINE.json's per-pixel L2 tolerance in the MEAN, not bit for bit and not for every pixel on glossy scenes — DESIGN.md §2
     * "Tolerance mode").  With RL_STREAM_REFERENCE_ORDER the first flipped decision of a block shifts the rest of that block's stream: the render is then
     * statistically, not seed-for-seed, the exact build's. */
    uint32_t numerics;
} rl_path_params;

void rl_path_params_default(rl_path_params* params);   /* CLI defaults: examples/cli.rs:53-61,167-168; stream_mode = RL_STREAM_REFERENCE_ORDER */

/* Counters returned by a render (device atomics). */
typedef struct rl_render_stats {
    uint64_t camera_samples;      /* W*H*spp of this shard */
    uint64_t vertices;            /* expanded non-sensor vertices (Sigma V) */
    uint64_t extension_rays;      /* closest-hit rays traced (primary + bounce) */
    uint64_t shadow_rays;         /* NEE visibility rays traced (Sigma S) */
    uint64_t rng_draws;           /* f32 draws consumed */
    uint64_t iterations;          /* wavefront iterations */
    uint64_t kernel_launches;
    double render_ms;             /* wall time of the call (the reference's own timed region, mod.rs:324-334) */
    /* per-kernel accumulated device time from HIP events on the render stream (ms) */
    double ms_raygen, ms_extend, ms_shade, ms_shadow, ms_prepass, ms_other;   /* ms_other = the fused kernel; ms_prepass = k_stream_chain, the draw-count pass of reference-order streams */
    uint64_t n_extend_launches;
    uint64_t reserved[4];         /* k_stream_spec: samples walked speculatively / serially / by the probes, lanes per block (0: k_stream_chain ran) */
    /* reference-order streams in two passes (round 6): */
    uint32_t chunks;              /* chunks of block cursors the frame was cut into (the recorded sampler states fit their buffer); 0 = not the two-pass form */
    uint32_t overlapped;          /* 1: the evaluation pass ran BESIDE the chain pass (every chunk) — then ms_other is only the part of it left after the chain pass
                                   * had ended; 0: after it */
    double ms_eval_span;          /* the evaluation pass from its first launch to its last completion (overlapped: host clock over launches on several streams;
                                   * otherwise = ms_other): what a roofline of k_path_fused must divide by */
} rl_render_stats;

/* Opaque device context: BVHAccel::new(scene) (src/accel.rs:202-239) + flattened scene in HBM. */
typedef struct rl_context rl_context;

/* Number of HIP devices visible to the process (0 and RL_ERR_NO_DEVICE when there is none). */
int rl_device_count(int* count);

/* IntegratorType::compute's untimed prologue (src/integrators/mod.rs:280): builds the BVH2
 * (full-sweep SAH, leaf <= 2) on the host and uploads scene + BVH to `device` (HIP ordinal).
 * Fails with RL_ERR_NO_DEVICE when no GPU is present — there is no CPU fallback. */
int rl_context_create(const rl_scene* scene, int device, rl_context** out);
void rl_context_destroy(rl_context* ctx);
/* Execution options of a context (see the list at the top of this header): `value` as text, NULL = back to the default.  RL_ERR_INVALID_ARGUMENT for an unknown
 * name, RL_ERR_UNSUPPORTED for the two creation-time options.  Not to be called while a render runs on the same context.  rl_context_get_option returns the text
 * an option holds, NULL when it is not set.  (No counterpart in rustlight: these are test / measurement hooks of this implementation.) */
int rl_context_set_option(rl_context* ctx, const char* name, const char* value);
const char* rl_context_get_option(const rl_context* ctx, const char* name);
const char* rl_last_error(void);

/* generate_img_blocks (src/integrators/mod.rs:351-374): number of <=16x16 blocks, and the
 * per-block seeds `master.next_u64()` drawn in creation order (x-major). Advances `master`. */
size_t rl_block_count(uint32_t width, uint32_t height);
int rl_generate_block_seeds(rl_sampler* master, uint32_t width, uint32_t height,
                            uint64_t* seeds_out, size_t n_blocks);

/* Integrator::compute for IntegratorPathTracing (explicit/path.rs:186-196 -> compute_mc,
 * integrators/mod.rs:403-450).  Renders this shard's blocks into `out_rgb` (W*H*3 f32; pixels of
 * other shards are written as 0 so that a sum over shards is the full image).
 * out_is_device != 0: `out_rgb` is a device pointer on the context's device (e.g. a torch tensor
 * that is then reduced with RCCL); otherwise a host pointer (the framebuffer download is timed).
 * `stream`: a hipStream_t to enqueue on, or NULL for the context's own stream.  Blocking.
 * In RL_STREAM_REFERENCE_ORDER (two passes through the persistent kernel) the call also launches on low-priority streams the context owns: the evaluation pass runs beside the
 * chain pass, launched by THIS thread over the blocks the chain kernel reports complete through a few words of mapped host memory (the thread polls them every 50 us until the
 * chain pass has ended; nothing on the device waits).  The image and the counters are those of the two passes back to back (RL_NO_OVERLAP=1: a test knob that keeps them so).
 * Frames in flight: contexts share nothing mutable (each owns its stream and buffers; rl_last_error is per thread), so independent frames may be
 * rendered concurrently from several host threads, one context each, on the same device — a frame's render is a chain of dependent launches whose
 * tail leaves much of the chip idle, and another context's frame fills it (Cornell box 1080p x 128 spp in reference-order streams: 1.07 -> 1.4 G
 * samples/s with two or three in flight; the images are those of one frame after the other).  What rustlight's `-a` wrapper (avg.rs:5-131: N independent
 * renders of one scene) can use as it is; host mirrors: integrator.hpp IntegratorPathTracing::frames_in_flight, rustlight-amd --frames-in-flight K. */
int rl_render_path(rl_context* ctx, const rl_path_params* params, const uint64_t* block_seeds,
                   size_t n_blocks, float* out_rgb, int out_is_device, void* stream,
                   rl_render_stats* stats);

/* Frames in flight behind one call (the progressive wrappers' passes, avg.rs:5-131 / equal_time.rs:4-66: N independent renders of one scene): frame f — block
 * seeds `block_seeds[f]`, host image `out_rgb[f]` (W*H*3 f32) — renders on `ctxs[f % k]` from host thread f % k, k = min(n_ctx, n_frames); `ctxs` are distinct
 * contexts of the same scene.  Returns when every frame is done; the images (and `stats[f]`, if not NULL) are those of `n_frames` rl_render_path calls one after
 * the other.  On an error the first failing thread's code is returned and rl_last_error carries its message; the other threads stop before their next frame
 * (frames not yet started are not rendered).  Every context in flight keeps its own render buffers (several GB at 1080p x 128 spp in reference-order streams)
 * until it is destroyed.  The process environment must not be changed (setenv / putenv) while frames are in flight: the library reads its test-only RL_* knobs
 * with getenv during a render. */
int rl_render_path_frames(rl_context* const* ctxs, size_t n_ctx, const rl_path_params* params,
                          const uint64_t* const* block_seeds, size_t n_blocks, size_t n_frames,
                          float* const* out_rgb, rl_render_stats* stats);

/* ---- the whole scene as ONE plain-old-data description ("SceneDesc POD: counts + pointers") ------------------
 * What a Rust host fills from `&Scene` (src/scene.rs:16-30) in one go instead of the builder calls above; the arrays are only read during
 * the call.  Equivalent to: rl_scene_create, rl_scene_set_camera_matrices (has_camera_matrices) or rl_scene_set_camera, rl_scene_add_bitmap (in order: their ids are 0, 1, ...), rl_scene_add_mesh
 * (in order), rl_scene_set_medium, rl_scene_add_point_light / _directional_light (in order), rl_scene_set_environment[_map],
 * rl_scene_enable_ats, rl_scene_build_emitters — with the same checks and error codes.
 * The structs below carry no size or version member: ZERO-INITIALISE them (memset / `= {0}` / Rust `std::mem::zeroed()`) before filling the fields you know —
 * fields added later (round 4: emission_type / _scale / _bitmap_id, has_camera_matrices / sample_to_camera / to_world) then read as 0 = the old behaviour
 * (EmissionType::Color, Camera::new from the scalar parameters); a consumer built against an older header must be rebuilt (the layout is checked by
 * tests/test_abi_layout.py against this header, the ctypes mirror and the Rust block of INTEGRATION.md). */
typedef struct rl_mesh_desc {          /* struct Mesh (src/geometry.rs:107-119) */
    const float* vertices; size_t n_vertices;      /* xyz */
    const uint32_t* indices; size_t n_triangles;   /* 3 per triangle */
    const float* normals;                          /* xyz per vertex or NULL */
    const float* uv;                               /* uv per vertex or NULL */
    rl_bsdf_desc bsdf;
    int32_t has_emission; float emission_rgb[3];   /* EmissionType::Color */
    int32_t emission_type; float emission_scale; int32_t emission_bitmap_id;   /* rl_emission_type; != RL_EMISSION_COLOR: as rl_scene_set_mesh_emission (needs has_emission and uv) */
} rl_mesh_desc;
typedef struct rl_bitmap_desc { uint32_t width, height; const float* rgb; } rl_bitmap_desc;
typedef struct rl_light_desc { int32_t kind;       /* 0 = PointEmitter { position = a }, 1 = DirectionalLight { direction = a } */
                               float a[3]; float intensity[3]; } rl_light_desc;
typedef struct rl_scene_desc {
    uint32_t width, height; float fov_degrees; int32_t fov_axis; float to_world[16]; int32_t flip;   /* Camera::new */
    const rl_mesh_desc* meshes; size_t n_meshes;
    const rl_bitmap_desc* bitmaps; size_t n_bitmaps;
    const rl_light_desc* lights; size_t n_lights;
    int32_t has_environment; float environment_rgb[3];            /* EnvironmentLightColor::Constant */
    uint32_t env_map_width, env_map_height; const float* env_map_rgb;   /* EnvironmentLightColor::Texture (0 x 0 / NULL: none) */
    int32_t has_medium; float sigma_a[3], sigma_s[3]; int32_t phase_type; float g;   /* HomogenousVolume */
    int32_t build_ats;                                             /* Scene::build_emitters(build_ats) */
    /* != 0: the camera is (width, height, sample_to_camera, to_world) as rl_scene_set_camera_matrices takes it — rustlight's own matrices; fov_degrees,
     * fov_axis and flip are ignored.  0: Camera::new from fov / flip (what scene files carry). */
    int32_t has_camera_matrices; float sample_to_camera[16];
} rl_scene_desc;
int rl_scene_create_from_desc(const rl_scene_desc* desc, rl_scene** out);

/* ---- several GPUs of one node behind one call ------------------------------------------------------
 * The reference merges its per-block bitmaps with `accumulate_bitmap` (src/integrators/mod.rs:445-448); here every GPU renders the
 * blocks b % N == g into its own zeroed framebuffer in HBM and ONE ncclReduce(sum, root = first device) over xGMI merges them —
 * device to device, no per-GPU download, no host adds; sums with zeros are exact, so the image equals the 1-GPU image bit for
 * bit.  One process: rl_multi owns N device contexts (BVHAccel::new once per device, untimed), an RCCL communicator clique
 * (ncclCommInitAll) and the per-device framebuffers.  `devices` = HIP ordinals, one per shard (NULL: round-robin over the
 * visible devices).  Shards that share a device (more shards than GPUs: a plumbing mode for tests) are added on that device
 * before the reduce; the communicator spans the distinct devices. */
typedef struct rl_multi rl_multi;
int rl_multi_create(const rl_scene* scene, const int* devices, int n_shards, rl_multi** out);
void rl_multi_destroy(rl_multi* m);
int rl_multi_info(const rl_multi* m, int* n_shards, int* n_comm_ranks, int* rccl_version);
/* Diagnostics as one JSON object (NUL-terminated, `capacity` bytes available): shards, RCCL version, communicator ranks, how the framebuffers are
 * merged ("ncclReduce(sum) onto device D" or "host sum (<why>)": when the communicator cannot be built or a reduce fails / exceeds
 * RL_MULTI_REDUCE_TIMEOUT_S the shards are still rendered on their GPUs and their sums added on the host — same bits —, with a line on stderr;
 * RL_MULTI_NO_FALLBACK=1 turns that into RL_ERR_HIP), and per device: name, CUs, shards, hipDeviceCanAccessPeer towards every other device; of the
 * last render: wall ms of the merge step and kernel ms per shard. */
int rl_multi_describe(const rl_multi* m, char* buf, size_t capacity);
/* Device and counters of one shard of the last rl_multi_render_path call (per-GPU kernel time: stats->ms_other [+ ms_prepass]). */
int rl_multi_shard_stats(const rl_multi* m, int shard, int* device, rl_render_stats* stats);
/* Integrator::compute over all shards: params->shard_index / shard_count are set per GPU by the call; `out_rgb` is a HOST
 * buffer of W*H*3 f32 (the framebuffer download from the root GPU is part of the call); `stats` = sums over the shards
 * (render_ms, ms_other: maximum).  Blocking. */
int rl_multi_render_path(rl_multi* m, const rl_path_params* params, const uint64_t* block_seeds, size_t n_blocks,
                         float* out_rgb, rl_render_stats* stats);

/* ---- the two other `compute_mc` integrators that reuse this path's kernels ---- */

/* struct IntegratorAO { max_distance: Option<f32>, normal_correction } (src/integrators/ao.rs:4-7; CLI `ao -d 1.0 -n`,
 * examples/cli.rs:149-154,856-865) and struct IntegratorDirect { nb_bsdf_samples, nb_light_samples }
 * (src/integrators/direct.rs:5-8; CLI `direct -b 1 -l 1`). */
typedef struct rl_mc_params {
    uint32_t spp;               /* scene.nb_samples */
    int32_t stream_mode;        /* rl_stream_mode */
    int32_t seed_variant;
    uint32_t shard_index, shard_count;
    int32_t has_max_distance;   /* ao: Option<f32> */
    float max_distance;
    int32_t normal_correction;  /* ao */
    uint32_t nb_bsdf_samples;   /* direct */
    uint32_t nb_light_samples;  /* direct */
    uint32_t reserved[4];
} rl_mc_params;

/* Integrator::compute for IntegratorAO (src/integrators/ao.rs:9-70). Same output / sharding contract as rl_render_path.  In RL_STREAM_REFERENCE_ORDER both
 * integrators run in two passes like `path` (stats->ms_prepass: the pass that records where every camera sample starts in its block's stream). */
int rl_render_ao(rl_context* ctx, const rl_mc_params* params, const uint64_t* block_seeds, size_t n_blocks,
                 float* out_rgb, int out_is_device, void* stream, rl_render_stats* stats);
/* Integrator::compute for IntegratorDirect (src/integrators/direct.rs:10-233): direct lighting with the power
 * heuristic `mis_weight` (src/integrators/mod.rs:462-478). */
int rl_render_direct(rl_context* ctx, const rl_mc_params* params, const uint64_t* block_seeds, size_t n_blocks,
                     float* out_rgb, int out_is_device, void* stream, rl_render_stats* stats);

/* Batched `Acceleration::trace` (src/accel.rs:292-315): rays (o, d, tnear=1e-4, tfar=MAX).
 * Host pointers.  mesh[i] = -1 on a miss. */
int rl_trace_batch(rl_context* ctx, size_t n, const float* origins, const float* directions,
                   float* t_out, float* u_out, float* v_out, int32_t* mesh_out, int32_t* tri_out);

/* Batched `Acceleration::visible` (src/accel.rs:316-343). visible_out[i] = 1 if unoccluded. */
int rl_visible_batch(rl_context* ctx, size_t n, const float* p0, const float* p1,
                     uint8_t* visible_out);

/* Bitmap::read_pfm (src/structure.rs:563-607): colour PFM, little endian ("-1.0"), rows stored bottom-up and returned
 * top-down, RGB f32.  `rgb == NULL` only reports the size; otherwise `capacity_floats >= 3 * w * h`. */
int rl_load_pfm(const char* path, uint32_t* width, uint32_t* height, float* rgb, size_t capacity_floats);
/* Bitmap::read (src/structure.rs:670-683): by extension — .pfm, .exr (read_exr: R, G, B of a scanline file, NONE / RLE / ZIPS / ZIP,
 * HALF / FLOAT / UINT), .png (8/16-bit), .jpg / .jpeg (baseline and progressive Huffman) or .tga (value / 255 as read_ldr_image does). */
int rl_load_image(const char* path, uint32_t* width, uint32_t* height, float* rgb, size_t capacity_floats);
/* Bitmap::save_pfm (src/structure.rs:547-560): bottom-up rows, |value|, little-endian, "-1.0" scale. */
int rl_save_pfm(const char* path, const float* rgb, uint32_t width, uint32_t height);

/* Bitmap::save_ldr_image (src/structure.rs:471-484) with Color::to_rgba (161-168): 8-bit RGB PNG,
 * (min(c, 1)^(1/2.2) * 255) as u8. */
int rl_save_png(const char* path, const float* rgb, uint32_t width, uint32_t height);
/* Bitmap::save_exr (src/structure.rs:490-527): scanline OpenEXR, FLOAT channels R, G, B, no compression. */
int rl_save_exr(const char* path, const float* rgb, uint32_t width, uint32_t height);
/* Bitmap::save (src/structure.rs:528-545): dispatch on the file extension (.pfm | .png | .exr). */
int rl_save_image(const char* path, const float* rgb, uint32_t width, uint32_t height);

/* Library/build info (for tests: which arch the kernels were compiled for). */
const char* rl_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* RUSTLIGHT_AMD_H */
