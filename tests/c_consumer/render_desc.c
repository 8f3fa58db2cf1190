/* render_desc.c — a plain C99 consumer of the drop-in boundary: what a Rust `extern "C"` caller does, minus Rust.
 *   rl_scene_create_from_desc (the whole scene as one POD)  ->  rl_context_create (BVHAccel::new + upload)
 *   ->  rl_sampler_seed + rl_generate_block_seeds (generate_img_blocks, src/integrators/mod.rs:357-371)
 *   ->  rl_render_path (Integrator::compute of IntegratorPathTracing, src/integrators/mod.rs:219-233)
 * The scene arrays come from scene_data.h, which tests/test_gpu_parity.py::test_c99_consumer_renders_the_same_image writes from the fixture
 * scene; the image goes to argv[1] as raw little-endian f32 RGB, the counters to stdout.  Built with `gcc -std=c99 -pedantic`:
 * nothing but include/rustlight_amd.h and the shared library. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rustlight_amd.h"
#include "scene_data.h"   /* SCENE_WIDTH, SCENE_HEIGHT, SCENE_SPP, SCENE_SEED, SCENE_STREAM_MODE, scene_meshes[], SCENE_N_MESHES, scene_sample_to_camera[16], scene_to_world[16] (and scene_fov / scene_fov_axis / scene_flip for -DSCENE_CAMERA_FROM_FOV) */

static int fail(const char* what, int rc) {
    fprintf(stderr, "%s failed: %d (%s)\n", what, rc, rl_last_error());
    return 1;
}

int main(int argc, char** argv) {
    rl_scene_desc desc;
    rl_scene* scene = NULL;
    rl_context* ctx = NULL;
    rl_sampler master;
    rl_path_params params;
    rl_render_stats stats;
    size_t n_blocks, n_floats = (size_t)3 * SCENE_WIDTH * SCENE_HEIGHT;
    uint64_t* seeds;
    float* image;
    FILE* f;
    int rc, n_dev = 0;
    if (argc < 2) { fprintf(stderr, "usage: render_desc out.raw\n"); return 2; }
    if (rl_device_count(&n_dev) != RL_OK || n_dev < 1) { fprintf(stderr, "no HIP device: the MI355X path has no CPU fallback\n"); return 3; }

    (void)scene_fov; (void)scene_fov_axis; (void)scene_flip; (void)scene_sample_to_camera;   /* (one of the two camera forms stays unused) */
    memset(&desc, 0, sizeof(desc));
    desc.width = SCENE_WIDTH; desc.height = SCENE_HEIGHT;
#ifdef SCENE_CAMERA_FROM_FOV
    desc.fov_degrees = scene_fov; desc.fov_axis = scene_fov_axis; desc.flip = scene_flip;          /* Camera::new's arguments (what scene files carry) */
#else
    desc.has_camera_matrices = 1;                                                                    /* the two matrices rustlight's Camera holds (camera.rs:5-15) */
    memcpy(desc.sample_to_camera, scene_sample_to_camera, sizeof(desc.sample_to_camera));
#endif
    memcpy(desc.to_world, scene_to_world, sizeof(desc.to_world));
    desc.meshes = scene_meshes; desc.n_meshes = SCENE_N_MESHES;
    if ((rc = rl_scene_create_from_desc(&desc, &scene)) != RL_OK) return fail("rl_scene_create_from_desc", rc);
    if ((rc = rl_context_create(scene, 0, &ctx)) != RL_OK) return fail("rl_context_create", rc);

    n_blocks = rl_block_count(SCENE_WIDTH, SCENE_HEIGHT);
    seeds = (uint64_t*)malloc(n_blocks * sizeof(uint64_t));
    image = (float*)malloc(n_floats * sizeof(float));
    if (!seeds || !image) return 4;
    rl_sampler_seed(&master, SCENE_SEED, 0);                       /* -r independent:SEED */
    if ((rc = rl_generate_block_seeds(&master, SCENE_WIDTH, SCENE_HEIGHT, seeds, n_blocks)) != RL_OK) return fail("rl_generate_block_seeds", rc);

    rl_path_params_default(&params);                               /* the CLI's defaults; stream_mode = RL_STREAM_REFERENCE_ORDER */
    params.spp = SCENE_SPP;
    params.stream_mode = SCENE_STREAM_MODE;
    if ((rc = rl_render_path(ctx, &params, seeds, n_blocks, image, 0, NULL, &stats)) != RL_OK) return fail("rl_render_path", rc);

    {   /* frames in flight behind one call: the same frame twice on two contexts at once must be the same image twice (rl_render_path_frames) */
        rl_context* ctx2 = NULL;
        rl_context* both[2];
        const uint64_t* frame_seeds[2];
        float* frame_out[2];
        rl_render_stats frame_stats[2];
        int k;
        if ((rc = rl_context_create(scene, 0, &ctx2)) != RL_OK) return fail("rl_context_create (2)", rc);
        both[0] = ctx; both[1] = ctx2;
        frame_seeds[0] = seeds; frame_seeds[1] = seeds;
        for (k = 0; k < 2; k++) { frame_out[k] = (float*)malloc(n_floats * sizeof(float)); if (!frame_out[k]) return 4; }
        if ((rc = rl_render_path_frames(both, 2, &params, frame_seeds, n_blocks, 2, frame_out, frame_stats)) != RL_OK) return fail("rl_render_path_frames", rc);
        for (k = 0; k < 2; k++) {
            if (memcmp(frame_out[k], image, n_floats * sizeof(float)) != 0 || frame_stats[k].rng_draws != stats.rng_draws) { fprintf(stderr, "frame %d in flight differs from the plain render\n", k); return 6; }
            free(frame_out[k]);
        }
        rl_context_destroy(ctx2);
    }
    f = fopen(argv[1], "wb");
    if (!f || fwrite(image, sizeof(float), n_floats, f) != n_floats) { fprintf(stderr, "cannot write %s\n", argv[1]); return 5; }
    fclose(f);
    printf("camera_samples %llu vertices %llu extension_rays %llu shadow_rays %llu rng_draws %llu\n", (unsigned long long)stats.camera_samples,
           (unsigned long long)stats.vertices, (unsigned long long)stats.extension_rays, (unsigned long long)stats.shadow_rays, (unsigned long long)stats.rng_draws);
    free(seeds); free(image);
    rl_context_destroy(ctx);
    rl_scene_destroy(scene);
    return 0;
}
