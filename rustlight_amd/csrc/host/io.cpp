// io.cpp — image output and host-only debug hooks.
#include <cstdio>
#include <cstring>

#include "../kernels/wavefront.h"
#include "scene.h"

using namespace rl;

extern "C" {

// Bitmap::save_pfm (src/structure.rs:547-560): "PF\nW H\n-1.0\n", rows bottom-up, |r| |g| |b| as LE f32
int rl_save_pfm(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb) return RL_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RL_ERR_IO;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    for (uint32_t y = 0; y < height; y++) {
        const float* row = rgb + (size_t)3 * width * (height - y - 1);
        for (uint32_t x = 0; x < 3 * width; x++) {
            float v = std::fabs(row[x]);
            if (std::fwrite(&v, 4, 1, f) != 1) { std::fclose(f); return RL_ERR_IO; }
        }
    }
    std::fclose(f);
    return RL_OK;
}

int rl_debug_bvh(const rl_scene* scene, uint64_t* n_nodes, uint64_t* n_prims, float* boxes, uint64_t* info, uint64_t* count,
                 int32_t* prim_mesh, int32_t* prim_tri) {
    if (!scene || !n_nodes || !n_prims) return RL_ERR_INVALID_ARGUMENT;
    BvhBuild b;
    build_bvh(*scene, &b);
    *n_nodes = b.ref_info.size();
    *n_prims = b.ref_prim_mesh.size();
    if (boxes) std::memcpy(boxes, b.ref_boxes.data(), b.ref_boxes.size() * sizeof(float));
    if (info) std::memcpy(info, b.ref_info.data(), b.ref_info.size() * sizeof(uint64_t));
    if (count) std::memcpy(count, b.ref_count.data(), b.ref_count.size() * sizeof(uint64_t));
    if (prim_mesh) std::memcpy(prim_mesh, b.ref_prim_mesh.data(), b.ref_prim_mesh.size() * sizeof(int32_t));
    if (prim_tri) std::memcpy(prim_tri, b.ref_prim_tri.data(), b.ref_prim_tri.size() * sizeof(int32_t));
    return RL_OK;
}

// Camera::generate on the host (src/camera.rs:81-91) — same arithmetic as k_raygen
int rl_debug_camera_ray(const rl_scene* scene, float px, float py, float* origin, float* direction) {
    if (!scene || !scene->has_camera) return RL_ERR_INVALID_ARGUMENT;
    Vec3 near_p = scene->sample_to_camera.xform_point({px / (float)scene->width, py / (float)scene->height, 0.0f});
    Vec3 d = vnormalize(near_p);
    Vec3 w = scene->to_world.xform_vector(d);
    origin[0] = scene->cam_pos.x; origin[1] = scene->cam_pos.y; origin[2] = scene->cam_pos.z;
    direction[0] = w.x; direction[1] = w.y; direction[2] = w.z;
    return RL_OK;
}

}  // extern "C"
