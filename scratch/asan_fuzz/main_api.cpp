// Dev: adversarial meshes (NaN / inf vertices, zero-area and coincident triangles, 1e30 / 1e-30 coordinates, an out-of-range index) through
// rl_scene_add_mesh, rl_scene_build_emitters (with and without the light tree) and build_bvh under ASAN / UBSAN; built like main_scene.cpp.
// Round 1: 150 seeds, no finding, no hang.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../include/rustlight_amd.h"
#include "../../rustlight_amd/csrc/host/scene.h"
int main(int argc, char** argv) {
    unsigned seed = argc > 1 ? std::atoi(argv[1]) : 1;
    std::mt19937 g(seed);
    auto U = [&](float a, float b) { return std::uniform_real_distribution<float>(a, b)(g); };
    rl_scene* s = nullptr; rl_scene_create(&s);
    float tw[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,5,1};
    rl_scene_set_camera(s, 32, 32, 40.f, 0, tw, 0);
    int n_mesh = 1 + g() % 4;
    for (int m = 0; m < n_mesh; m++) {
        int nt = g() % 40, kind = g() % 8;
        std::vector<float> v; std::vector<uint32_t> idx;
        for (int t = 0; t < nt; t++) {
            float c[3] = {U(-2, 2), U(-2, 2), U(-2, 2)};
            for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) {
                float x = c[a] + U(-0.3f, 0.3f);
                if (kind == 1 && g() % 7 == 0) x = NAN;
                if (kind == 2 && g() % 7 == 0) x = INFINITY;
                if (kind == 3) x = c[a];                       // zero-area triangles, coincident centroids
                if (kind == 4) x = (float)(g() % 3);           // many identical centroids / bounds
                if (kind == 5) x *= 1e30f;
                if (kind == 6) x *= 1e-30f;
                v.push_back(x);
            }
            idx.push_back(3 * t); idx.push_back(3 * t + 1); idx.push_back(3 * t + 2);
        }
        if (kind == 7 && !idx.empty()) idx[g() % idx.size()] = 1000000;     // index out of range
        rl_bsdf_desc b{}; b.type = 0; b.diffuse.type = 0; b.diffuse.color0[0] = b.diffuse.color0[1] = b.diffuse.color0[2] = 0.5f;
        float em[3] = {1, 1, 1};
        int rc = rl_scene_add_mesh(s, v.data(), v.size() / 3, idx.data(), idx.size() / 3, nullptr, nullptr, &b, (g() % 2) ? em : nullptr);
        std::printf("add_mesh kind %d tris %d -> %d\n", kind, nt, rc);
    }
    rl_scene_enable_ats(s, g() % 2);
    int rc = rl_scene_build_emitters(s);
    std::printf("build_emitters -> %d\n", rc);
    rl::BvhBuild b;
    rl::build_bvh(*s, &b);
    std::printf("bvh ok\n");
    rl_scene_destroy(s);
    return 0;
}
