// scene.h — host-side flattened scene: the POD counterpart of rustlight's `Scene` / `Mesh` /
// `Camera` / `EmitterSampler` / `HomogenousVolume` (src/scene.rs:16-30, src/geometry.rs:107-119,
// src/camera.rs:5-15, src/emitter.rs:1491-1496, src/volume.rs:73-80).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rustlight_amd.h"
#include "../device_types.h"
#include "hostmath.h"

namespace rl {

struct HostMesh {
    std::vector<Vec3> positions;
    std::vector<uint32_t> indices;   // 3 per triangle
    std::vector<Vec3> normals;       // empty if None
    std::vector<float> uvs;          // 2 per vertex, empty if None
    rl_bsdf_desc bsdf;
    bool is_light = false;
    float emission[3] = {0, 0, 0};
    int emission_type = 0;           // EmissionType: 0 Color { v = emission }, 1 HSV { scale }, 2 Texture { scale, img } (geometry.rs:99-104)
    float emission_scale = 1.0f;
    int emission_bitmap = -1;
    // Distribution1D over triangle areas (src/math.rs:398-445)
    std::vector<float> cdf;
    float func_int = 0.0f;
    size_t n_tris() const { return indices.size() / 3; }
    float area_total() const { return func_int * (float)(cdf.size() - 1); }   // Distribution1D::total
};

struct HostBitmap { uint32_t w, h; std::vector<float> rgb; };

}  // namespace rl

// The opaque handle of the C-ABI.
struct rl_scene {
    // camera
    bool has_camera = false;
    uint32_t width = 0, height = 0;
    float fov_degrees = 0;
    int fov_axis = 1;
    bool flip = false;
    rl::Mat4 to_world, sample_to_camera;
    rl::Vec3 cam_pos{0, 0, 0};
    // geometry
    std::vector<rl::HostMesh> meshes;
    std::vector<rl::HostBitmap> bitmaps;
    // medium
    rl::MediumRecord medium{};
    // emitters (Scene::build_emitters)
    bool emitters_built = false;
    std::vector<rl::EmitterRecord> other_emitters;   // point / directional lights, insertion order
    bool has_env = false; float env_color[3] = {0, 0, 0};
    // EnvironmentLightColor::Texture: the image and its Distribution2D (built by rl_scene_build_emitters)
    rl::HostBitmap env_map{0, 0, {}};
    std::vector<float> env_cond_cdf, env_cond_func, env_marg_cdf;
    float env_marg_func_int = 0.0f;
    std::vector<rl::EmitterRecord> emitters;          // emissive meshes (mesh order), environment, others
    std::vector<float> emitters_cdf;        // n + 1
    float bsphere_center[3] = {0, 0, 0};
    float bsphere_radius = 0;
    // LightSamplerATS, built by rl_scene_build_emitters when want_ats (Scene::build_emitters(build_ats), `-x ats`)
    bool want_ats = false;
    int32_t ats_root = -1;
    std::vector<rl::LightNode> ats_nodes;
    std::vector<int32_t> ats_light_emitter, ats_light_prim;   // light proxies in tree order
    std::vector<uint32_t> ats_leaf_of;                        // leaf of (ats_emitter_base[emitter] + triangle)
    std::vector<uint32_t> ats_emitter_base;

    bool rebuild_camera();   // Camera::new
};

namespace rl {

// Distribution1DConstruct::normalize (src/math.rs:419-443)
void build_cdf(const std::vector<float>& elements, std::vector<float>* cdf, float* func_int);

// BVH build result in the GPU layout.
struct BvhBuild {
    std::vector<BvhNode> nodes;
    std::vector<TriRecord> tris;
    float root_min[3], root_max[3];
    int32_t root = RL_CHILD_NONE;
    uint32_t stack_depth = 1;
    // the reference-shaped node list, kept for tests (node boxes / info / count, primitive refs)
    std::vector<float> ref_boxes;       // 6 per node
    std::vector<uint64_t> ref_info, ref_count;
    std::vector<int32_t> ref_prim_mesh, ref_prim_tri;
};
// BVHAccel::new (src/accel.rs:115-239): full-sweep SAH over all triangles, leaves of <= 2.
void build_bvh(const rl_scene& scene, BvhBuild* out);
// The inner nodes of `bvh` re-laid in blocks of 16 slots: a block holds one treelet (a connected piece of the tree around its root, chosen largest box first) or
// several small ones; references are renumbered (block * 16 + slot), boxes and leaves untouched — every traversal visits the same nodes in the same order.
void treelet_blocks(const BvhBuild& bvh, std::vector<BvhNode>* nodes_out, int32_t* root_out);
// Two-level records of the exact build (device_types.h: BvhNode2): node i of `bvh.nodes` with its grandchildren's boxes.  A child is expanded only when it is an
// inner node whose stored box equals the fmin / fmax union of its own two children's boxes (always, for boxes made by range_box; checked here so that the
// kernel's shortcut is right by construction).  stats (optional): [0] expanded children, [1] leaf children, [2] inner children left unexpanded by the check.
void two_level_nodes(const BvhBuild& bvh, std::vector<BvhNode2>* out, uint64_t* stats3 = nullptr);
// The tolerance build's BVH4 (device_types.h: Bvh4Node): the BVH2 of `bvh` collapsed — the child with the largest box is replaced by its own two
// children until a node has four (or only leaves are left) — and its child boxes quantised conservatively.  Same leaves, same triangle order.
struct Bvh4Build { std::vector<Bvh4Node> nodes; int32_t root = RL_CHILD_NONE; uint32_t stack_depth = 1; };
void build_bvh4(const BvhBuild& bvh, Bvh4Build* out);

// Flattened shading arrays (global vertex / index / cdf / material tables).
struct FlatScene {
    std::vector<uint32_t> tri_indices;
    std::vector<float> positions, normals, uvs;
    std::vector<MeshRecord> meshes;
    std::vector<Material> materials;
    std::vector<BitmapDesc> bitmaps;
    std::vector<float> bitmap_texels;
    std::vector<float> mesh_cdf;
    std::vector<uint32_t> mesh_tri_base;
};
void flatten_scene(const rl_scene& scene, FlatScene* out);

int build_light_tree(rl_scene* scene, std::string* err);   // lighttree.cpp
int read_pfm(const char* path, uint32_t* w, uint32_t* h, std::vector<float>* rgb);   // Bitmap::read_pfm
int load_pbrt(const char* path, bool use_shading_normals, rl_scene** out, std::string* err);

}  // namespace rl
