// device_types.h — POD layouts shared by the host code (C++) and the HIP kernels.
//
// HBM layout of the flattened scene (DESIGN.md §Data layout).  The reference keeps AoS
// `Vec<Vector3<f32>>` per mesh behind `Arc<Mesh>` and 40-byte BVH nodes with separate child
// boxes (src/accel.rs:79-94, src/geometry.rs:107-119); here everything a ray touches is packed
// into 64-byte records so that one lane fetches a whole record with four 16-byte loads.
#pragma once
#include <stdint.h>

namespace rl {

// ---- BVH2 node: both child boxes in the parent (one fetch decides the descent order) ------
// child encoding: >= 0 : index of an inner node;  < 0 : leaf = ~((first_prim << 2) | count), count in {1,2}
//                 RL_CHILD_NONE: empty slot (box never hit)
struct alignas(16) BvhNode {
    float lmin[3]; float lmax0;     // left child  p_min, p_max.x
    float lmax12[2]; float rmin01[2];  // left p_max.yz, right p_min.xy
    float rmin2; float rmax[3];     // right p_min.z, p_max
    int32_t left, right; int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");
static const int32_t RL_CHILD_NONE = (int32_t)0x80000000;

// ---- two-level record of the exact build on scenes that stream their BVH (round 5): node N with the exact f32 boxes of its (up to four) GRANDCHILDREN,
// so that one fetch decides two levels of the reference's recursion (src/accel.rs:256-287) with the identical comparisons in the identical order:
// AABB::intersect (src/structure.rs:849-869) reads the ray and the box only — never its.t — so the slab result of a grandchild box is the same number whether it is
// evaluated on this trip or on the next, and its.t cannot change in between (no triangle is tested between a node and its near child).  A child's own box is the
// exact min / max union of its two children's (range_box folds the same primitive boxes with the same fmin / fmax; the host checks every node, bvh.cpp:
// two_level_nodes), so the record needs no child boxes: the child's slab distances are the min / max of its two slots' per-axis distances (the plane -> distance
// map (p - o) * (1 / d) is monotone for a finite ray; rays with a zero / infinite / NaN component take the union of the PLANES instead: trace.hip.h, traverse2).
// Slots 2c, 2c + 1 belong to child c.  A child that is a leaf — or an inner node whose box failed the host's union check — is NOT expanded: its own box and
// reference sit in slot 2c and slot 2c + 1 is empty (lo = +inf, hi = -inf, RL_CHILD_NONE).  Plane-major: one float4 per plane holds the four slots.
struct alignas(16) BvhNode2 {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    int32_t slot[4];          // child encoding as in BvhNode; inner indices refer to this array (same numbering as `nodes`)
    int32_t child[2];         // the two children themselves (what the traversal stack holds)
    int32_t pad[2];
};
static_assert(sizeof(BvhNode2) == 128, "BvhNode2 must be 128 bytes");

// ---- BVH4 node of the tolerance build (`numerics = fast`, scenes that stream their BVH): the SAME BVH2 collapsed two levels at a time
// (host: build_bvh4) with the child boxes quantised to 8 bits per plane on the node's own grid — conservatively (a quantised box contains the
// child box), so a traversal visits a superset of the leaves the BVH2 visits.  One 64-byte fetch decides four children: about half the node
// trips — and half the vector-memory instructions, which is what bounds the streaming kernel — of the BVH2.
// plane value = org[axis] + q * 2^(exp[axis] - 127);  qlo / qhi are stored [axis][child] so that one dword holds the four children's planes.
struct alignas(16) Bvh4Node {
    float org[3];
    uint32_t exps;            // biased exponents of the x | y << 8 | z << 16 grid steps
    uint32_t qlo[3];          // per axis: child 0..3 lower plane in byte 0..3
    uint32_t qhi[3];
    int32_t child[4];         // child encoding as in BvhNode; inner indices refer to the BVH4 node array; RL_CHILD_NONE = empty slot
    uint32_t pad[2];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node must be 64 bytes");

// ---- triangle record in BVH leaf order: ray-independent terms of Mesh::intersection_tri
//      (src/geometry.rs:358-410) are precomputed once on the host with the reference's f32 ops.
struct alignas(16) TriRecord {
    float v0[3]; float n0;   // v0, n_geo.x
    float e1[3]; float n1;   // e1 = v1 - v0, n_geo.y
    float e2[3]; float n2;   // e2 = v2 - v0, n_geo.z
    float det;               // |e1 x e2|
    int32_t mesh;            // mesh index
    int32_t tri;             // triangle index inside the mesh
    int32_t gtri;            // global triangle index (into tri_indices / shading arrays)
};
static_assert(sizeof(TriRecord) == 64, "TriRecord must be 64 bytes");

// ---- BSDFColor (src/bsdfs/mod.rs:11-29)
struct ColorTex {
    int32_t type;       // rl_tex_type
    float c0[3];
    float c1[3];
    float offset[2];
    float scale[2];
    float line_width;
    int32_t bitmap;     // index into bitmap table
    int32_t pad[3];
};
static_assert(sizeof(ColorTex) == 64, "ColorTex must be 64 bytes");

struct BitmapDesc { uint32_t w, h; uint64_t offset; };  // offset (in float3 texels) into bitmap_texels

// ---- material = one `impl BSDF` object (src/bsdfs/*.rs)
struct Material {
    int32_t type;            // rl_bsdf_type
    int32_t distribution;    // rl_microfacet_type
    float exponent, weight_specular;
    float alpha_u, alpha_v;
    float glass_eta, glass_inv_eta;
    int32_t smooth;          // BSDFType::is_smooth()
    int32_t twosided;        // BSDF::is_twosided()
    int32_t pad[2];
    ColorTex diffuse, specular, transmittance, eta, k;
};

// ---- per-mesh record (struct Mesh, src/geometry.rs:107-119)
struct MeshRecord {
    int32_t material;        // index into materials (one per mesh: `bsdf: Box<dyn BSDF>`)
    int32_t flags;           // MESH_*
    float emission[3];       // EmissionType::Color
    float inv_area;          // Mesh::pdf() = 1 / cdf.total()
    float emitter_pdf;       // EmitterSampler::pdf(self) = emitters_cdf.pdf(i); 0 if not an emitter
    uint32_t vertex_base;    // first vertex in the global vertex arrays
    uint32_t tri_base;       // first triangle in the global index array
    uint32_t n_tris;
    uint32_t cdf_base;       // first entry of this mesh's area cdf (n_tris + 1 entries)
    uint32_t ats_base;       // light tree: first entry of this mesh's triangles in ats_leaf_of (emissive meshes, when the tree is built)
    // EmissionType (src/geometry.rs:99-104): 0 = Color { v = emission }, 1 = HSV { scale }, 2 = Texture { scale, img } — the two uv-dependent kinds of
    // `-x hvs-light` / `-x texture-light` (examples/cli.rs:410-429); Mesh::emit (geometry.rs:184-206)
    int32_t emission_type;
    float emission_scale;
    int32_t emission_bitmap;
};
enum { MESH_HAS_NORMALS = 1, MESH_HAS_UV = 2, MESH_IS_LIGHT = 4 };

// one entry of EmitterSampler::emitters (src/emitter.rs:1491-1496)
enum { EMITTER_MESH = 0, EMITTER_ENV = 1, EMITTER_POINT = 2, EMITTER_DIRECTIONAL = 3 };
struct EmitterRecord {
    int32_t kind;
    int32_t mesh;            // EMITTER_MESH
    float v[3];              // point position / light direction
    float c[3];              // intensity / constant environment luminance
    float center[3];         // bounding sphere after Emitter::preprocess (radius * 1.1)
    float radius;
    float pad[4];
};
static_assert(sizeof(EmitterRecord) == 64, "EmitterRecord must be 64 bytes");

// one node of the `-x ats` light tree: LightBVHNode + the LightBounds fields importance_point reads (emitter.rs:901-935, 1094-1105)
struct LightNode {
    float bmin[3], bmax[3];      // bounds.aabb
    float axis[3];               // bounds.w
    float phi, cos_theta_o, cos_theta_e;
    int32_t left, right, parent; // -1 = none
    int32_t light;               // leaf: index into the light-proxy list; inner: -1
};
static_assert(sizeof(LightNode) == 64, "LightNode must be 64 bytes");

struct CameraRecord {   // struct Camera (src/camera.rs:5-15)
    float sample_to_camera[16];  // column-major
    float to_world[16];
    float position[3];
    uint32_t width, height;
};

struct MediumRecord {   // HomogenousVolume (src/volume.rs:73-80)
    int32_t enabled;
    float sigma_a[3], sigma_s[3], sigma_t[3];
    int32_t phase;
    float g;
};

// Everything the kernels need about the scene (device pointers).
struct DeviceScene {
    const BvhNode* nodes;
    const TriRecord* tris;
    float root_min[3], root_max[3];
    int32_t root;            // child encoding of the root (inner 0, a leaf, or NONE for an empty scene)
    uint32_t n_nodes, n_prims;
    uint32_t stack_depth;    // max number of simultaneously pending far children (+1)
    // streaming scenes: a copy of `nodes` laid out in blocks of 16 (1 KB) that each hold connected pieces of the tree (host: treelet_blocks) — k_stream_chain fetches
    // a whole block with the idle lanes of a chain's group and walks up to four levels without another round trip to memory (null when the scene is staged in LDS)
    const BvhNode* nodes_t;
    int32_t root_t;
    // exact build on streaming scenes: the two-level records (same node numbering as `nodes`; always uploaded: rl_trace_batch reads them on every scene)
    const BvhNode2* nodes2;
    // tolerance build on streaming scenes: the BVH2 above collapsed into quantised BVH4 nodes (null / NONE when the scene is staged in LDS)
    const Bvh4Node* nodes4;
    int32_t root4;
    uint32_t stack_depth4;
    // shading data
    const uint32_t* tri_indices;   // 3 per global triangle, already offset by vertex_base
    const float* positions;        // 3 per vertex
    const float* normals;          // 3 per vertex (zeros where the mesh has none)
    const float* uvs;              // 2 per vertex
    const MeshRecord* meshes;
    const Material* materials;
    const BitmapDesc* bitmaps;
    const float* bitmap_texels;
    // EmitterSampler (non-ATS): emitters in mesh order + cdf over flux.channel_max()
    const EmitterRecord* emitters;
    const float* emitters_cdf;     // n_emitters + 1
    uint32_t n_emitters;
    int32_t env_emitter;           // index of the environment emitter or -1
    float env_color[3];
    float env_pdf;                 // direct_pdf of the constant environment: 1/(4 pi) * p_sel
    // EnvironmentLightColor::Texture (emitter.rs:300-425): lat-long image + Distribution2D over luminance * sin(theta)
    uint32_t env_w, env_h;         // 0 x 0: constant environment
    const float* env_texels;       // 3 per pixel, row-major
    const float* env_cond_cdf;     // env_h rows of (env_w + 1)
    const float* env_cond_func;    // env_h rows of env_w
    const float* env_marg_cdf;     // env_h + 1
    float env_marg_func_int;
    float env_sel_pdf;             // p_sel of the environment emitter (EmitterSampler::pdf)
    // LightSamplerATS (`-x ats`): ats_root < 0 = plain emitter cdf
    int32_t ats_root;
    const LightNode* ats_nodes;
    const int32_t* ats_light_mesh;     // per light proxy: mesh index
    const int32_t* ats_light_prim;     // per light proxy: triangle inside the mesh
    const uint32_t* ats_leaf_of;       // leaf node of (mesh.ats_base + triangle)
    const float* mesh_cdf;         // concatenated per-mesh area cdfs
    uint32_t n_meshes;
    CameraRecord camera;
    MediumRecord medium;
};

}  // namespace rl
