// lighttree.cpp — host-side build of the `-x ats` light tree: LightSamplerATS::new over Mesh::convert_light_proxy
// (src/emitter.rs:726-781, 901-973, 1117-1292).  One proxy per emissive triangle (bounding box, orientation cone
// around the face normal with theta_o = 0 / theta_e = pi/2, flux = max channel of Le x area), a binary tree split with
// 12 buckets per axis on the SAOH-style cost `kr * sum(phi * M_omega * area)`, leaves of one light.  The tree is built
// once per scene (untimed, like the BVH) and flattened into 64-byte LightNode records the kernels walk.
//
// Every float below is produced by the operation sequence of the reference; the cone algebra's sin / cos / asin / acos
// come from detmath_shared.h, the same recipe the CPU oracle restates, so both sides build the same tree bit for bit.
#include <algorithm>
#include <cstring>

#include "../detmath_shared.h"
#include "scene.h"

namespace rl {
namespace {

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kHalfPiF = 1.57079632679489661923f;

struct Cone { Vec3 axis{0, 0, 1}; float cos_half = -1.0f; bool empty = false; };   // DirectionCone; default = entire sphere

struct Bounds {   // LightBounds (emitter.rs:901-935)
    Box3 box;
    Vec3 axis{0, 0, 1};
    float phi = 0, theta_o = 0, theta_e = 0, cos_o = 1, cos_e = 1;
    size_t count = 0;
    float phi_sqr = 0;
};
struct Proxy { int32_t emitter, prim; Bounds b; };

float clamped_acos(float v) { return dm::acosf_det(std::fmin(std::fmax(v, -1.0f), 1.0f)); }   // v.max(-1).min(1).acos()
float clamped_asin(float v) { return dm::asinf_det(std::fmin(std::fmax(v, -1.0f), 1.0f)); }

float angle_between(Vec3 a, Vec3 b) {   // emitter.rs:792-798
    if (vdot(a, b) < 0.0f) return kPiF - 2.0f * clamped_asin(vlen(vadd(b, a)) / 2.0f);
    return 2.0f * clamped_asin(vlen(vsub(b, a)) / 2.0f);
}

// rotate(sin, cos, axis).transpose().transform_vector(v) (emitter.rs:800-824): the matrix is filled column by column
// and then transposed, so the product reads its rows; `+ 0 * w` of the homogeneous multiply is kept (apply()).
Vec3 rotate_about(float angle_deg, Vec3 axis, Vec3 v) {
    const float rad = angle_deg * (kPiF / 180.0f);                      // f32::to_radians
    const float s = dm::sinf_det(rad), c = dm::cosf_det(rad);
    const Vec3 a = vnormalize(axis);
    Mat4 m = Mat4::identity();
    m.m[0][0] = a.x * a.x + (1.0f - a.x * a.x) * c; m.m[1][0] = a.x * a.y * (1.0f - c) - a.z * s;   m.m[2][0] = a.x * a.z * (1.0f - c) + a.y * s;
    m.m[0][1] = a.x * a.y * (1.0f - c) + a.z * s;   m.m[1][1] = a.y * a.y + (1.0f - a.y * a.y) * c; m.m[2][1] = a.y * a.z * (1.0f - c) - a.x * s;
    m.m[0][2] = a.x * a.z * (1.0f - c) - a.y * s;   m.m[1][2] = a.y * a.z * (1.0f - c) + a.x * s;   m.m[2][2] = a.z * a.z + (1.0f - a.z * a.z) * c;
    return m.xform_vector(v);
}

Cone merge_cones(const Cone& a, const Cone& b) {   // DirectionCone::union (emitter.rs:848-888)
    if (a.empty) return b;
    if (b.empty) return a;
    const float ta = clamped_acos(a.cos_half), tb = clamped_acos(b.cos_half), td = angle_between(a.axis, b.axis);
    if (std::fmin(td + tb, kPiF) <= ta) return a;
    if (std::fmin(td + ta, kPiF) <= tb) return b;
    const float to = (ta + td + tb) / 2.0f;
    if (to >= kPiF) return Cone();
    const float tr = to - ta;
    const Vec3 wr = vcross(a.axis, b.axis);
    if (vdot(wr, wr) == 0.0f) return Cone();
    Cone c;
    c.axis = rotate_about(tr * 57.2957795130823208767981548141051703f /* f32::to_degrees */, wr, a.axis);
    c.cos_half = dm::cosf_det(to);
    return c;
}

Bounds merge(const Bounds& a, const Bounds& b) {   // LightBounds::union (emitter.rs:947-973)
    if (a.phi == 0.0f) return b;
    if (b.phi == 0.0f) return a;
    Cone ca, cb;
    ca.axis = a.axis; ca.cos_half = a.cos_o;
    cb.axis = b.axis; cb.cos_half = b.cos_o;
    const Cone c = merge_cones(ca, cb);
    Bounds r;
    r.theta_o = clamped_acos(c.cos_half);
    r.theta_e = std::fmax(a.theta_e, b.theta_e);
    r.box = a.box; r.box.grow(b.box);
    r.axis = c.axis;
    r.phi = a.phi + b.phi;
    r.cos_o = dm::cosf_det(r.theta_o);
    r.cos_e = dm::cosf_det(r.theta_e);
    r.count = a.count + b.count;
    r.phi_sqr = a.phi_sqr + b.phi_sqr;
    return r;
}

float cone_measure(const Bounds& b) {   // `momega` (emitter.rs:1183-1191)
    const float tw = std::fmin(b.theta_o + b.theta_e, kPiF);
    return 2.0f * kPiF * (1.0f - dm::cosf_det(b.theta_o))
         + kHalfPiF * (2.0f * tw * dm::sinf_det(b.theta_o) - dm::cosf_det(b.theta_o - 2.0f * tw) - 2.0f * b.theta_o * dm::sinf_det(b.theta_o) + dm::cosf_det(b.theta_o));
}

size_t saturating_usize(float f) { if (!(f > 0.0f)) return 0; if (f >= 1.8446744e19f) return SIZE_MAX; return (size_t)f; }   // `as usize`

struct Builder {
    std::vector<LightNode>* nodes;
    std::vector<Bounds> node_bounds;        // full bounds per node (the device record keeps what importance_point reads)
    std::vector<Proxy>* lights;

    static constexpr size_t kBuckets = 12;
    static size_t bucket(const Box3& centroids, const Proxy& l, int dim) {   // AABB::offset (structure.rs:811-820)
        const Vec3 pc = l.b.box.centre();
        const Vec3 o = vsub(pc, centroids.lo), s = centroids.extent();
        const float rel = s.get(dim) != 0.0f ? o.get(dim) / s.get(dim) : 0.0f;
        return std::min(saturating_usize((float)kBuckets * rel), kBuckets - 1);
    }
    int32_t push(const Bounds& b, int32_t left, int32_t right, int32_t light) {
        LightNode n;
        std::memset(&n, 0, sizeof(n));
        n.bmin[0] = b.box.lo.x; n.bmin[1] = b.box.lo.y; n.bmin[2] = b.box.lo.z;
        n.bmax[0] = b.box.hi.x; n.bmax[1] = b.box.hi.y; n.bmax[2] = b.box.hi.z;
        n.axis[0] = b.axis.x; n.axis[1] = b.axis.y; n.axis[2] = b.axis.z;
        n.phi = b.phi; n.cos_theta_o = b.cos_o; n.cos_theta_e = b.cos_e;
        n.left = left; n.right = right; n.parent = -1; n.light = light;
        nodes->push_back(n);
        node_bounds.push_back(b);
        return (int32_t)nodes->size() - 1;
    }
    // build_bvh (emitter.rs:1117-1262): post-order node numbering, leaves refer to positions in the (reordered) light list
    int32_t build(size_t first, size_t n) {
        Proxy* lt = lights->data() + first;
        if (n == 1) return push(lt[0].b, -1, -1, (int32_t)first);
        Box3 all, centroids;
        for (size_t i = 0; i < n; i++) { all.grow(lt[i].b.box); centroids.grow(lt[i].b.box.centre()); }
        float best = FLT_MAX; int best_bucket = -1, best_dim = -1;
        for (int dim = 0; dim < 3; dim++) {
            if (centroids.hi.get(dim) == centroids.lo.get(dim)) continue;
            Bounds bins[kBuckets];
            for (size_t i = 0; i < n; i++) { const size_t k = bucket(centroids, lt[i], dim); bins[k] = merge(bins[k], lt[i].b); }
            const Vec3 ext = all.extent();
            const float kr = std::fmax(std::fmax(ext.x, ext.y), ext.z) / ext.get(dim);
            for (size_t cut = 0; cut + 1 < kBuckets; cut++) {
                Bounds lo, hi;
                for (size_t j = 0; j <= cut; j++) lo = merge(lo, bins[j]);
                for (size_t j = cut + 1; j < kBuckets; j++) hi = merge(hi, bins[j]);
                const float cost = kr * (lo.phi * cone_measure(lo) * lo.box.half_area() + hi.phi * cone_measure(hi) * hi.box.half_area());
                if (cost > 0.0f && cost < best) { best = cost; best_bucket = (int)cut; best_dim = dim; }
            }
        }
        size_t mid;
        if (best_dim < 0) mid = n / 2;
        else {   // itertools::partition: walk from the front, swap each failing element with the last passing one
            auto keep_left = [&](const Proxy& l) { return bucket(centroids, l, best_dim) <= (size_t)best_bucket; };
            size_t front = 0, back = n;
            mid = 0;
            while (front != back) {
                Proxy& f = lt[front++];
                if (!keep_left(f)) {
                    bool swapped = false;
                    while (back > front) if (keep_left(lt[--back])) { std::swap(f, lt[back]); swapped = true; break; }
                    if (!swapped) break;
                }
                mid++;
            }
        }
        const int32_t left = build(first, mid);
        const int32_t right = build(first + mid, n - mid);
        const int32_t id = push(merge(node_bounds[left], node_bounds[right]), left, right, -1);
        (*nodes)[left].parent = id; (*nodes)[right].parent = id;
        return id;
    }
};

}  // namespace

// Fills scene->ats_* from scene->emitters (all of which must be emissive meshes: `assert!(e.is_surface())`).
int build_light_tree(rl_scene* scene, std::string* err) {
    scene->ats_nodes.clear(); scene->ats_light_emitter.clear(); scene->ats_light_prim.clear();
    scene->ats_leaf_of.clear(); scene->ats_emitter_base.clear();
    scene->ats_root = -1;
    std::vector<Proxy> lights;
    for (size_t e = 0; e < scene->emitters.size(); e++) {
        if (scene->emitters[e].kind != EMITTER_MESH) { *err = "the ATS light tree needs surface emitters only (emitter.rs:1266-1268)"; return RL_ERR_UNSUPPORTED; }
        const HostMesh& m = scene->meshes[scene->emitters[e].mesh];
        scene->ats_emitter_base.push_back((uint32_t)lights.size());
        for (size_t t = 0; t < m.n_tris(); t++) {   // Mesh::convert_light_proxy (emitter.rs:726-781)
            const Vec3 v0 = m.positions[m.indices[3 * t]], v1 = m.positions[m.indices[3 * t + 1]], v2 = m.positions[m.indices[3 * t + 2]];
            // emit(&uv).channel_max() with uv "interpolated at the middle" of the triangle (emitter.rs:741-756); Mesh::emit: geometry.rs:184-206
            float le = std::fmax(m.emission[0], std::fmax(m.emission[1], m.emission[2]));
            if (m.emission_type != RL_EMISSION_COLOR && !m.uvs.empty()) {
                const uint32_t i0 = m.indices[3 * t], i1 = m.indices[3 * t + 1], i2 = m.indices[3 * t + 2];
                const float ux = ((m.uvs[2 * i0] + m.uvs[2 * i1]) + m.uvs[2 * i2]) / 3.0f, uy = ((m.uvs[2 * i0 + 1] + m.uvs[2 * i1 + 1]) + m.uvs[2 * i2 + 1]) / 3.0f;
                float c[3] = {0.0f, 0.0f, 0.0f};
                if (m.emission_type == RL_EMISSION_HSV) {
                    const float x = std::fmod(std::fabs(ux), 1.0f);
                    c[0] = x * 1.0f + (1.0f - x) * 0.0f; c[1] = x * 0.0f + (1.0f - x) * 1.0f; c[2] = x * 0.0f + (1.0f - x) * 0.0f;
                } else if (m.emission_bitmap >= 0 && (size_t)m.emission_bitmap < scene->bitmaps.size()) {
                    const HostBitmap& bm = scene->bitmaps[m.emission_bitmap];          // Bitmap::pixel_uv (structure.rs:434-453)
                    auto mod1 = [](float a) { return std::fmod(std::fmod(a, 1.0f) + 1.0f, 1.0f); };
                    auto as_usize = [](float f) -> unsigned long long { if (!(f > 0.0f)) return 0ull; if (f >= 1.8446744e19f) return ~0ull; return (unsigned long long)f; };
                    const unsigned long long x = as_usize(mod1(ux) * (float)bm.w), y = as_usize(mod1(uy) * (float)bm.h), i = (unsigned long long)bm.w * y + x;
                    if (i < (unsigned long long)bm.w * bm.h) { c[0] = bm.rgb[3 * i]; c[1] = bm.rgb[3 * i + 1]; c[2] = bm.rgb[3 * i + 2]; }
                }
                const float sc_ = m.emission_scale;
                if (std::isfinite(sc_)) { c[0] *= sc_; c[1] *= sc_; c[2] *= sc_; } else { c[0] = c[1] = c[2] = 0.0f; }     // Color * f32 (guarded)
                le = std::fmax(c[0], std::fmax(c[1], c[2]));
            }
            const Vec3 n = vcross(vsub(v1, v0), vsub(v2, v0));
            Proxy p;
            p.emitter = (int32_t)e; p.prim = (int32_t)t;
            p.b.axis = vnormalize(n);
            p.b.theta_o = 0.0f; p.b.theta_e = kHalfPiF;
            p.b.phi = le * vlen(n) * 0.5f;
            p.b.box.grow(v0); p.b.box.grow(v1); p.b.box.grow(v2);
            p.b.cos_o = dm::cosf_det(p.b.theta_o); p.b.cos_e = dm::cosf_det(p.b.theta_e);
            p.b.count = 1; p.b.phi_sqr = p.b.phi * p.b.phi;
            lights.push_back(p);
        }
    }
    if (lights.empty()) return RL_OK;                       // LightSamplerATS::new -> None
    Builder bld;
    bld.nodes = &scene->ats_nodes; bld.lights = &lights;
    scene->ats_root = bld.build(0, lights.size());
    scene->ats_leaf_of.assign(lights.size(), 0);
    for (size_t i = 0; i < scene->ats_nodes.size(); i++) {
        const LightNode& nd = scene->ats_nodes[i];
        if (nd.light < 0) continue;
        const Proxy& p = lights[nd.light];
        scene->ats_leaf_of[scene->ats_emitter_base[p.emitter] + (uint32_t)p.prim] = (uint32_t)i;   // query_to_nodes
    }
    for (const Proxy& p : lights) { scene->ats_light_emitter.push_back(p.emitter); scene->ats_light_prim.push_back(p.prim); }
    return RL_OK;
}

}  // namespace rl
