// bvh.cpp — host BVH2 construction in the GPU layout (untimed, like `BVHAccel::new` which
// IntegratorType::compute calls before starting its timer: src/integrators/mod.rs:280 vs 324).
//
// Algorithm = the reference's (src/accel.rs:115-239): per node a full sweep of the SAH cost
// n_l * A(left) + n_r * A(right) on the three axes over the primitives sorted by box centre,
// leaves of at most two triangles, median split as fallback.  Rust's `sort_by` with the
// comparator `a < b ? Less : Greater` is a stable sort by key (the stable merge sort only asks
// "is a less than b"), so std::stable_sort reproduces the primitive order — and therefore the
// same closest hit on exact-distance ties — exactly.
//
// Output layout (device_types.h): inner nodes only, each carrying both child boxes; leaves are
// folded into the parent's child slots; triangles are pre-transformed into 64-byte TriRecords.
#include <cmath>
#include <algorithm>
#include <cstring>

#include "scene.h"

namespace rl {
namespace {

struct PrimBox { Box3 box; int32_t mesh; uint32_t tri; };
struct RefNode { Box3 box; size_t info; size_t count; };

struct Builder {
    std::vector<PrimBox> prims;
    std::vector<RefNode> nodes;

    Box3 range_box(size_t first, size_t count) const {
        Box3 b;
        for (size_t i = 0; i < count; i++) b.grow(prims[first + i].box);
        return b;
    }
    void sort_axis(size_t first, size_t count, int axis) {
        std::stable_sort(prims.begin() + first, prims.begin() + first + count,
                         [axis](const PrimBox& a, const PrimBox& b) { return a.box.centre().get(axis) < b.box.centre().get(axis); });
    }
    void subdivide(size_t id) {
        if (nodes[id].count <= 2) return;
        const size_t n = nodes[id].count, first = nodes[id].info;
        nodes[id].count = 0;
        nodes[id].info = nodes.size();
        size_t best_pos = 0;
        float best_cost = INFINITY;
        int best_axis = 3;
        std::vector<float> cost(n - 1, 0.0f);
        for (int axis = 0; axis < 3; axis++) {
            sort_axis(first, n, axis);
            Box3 acc;
            for (size_t k = 0; k + 1 < n; k++) {       // right-to-left sweep
                size_t idx = n - k - 1;
                acc.grow(prims[first + idx].box);
                cost[idx - 1] = acc.half_area() * (float)(k + 1);
            }
            acc = Box3();
            for (size_t k = 0; k + 1 < n; k++) {       // left-to-right sweep
                acc.grow(prims[first + k].box);
                cost[k] += acc.half_area() * (float)(k + 1);
                if (cost[k] < best_cost) { best_cost = cost[k]; best_axis = axis; best_pos = k + 1; }
            }
        }
        if (best_axis < 3) sort_axis(first, n, best_axis);
        size_t split = (best_pos == n || best_pos == 0) ? std::max<size_t>((size_t)((float)n * 0.5f), 1) : best_pos;
        RefNode l{range_box(first, split), first, split};
        RefNode r{range_box(first + split, n - split), first + split, n - split};
        size_t li = nodes.size();
        nodes.push_back(l);
        nodes.push_back(r);
        subdivide(li);
        subdivide(li + 1);
    }
};

}  // namespace

void build_bvh(const rl_scene& scene, BvhBuild* out) {
    *out = BvhBuild();
    Builder b;
    Box3 root;
    std::vector<uint32_t> tri_base;
    uint32_t tb = 0;
    for (size_t m = 0; m < scene.meshes.size(); m++) {
        const HostMesh& mesh = scene.meshes[m];
        tri_base.push_back(tb);
        tb += (uint32_t)mesh.n_tris();
        for (size_t t = 0; t < mesh.n_tris(); t++) {
            Box3 bx;   // Mesh::compute_aabb_tri (src/geometry.rs:423-439)
            for (int k = 0; k < 3; k++) bx.grow(mesh.positions[mesh.indices[3 * t + k]]);
            bx.pad_degenerate(0.0001f);
            b.prims.push_back({bx, (int32_t)m, (uint32_t)t});
            root.grow(bx);
        }
    }
    b.nodes.push_back({root, 0, b.prims.size()});
    b.subdivide(0);

    // reference-shaped dump (tests compare it with the oracle's)
    for (const RefNode& n : b.nodes) {
        out->ref_boxes.insert(out->ref_boxes.end(), {n.box.lo.x, n.box.lo.y, n.box.lo.z, n.box.hi.x, n.box.hi.y, n.box.hi.z});
        out->ref_info.push_back(n.info);
        out->ref_count.push_back(n.count);
    }
    for (const PrimBox& p : b.prims) { out->ref_prim_mesh.push_back(p.mesh); out->ref_prim_tri.push_back((int32_t)p.tri); }

    // triangle records in leaf order
    out->tris.resize(b.prims.size());
    for (size_t i = 0; i < b.prims.size(); i++) {
        const HostMesh& mesh = scene.meshes[b.prims[i].mesh];
        uint32_t t = b.prims[i].tri;
        Vec3 v0 = mesh.positions[mesh.indices[3 * t]], v1 = mesh.positions[mesh.indices[3 * t + 1]], v2 = mesh.positions[mesh.indices[3 * t + 2]];
        Vec3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
        Vec3 c = vcross(e1, e2);
        Vec3 n = vnormalize(c);   // n_geo (geometry.rs:370)
        TriRecord& r = out->tris[i];
        r.v0[0] = v0.x; r.v0[1] = v0.y; r.v0[2] = v0.z;
        r.e1[0] = e1.x; r.e1[1] = e1.y; r.e1[2] = e1.z;
        r.e2[0] = e2.x; r.e2[1] = e2.y; r.e2[2] = e2.z;
        r.n0 = n.x; r.n1 = n.y; r.n2 = n.z;
        r.det = vlen(c);          // det (geometry.rs:381)
        r.mesh = b.prims[i].mesh;
        r.tri = (int32_t)t;
        r.gtri = (int32_t)(tri_base[b.prims[i].mesh] + t);
    }

    out->root_min[0] = root.lo.x; out->root_min[1] = root.lo.y; out->root_min[2] = root.lo.z;
    out->root_max[0] = root.hi.x; out->root_max[1] = root.hi.y; out->root_max[2] = root.hi.z;
    if (b.prims.empty()) { out->root = RL_CHILD_NONE; return; }

    // inner-node numbering in creation order; leaves folded into child slots
    std::vector<int32_t> inner_id(b.nodes.size(), -1);
    int32_t n_inner = 0;
    for (size_t i = 0; i < b.nodes.size(); i++)
        if (b.nodes[i].count == 0) inner_id[i] = n_inner++;
    auto encode = [&](size_t i) -> int32_t {
        const RefNode& n = b.nodes[i];
        if (n.count == 0) return inner_id[i];
        return ~(int32_t)(((uint32_t)n.info << 2) | (uint32_t)n.count);
    };
    out->nodes.resize(n_inner);
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (b.nodes[i].count != 0) continue;
        const RefNode& l = b.nodes[b.nodes[i].info];
        const RefNode& r = b.nodes[b.nodes[i].info + 1];
        BvhNode& d = out->nodes[inner_id[i]];
        std::memset(&d, 0, sizeof(d));
        d.lmin[0] = l.box.lo.x; d.lmin[1] = l.box.lo.y; d.lmin[2] = l.box.lo.z;
        d.lmax0 = l.box.hi.x; d.lmax12[0] = l.box.hi.y; d.lmax12[1] = l.box.hi.z;
        d.rmin01[0] = r.box.lo.x; d.rmin01[1] = r.box.lo.y; d.rmin2 = r.box.lo.z;
        d.rmax[0] = r.box.hi.x; d.rmax[1] = r.box.hi.y; d.rmax[2] = r.box.hi.z;
        d.left = encode(b.nodes[i].info);
        d.right = encode(b.nodes[i].info + 1);
    }
    out->root = encode(0);
    // traversal stack bound: one pending far child per inner level
    std::vector<uint32_t> depth(b.nodes.size(), 0);
    uint32_t max_depth = 0;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        if (b.nodes[i].count != 0) continue;
        depth[b.nodes[i].info] = depth[b.nodes[i].info + 1] = depth[i] + 1;
        max_depth = std::max(max_depth, depth[i] + 1);
    }
    out->stack_depth = max_depth + 1;
}

// ------------------------------------------------------------------------------------------
// BVH4 of the tolerance build
namespace {
struct Child2 { int32_t code; float lo[3], hi[3]; };
static void children_of(const BvhNode& nd, Child2* l, Child2* r) {
    l->code = nd.left; r->code = nd.right;
    l->lo[0] = nd.lmin[0]; l->lo[1] = nd.lmin[1]; l->lo[2] = nd.lmin[2]; l->hi[0] = nd.lmax0; l->hi[1] = nd.lmax12[0]; l->hi[2] = nd.lmax12[1];
    r->lo[0] = nd.rmin01[0]; r->lo[1] = nd.rmin01[1]; r->lo[2] = nd.rmin2; r->hi[0] = nd.rmax[0]; r->hi[1] = nd.rmax[1]; r->hi[2] = nd.rmax[2];
}
static double area_of(const Child2& c) {
    const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
}  // namespace

void two_level_nodes(const BvhBuild& bvh, std::vector<BvhNode2>* out, uint64_t* stats3) {
    out->assign(bvh.nodes.size(), BvhNode2{});
    uint64_t n_exp = 0, n_leaf = 0, n_plain = 0;
    // a == b as boxes: every plane equal as a float (+0 == -0: the sign of a zero plane never reaches a verdict or a distance — entry distances are >= tnear > 0);
    // range_box never produces a NaN plane (fmin / fmax from +-FLT_MAX skip NaN vertices), so a NaN here only means "do not expand"
    auto same = [](const Child2& a, const float lo[3], const float hi[3]) {
        for (int k = 0; k < 3; k++) if (!(a.lo[k] == lo[k]) || !(a.hi[k] == hi[k])) return false;
        return true;
    };
    for (size_t i = 0; i < bvh.nodes.size(); i++) {
        BvhNode2& r = (*out)[i];
        Child2 ch[2];
        children_of(bvh.nodes[i], &ch[0], &ch[1]);
        for (int c = 0; c < 2; c++) {
            r.child[c] = ch[c].code;
            Child2 g[2];
            bool expand = false;
            if (ch[c].code >= 0) {
                children_of(bvh.nodes[ch[c].code], &g[0], &g[1]);
                float lo[3], hi[3];
                for (int k = 0; k < 3; k++) { lo[k] = std::fmin(g[0].lo[k], g[1].lo[k]); hi[k] = std::fmax(g[0].hi[k], g[1].hi[k]); }     // AABB::union_aabb (structure.rs:779-784)
                expand = same(ch[c], lo, hi) && g[0].code != RL_CHILD_NONE && g[1].code != RL_CHILD_NONE;
                if (expand) n_exp++; else n_plain++;
            } else n_leaf++;
            if (!expand) {
                g[0] = ch[c];
                g[1].code = RL_CHILD_NONE;
                for (int k = 0; k < 3; k++) { g[1].lo[k] = INFINITY; g[1].hi[k] = -INFINITY; }      // the identity of the union, never entered
            }
            for (int s = 0; s < 2; s++) {
                const int q = 2 * c + s;
                r.lox[q] = g[s].lo[0]; r.loy[q] = g[s].lo[1]; r.loz[q] = g[s].lo[2];
                r.hix[q] = g[s].hi[0]; r.hiy[q] = g[s].hi[1]; r.hiz[q] = g[s].hi[2];
                r.slot[q] = g[s].code;
            }
        }
    }
    if (stats3) { stats3[0] = n_exp; stats3[1] = n_leaf; stats3[2] = n_plain; }
}

void treelet_blocks(const BvhBuild& bvh, std::vector<BvhNode>* nodes_out, int32_t* root_out) {
    nodes_out->clear();
    *root_out = bvh.root;
    const size_t n = bvh.nodes.size();
    if (bvh.root < 0 || n == 0) return;
    std::vector<int32_t> new_index(n, -1);
    std::vector<int32_t> roots{bvh.root};
    size_t n_slots = 0;                       // slots handed out so far (whole blocks of 16; the last block may be open)
    auto less = [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) { return a.first < b.first || (a.first == b.first && a.second > b.second); };
    while (!roots.empty()) {
        const int32_t r = roots.back();
        roots.pop_back();
        // the treelet around r: largest box first, at most 16 nodes
        std::vector<std::pair<double, int32_t>> heap{{1e300, r}};
        std::vector<int32_t> picked;
        while (!heap.empty() && picked.size() < 16) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const int32_t i = heap.back().second;
            heap.pop_back();
            picked.push_back(i);
            Child2 l, rr;
            children_of(bvh.nodes[i], &l, &rr);
            if (l.code >= 0) { heap.push_back({area_of(l), l.code}); std::push_heap(heap.begin(), heap.end(), less); }
            if (rr.code >= 0) { heap.push_back({area_of(rr), rr.code}); std::push_heap(heap.begin(), heap.end(), less); }
        }
        for (const auto& e : heap) roots.push_back(e.second);        // the frontier's inner children start treelets of their own
        // into the open block if it still has room for the whole treelet, else into a fresh block
        const size_t free_slots = (16 - n_slots % 16) % 16;
        if (picked.size() > free_slots) n_slots = (n_slots + 15) / 16 * 16;
        for (int32_t i : picked) new_index[i] = (int32_t)n_slots++;
    }
    nodes_out->assign((n_slots + 15) / 16 * 16, BvhNode{});
    for (BvhNode& d : *nodes_out) { d.left = RL_CHILD_NONE; d.right = RL_CHILD_NONE; }
    for (size_t i = 0; i < n; i++) {
        if (new_index[i] < 0) continue;          // (unreachable nodes: none in a tree)
        BvhNode d = bvh.nodes[i];
        if (d.left >= 0) d.left = new_index[d.left];
        if (d.right >= 0) d.right = new_index[d.right];
        (*nodes_out)[new_index[i]] = d;
    }
    *root_out = new_index[bvh.root];
}

void build_bvh4(const BvhBuild& bvh, Bvh4Build* out) {
    *out = Bvh4Build();
    out->root = bvh.root;
    if (bvh.root < 0) return;                    // empty scene or a single leaf: the leaf code is the root
    // iterative collapse: work items = (BVH2 inner node, slot of the BVH4 node that stands for it, its depth)
    struct Work { int32_t node2; int32_t node4; uint32_t depth; };
    std::vector<Work> todo;
    out->nodes.emplace_back();
    todo.push_back({bvh.root, 0, 1});
    out->root = 0;
    uint32_t max_depth = 1;
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        max_depth = std::max(max_depth, w.depth);
        Child2 ch[4];
        int n = 2;
        children_of(bvh.nodes[w.node2], &ch[0], &ch[1]);
        while (n < 4) {
            int best = -1;
            double best_area = -1.0;
            for (int k = 0; k < n; k++)
                if (ch[k].code >= 0) { const double a = area_of(ch[k]); if (a > best_area) { best_area = a; best = k; } }
            if (best < 0) break;
            Child2 l, r;
            children_of(bvh.nodes[ch[best].code], &l, &r);
            for (int k = n; k > best + 1; k--) ch[k] = ch[k - 1];     // the two grandchildren take the child's place, in order
            ch[best] = l; ch[best + 1] = r;
            n++;
        }
        // the node's grid: origin = lower corner of the union, step = smallest power of two with 255 steps covering the extent
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < n; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], ch[k].lo[a]); hi[a] = std::max(hi[a], ch[k].hi[a]); }
        Bvh4Node nd;
        std::memset(&nd, 0, sizeof(nd));
        uint32_t exps = 0;
        double step[3];
        for (int a = 0; a < 3; a++) {
            if (!(lo[a] <= hi[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) { lo[a] = -3.0e38f; hi[a] = 3.0e38f; }   // non-finite geometry: a box that is always entered
            int e = 0;
            // the child planes are widened by a whole grid step on each side (below), and the grid starts two steps below the box: the device evaluates a
            // plane as (org - ray origin) / d + q * step / d in f32, whose rounding grows with the distance of the ray origin — a fraction of a step of
            // slack is not conservative for a small node seen from far away, a whole step is for origins up to ~2^20 steps off
            const double ext = ((double)hi[a] - (double)lo[a]) * (1.0 + 1e-6) / 250.0;
            if (ext > 0.0) { (void)std::frexp(ext, &e); } else e = -125;                    // ext <= 2^e
            e = std::max(-125, std::min(127, e));
            step[a] = std::ldexp(1.0, e);
            const double o2 = (double)lo[a] - 2.0 * step[a];
            nd.org[a] = std::isfinite(o2) && (double)(float)o2 <= (double)lo[a] - step[a] ? (float)o2 : lo[a];
            exps |= (uint32_t)(e + 127) << (8 * a);
        }
        nd.exps = exps;
        for (int k = 0; k < 4; k++) {
            uint32_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};      // empty slot: an inverted box, never entered
            nd.child[k] = RL_CHILD_NONE;
            if (k < n) {
                for (int a = 0; a < 3; a++) {
                    double l = std::floor(((double)ch[k].lo[a] - (double)nd.org[a]) / step[a] - 1e-3) - 1.0;
                    double h = std::ceil(((double)ch[k].hi[a] - (double)nd.org[a]) / step[a] + 1e-3) + 1.0;
                    if (!(l == l)) l = 0.0;
                    if (!(h == h)) h = 255.0;
                    ql[a] = (uint32_t)std::max(0.0, std::min(255.0, l));
                    qh[a] = (uint32_t)std::max(0.0, std::min(255.0, h));
                }
                if (ch[k].code >= 0) {
                    nd.child[k] = (int32_t)out->nodes.size();
                    out->nodes.emplace_back();
                    todo.push_back({ch[k].code, nd.child[k], w.depth + 1});
                } else nd.child[k] = ch[k].code;
            }
            for (int a = 0; a < 3; a++) { nd.qlo[a] |= ql[a] << (8 * k); nd.qhi[a] |= qh[a] << (8 * k); }
        }
        out->nodes[w.node4] = nd;
    }
    out->stack_depth = 3 * max_depth + 1;       // at most three pending siblings per level
}

}  // namespace rl

// ------------------------------------------------------------------------------------------
// Host-only check of the two structures derived from the BVH2 (test hook, no GPU): the treelet-blocked copy must be the same tree (same boxes, same
// leaves, every inner node in exactly one slot), and every BVH4 node must cover the same set of BVH2 subtrees with boxes that contain the originals.
// out[0] = BVH2 inner nodes, [1] = slots of the blocked copy, [2] = BVH4 nodes, [3] = violations found, [4] = BVH4 leaves, [5] = BVH2 leaves.
// Host-only check of the two-level records (test hook, no GPU): walking them two levels at a time must meet the very nodes, boxes and leaves of the BVH2, every
// expanded child's box must be the union of its slots.  out[0] = BVH2 inner nodes, [1] = expanded children, [2] = leaf children, [3] = inner children left
// unexpanded, [4] = violations, [5] = leaves reached through the records.
extern "C" int rl_debug_check_two_level(const rl_scene* scene, uint64_t* out6) {
    using namespace rl;
    if (!scene || !out6) return -1;
    BvhBuild bvh;
    build_bvh(*scene, &bvh);
    std::vector<BvhNode2> n2;
    uint64_t st[3] = {0, 0, 0};
    two_level_nodes(bvh, &n2, st);
    uint64_t bad = 0, leaves = 0;
    if (n2.size() != bvh.nodes.size()) bad++;
    std::vector<char> seen(bvh.nodes.size(), 0);
    std::vector<int32_t> todo;
    if (bvh.root >= 0) todo.push_back(bvh.root);
    while (!todo.empty() && !bad) {
        const int32_t i = todo.back();
        todo.pop_back();
        if (i < 0 || (size_t)i >= n2.size() || seen[i]) { bad++; break; }
        seen[i] = 1;
        const BvhNode2& r = n2[i];
        Child2 ch[2];
        children_of(bvh.nodes[i], &ch[0], &ch[1]);
        for (int c = 0; c < 2; c++) {
            if (r.child[c] != ch[c].code) bad++;
            const int a = 2 * c, b = a + 1;
            const float lo[3] = {std::fmin(r.lox[a], r.lox[b]), std::fmin(r.loy[a], r.loy[b]), std::fmin(r.loz[a], r.loz[b])};
            const float hi[3] = {std::fmax(r.hix[a], r.hix[b]), std::fmax(r.hiy[a], r.hiy[b]), std::fmax(r.hiz[a], r.hiz[b])};
            for (int k = 0; k < 3; k++) if (!(lo[k] == ch[c].lo[k]) || !(hi[k] == ch[c].hi[k])) bad++;        // the union of the slots IS the child's box
            if (r.slot[b] == RL_CHILD_NONE) {         // not expanded: the child itself
                if (r.slot[a] != ch[c].code) bad++;
                if (ch[c].code >= 0) todo.push_back(ch[c].code); else if (ch[c].code != RL_CHILD_NONE) leaves++;
            } else {
                if (ch[c].code < 0) { bad++; continue; }
                Child2 g[2];
                children_of(bvh.nodes[ch[c].code], &g[0], &g[1]);
                for (int s = 0; s < 2; s++) {
                    const int q = a + s;
                    if (r.slot[q] != g[s].code) bad++;
                    const float bl[3] = {r.lox[q], r.loy[q], r.loz[q]}, bh[3] = {r.hix[q], r.hiy[q], r.hiz[q]};
                    if (std::memcmp(bl, g[s].lo, 12) != 0 || std::memcmp(bh, g[s].hi, 12) != 0) bad++;
                }
                todo.push_back(ch[c].code);                // (its own record is what a traversal fetches when the child is popped from the stack)
            }
        }
    }
    for (size_t i = 0; i < seen.size(); i++) if (bvh.root >= 0 && !seen[i]) bad++;
    out6[0] = bvh.nodes.size(); out6[1] = st[0]; out6[2] = st[1]; out6[3] = st[2]; out6[4] = bad; out6[5] = leaves;
    return 0;
}

extern "C" int rl_debug_check_derived_bvhs(const rl_scene* scene, uint64_t* out6) {
    using namespace rl;
    if (!scene || !out6) return -1;
    BvhBuild bvh;
    build_bvh(*scene, &bvh);
    std::vector<BvhNode> blocks;
    int32_t root_t = RL_CHILD_NONE;
    treelet_blocks(bvh, &blocks, &root_t);
    Bvh4Build b4;
    build_bvh4(bvh, &b4);
    uint64_t bad = 0, leaves2 = 0, leaves4 = 0;
    // ---- blocked copy: walk both trees in step
    if (bvh.root >= 0) {
        std::vector<char> seen(blocks.size(), 0);
        std::vector<std::pair<int32_t, int32_t>> todo{{bvh.root, root_t}};
        while (!todo.empty()) {
            const auto [a, b] = todo.back();
            todo.pop_back();
            if (b < 0 || (size_t)b >= blocks.size() || seen[b]) { bad++; continue; }
            seen[b] = 1;
            const BvhNode &x = bvh.nodes[a], &y = blocks[b];
            if (std::memcmp(&x, &y, 48) != 0) bad++;                      // the twelve planes
            const int32_t xa[2] = {x.left, x.right}, ya[2] = {y.left, y.right};
            for (int k = 0; k < 2; k++) {
                if (xa[k] >= 0) { if (ya[k] < 0) bad++; else todo.push_back({xa[k], ya[k]}); }
                else { if (xa[k] != ya[k]) bad++; if (xa[k] != RL_CHILD_NONE) leaves2++; }
            }
        }
        size_t used = 0;
        for (char c : seen) used += c;
        if (used != bvh.nodes.size()) bad++;
    }
    // ---- BVH4: every node stands for a BVH2 node; its children are that node's descendants, boxes conservative
    if (bvh.root >= 0 && b4.root >= 0) {
        std::vector<std::pair<int32_t, int32_t>> todo{{bvh.root, b4.root}};
        while (!todo.empty()) {
            const auto [a, b] = todo.back();
            todo.pop_back();
            const Bvh4Node& nd = b4.nodes[b];
            // the frontier the collapse must have produced: expand `a` exactly as build_bvh4 does
            Child2 ch[4];
            int n = 2;
            children_of(bvh.nodes[a], &ch[0], &ch[1]);
            while (n < 4) {
                int best = -1; double best_area = -1.0;
                for (int k = 0; k < n; k++) if (ch[k].code >= 0) { const double ar = area_of(ch[k]); if (ar > best_area) { best_area = ar; best = k; } }
                if (best < 0) break;
                Child2 l, r;
                children_of(bvh.nodes[ch[best].code], &l, &r);
                for (int k = n; k > best + 1; k--) ch[k] = ch[k - 1];
                ch[best] = l; ch[best + 1] = r; n++;
            }
            for (int k = 0; k < 4; k++) {
                if (k >= n) { if (nd.child[k] != RL_CHILD_NONE) bad++; continue; }
                for (int ax = 0; ax < 3; ax++) {
                    const float step = std::ldexp(1.0f, (int)((nd.exps >> (8 * ax)) & 0xffu) - 127);
                    const float lo = nd.org[ax] + (float)((nd.qlo[ax] >> (8 * k)) & 0xffu) * step, hi = nd.org[ax] + (float)((nd.qhi[ax] >> (8 * k)) & 0xffu) * step;
                    if (std::isfinite(ch[k].lo[ax]) && std::isfinite(ch[k].hi[ax]) && (!(lo <= ch[k].lo[ax]) || !(hi >= ch[k].hi[ax]))) bad++;
                }
                if (ch[k].code >= 0) { if (nd.child[k] < 0) bad++; else todo.push_back({ch[k].code, nd.child[k]}); }
                else { if (nd.child[k] != ch[k].code) bad++; leaves4++; }
            }
        }
    }
    out6[0] = bvh.nodes.size(); out6[1] = blocks.size(); out6[2] = b4.nodes.size(); out6[3] = bad; out6[4] = leaves4; out6[5] = leaves2;
    return 0;
}
