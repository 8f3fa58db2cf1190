// scene.cpp — host-side scene construction (untimed prologue of the render).
#include "scene.h"
#include "../kernels/wavefront.h"
#include "../detmath_shared.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rl {

static const float kEpsilon = 0.0001f;  // constants::EPSILON (src/lib.rs:51)
static const float kPi = 3.14159265358979323846f;

void build_cdf(const std::vector<float>& elements, std::vector<float>* cdf, float* func_int) {
    cdf->clear();
    cdf->reserve(elements.size() + 1);
    float cur = 0.0f;
    const float n = (float)elements.size();
    for (float e : elements) {
        cdf->push_back(cur);
        cur += e / n;
    }
    cdf->push_back(cur);
    if (cur != 0.0f)
        for (float& x : *cdf) x /= cur;
    cdf->back() = 1.0f;
    *func_int = cur;
}

static void convert_color(const rl_color_desc& d, ColorTex* t) {
    std::memset(t, 0, sizeof(*t));
    t->type = d.type;
    for (int i = 0; i < 3; i++) { t->c0[i] = d.color0[i]; t->c1[i] = d.color1[i]; }
    for (int i = 0; i < 2; i++) { t->offset[i] = d.offset[i]; t->scale[i] = d.scale[i]; }
    t->line_width = d.line_width;
    t->bitmap = d.bitmap_id;
}

static Material convert_material(const rl_bsdf_desc& b) {
    Material m;
    std::memset(&m, 0, sizeof(m));
    m.type = b.type;
    m.distribution = b.distribution;
    m.exponent = b.exponent;
    m.weight_specular = b.weight_specular;
    m.alpha_u = b.alpha_u;
    m.alpha_v = b.alpha_v;
    m.glass_eta = b.glass_eta;
    m.glass_inv_eta = 1.0f / b.glass_eta;   // BSDFGlass::eta() (src/bsdfs/glass.rs:44-49)
    // BSDF::bsdf_type().is_smooth() — DELTA or NULL (src/bsdfs/mod.rs:157-161)
    bool delta = false;
    switch (b.type) {
        case RL_BSDF_METAL: delta = b.distribution == RL_MICROFACET_NONE; break;
        case RL_BSDF_GLASS: delta = true; break;
        case RL_BSDF_SUBSTRATE: delta = b.distribution == RL_MICROFACET_NONE; break;  // DELTA | DIFFUSE
        default: break;
    }
    m.smooth = delta ? 1 : 0;
    m.twosided = b.type == RL_BSDF_GLASS ? 0 : 1;
    convert_color(b.diffuse, &m.diffuse);
    convert_color(b.specular, &m.specular);
    convert_color(b.transmittance, &m.transmittance);
    convert_color(b.eta, &m.eta);
    convert_color(b.k, &m.k);
    return m;
}

void flatten_scene(const rl_scene& scene, FlatScene* out) {
    *out = FlatScene();
    uint32_t vbase = 0, tbase = 0;
    for (size_t mi = 0; mi < scene.meshes.size(); mi++) {
        const HostMesh& m = scene.meshes[mi];
        MeshRecord r;
        std::memset(&r, 0, sizeof(r));
        r.material = (int32_t)out->materials.size();
        out->materials.push_back(convert_material(m.bsdf));
        r.flags = (m.normals.empty() ? 0 : MESH_HAS_NORMALS) | (m.uvs.empty() ? 0 : MESH_HAS_UV) | (m.is_light ? MESH_IS_LIGHT : 0);
        for (int i = 0; i < 3; i++) r.emission[i] = m.emission[i];
        r.emission_type = m.emission_type; r.emission_scale = m.emission_scale; r.emission_bitmap = m.emission_bitmap;
        r.inv_area = 1.0f / m.area_total();
        r.emitter_pdf = 0.0f;
        r.vertex_base = vbase;
        r.tri_base = tbase;
        r.n_tris = (uint32_t)m.n_tris();
        r.cdf_base = (uint32_t)out->mesh_cdf.size();
        out->mesh_tri_base.push_back(tbase);
        out->mesh_cdf.insert(out->mesh_cdf.end(), m.cdf.begin(), m.cdf.end());
        for (size_t v = 0; v < m.positions.size(); v++) {
            out->positions.push_back(m.positions[v].x); out->positions.push_back(m.positions[v].y); out->positions.push_back(m.positions[v].z);
            Vec3 n = m.normals.empty() ? Vec3{0, 0, 0} : m.normals[v];
            out->normals.push_back(n.x); out->normals.push_back(n.y); out->normals.push_back(n.z);
            out->uvs.push_back(m.uvs.empty() ? 0.0f : m.uvs[2 * v]);
            out->uvs.push_back(m.uvs.empty() ? 0.0f : m.uvs[2 * v + 1]);
        }
        for (uint32_t i : m.indices) out->tri_indices.push_back(i + vbase);
        vbase += (uint32_t)m.positions.size();
        tbase += (uint32_t)m.n_tris();
        out->meshes.push_back(r);
    }
    // EmitterSampler::pdf(emitter) = emitters_cdf.pdf(i) (src/emitter.rs:1510-1526)
    for (size_t e = 0; e < scene.emitters.size(); e++)
        if (scene.emitters[e].kind == EMITTER_MESH)
            out->meshes[scene.emitters[e].mesh].emitter_pdf = scene.emitters_cdf[e + 1] - scene.emitters_cdf[e];
    uint64_t off = 0;
    for (const HostBitmap& b : scene.bitmaps) {
        out->bitmaps.push_back({b.w, b.h, off});
        out->bitmap_texels.insert(out->bitmap_texels.end(), b.rgb.begin(), b.rgb.end());
        off += (uint64_t)b.w * b.h;
    }
}

}  // namespace rl

using namespace rl;

// Camera::new (src/camera.rs:31-67)
bool rl_scene::rebuild_camera() {
    Mat4 to_local;
    if (!to_world.inverse(&to_local)) return false;
    const float x_v = flip ? 1.0f : -1.0f;
    const float aspect = (float)width / (float)height;
    const float fov_rad = fov_axis == 0 ? fov_degrees * kPi / 180.0f : fov_degrees * aspect * kPi / 180.0f;
    Mat4 camera_to_sample = Mat4::scale(-0.5f, -0.5f * aspect, 1.0f)
                                .times(Mat4::translate(-1.0f, -1.0f / aspect, 0.0f))
                                .times(perspective(fov_rad, 1.0f, 1e-2f, 1000.0f))
                                .times(Mat4::scale(x_v, 1.0f, -1.0f));
    if (!camera_to_sample.inverse(&sample_to_camera)) return false;
    cam_pos = to_world.xform_point({0.0f, 0.0f, 0.0f});   // Camera::position (camera.rs:140-142)
    has_camera = true;
    return true;
}

extern "C" {

int rl_scene_create(rl_scene** out) {
    if (!out) return RL_ERR_INVALID_ARGUMENT;
    *out = new rl_scene();
    return RL_OK;
}
void rl_scene_destroy(rl_scene* scene) { delete scene; }

int rl_scene_set_camera(rl_scene* scene, uint32_t width, uint32_t height, float fov_degrees, int fov_axis,
                        const float to_world[16], int flip) {
    if (!scene || !to_world || width == 0 || height == 0 || (fov_axis != 0 && fov_axis != 1)) return RL_ERR_INVALID_ARGUMENT;
    scene->width = width; scene->height = height;
    scene->fov_degrees = fov_degrees; scene->fov_axis = fov_axis; scene->flip = flip != 0;
    scene->to_world = Mat4::from_cols(to_world);
    return scene->rebuild_camera() ? RL_OK : RL_ERR_INVALID_ARGUMENT;
}

// the camera from the two matrices rustlight's Camera holds (camera.rs:5-15): nothing of Camera::new is re-derived
int rl_scene_set_camera_matrices(rl_scene* scene, uint32_t width, uint32_t height, const float sample_to_camera[16], const float to_world[16]) {
    if (!scene || !sample_to_camera || !to_world || width == 0 || height == 0) return RL_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 16; i++)
        if (!std::isfinite(sample_to_camera[i]) || !std::isfinite(to_world[i])) { rl_set_error("camera matrices must be finite"); return RL_ERR_INVALID_ARGUMENT; }
    scene->width = width; scene->height = height;
    scene->fov_degrees = 0.0f; scene->fov_axis = 0; scene->flip = false;       // (not derived from: the matrices are authoritative)
    scene->to_world = Mat4::from_cols(to_world);
    scene->sample_to_camera = Mat4::from_cols(sample_to_camera);
    scene->cam_pos = scene->to_world.xform_point({0.0f, 0.0f, 0.0f});   // Camera::position (camera.rs:140-142)
    scene->has_camera = true;
    return RL_OK;
}
int rl_scene_get_camera_matrices(const rl_scene* scene, float sample_to_camera[16], float to_world[16], float position[3]) {
    if (!scene || !scene->has_camera || !sample_to_camera || !to_world || !position) return RL_ERR_INVALID_ARGUMENT;
    scene->sample_to_camera.to_cols(sample_to_camera);
    scene->to_world.to_cols(to_world);
    position[0] = scene->cam_pos.x; position[1] = scene->cam_pos.y; position[2] = scene->cam_pos.z;
    return RL_OK;
}

// EmissionType::HSV / Texture on one light mesh (geometry.rs:99-104) — what examples/cli.rs:410-429 assigns under `-x hvs-light` / `-x texture-light`
int rl_scene_set_mesh_emission(rl_scene* scene, uint32_t mesh, int type, float scale, int bitmap_id) {
    if (!scene || mesh >= scene->meshes.size() || type < RL_EMISSION_COLOR || type > RL_EMISSION_TEXTURE) return RL_ERR_INVALID_ARGUMENT;
    HostMesh& m = scene->meshes[mesh];
    if (!m.is_light) { rl_set_error("mesh " + std::to_string(mesh) + " is not a light (EmissionType::Zero)"); return RL_ERR_INVALID_ARGUMENT; }
    if (type != RL_EMISSION_COLOR && m.uvs.empty()) { rl_set_error("HSV / texture emission needs uv coordinates on the light mesh (Mesh::emit unwraps them: geometry.rs:200-203)"); return RL_ERR_INVALID_ARGUMENT; }
    if (type == RL_EMISSION_TEXTURE && (bitmap_id < 0 || (size_t)bitmap_id >= scene->bitmaps.size())) { rl_set_error("texture emission without a valid bitmap id"); return RL_ERR_INVALID_ARGUMENT; }
    m.emission_type = type; m.emission_scale = scale; m.emission_bitmap = type == RL_EMISSION_TEXTURE ? bitmap_id : -1;
    scene->emitters_built = false;
    return RL_OK;
}
// examples/cli.rs:410-429: every light mesh becomes HSV { scale } / Texture { scale, img } with scale = the luminance of its colour (1 if it had none)
int rl_scene_override_light_emission(rl_scene* scene, int type, int bitmap_id) {
    if (!scene || (type != RL_EMISSION_HSV && type != RL_EMISSION_TEXTURE)) return RL_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < scene->meshes.size(); i++) {
        HostMesh& m = scene->meshes[i];
        if (!m.is_light) continue;
        const float scale = m.emission_type == RL_EMISSION_COLOR ? (m.emission[0] * 0.212671f + m.emission[1] * 0.715160f) + m.emission[2] * 0.072169f : 1.0f;   // Color::luminance (structure.rs:173-176)
        const int rc = rl_scene_set_mesh_emission(scene, (uint32_t)i, type, scale, bitmap_id);
        if (rc != RL_OK) return rc;
    }
    return RL_OK;
}

int rl_scene_scale_image(rl_scene* scene, float s) {
    if (!scene || !scene->has_camera) return RL_ERR_INVALID_ARGUMENT;
    // the scale is caller input: a negative, NaN or infinite value must be refused before it reaches the float -> unsigned conversion
    // (undefined behaviour outside the target's range)
    if (!(s > 0.0f) || !std::isfinite(s)) { rl_set_error("image scale must be a positive finite number"); return RL_ERR_INVALID_ARGUMENT; }
    // Camera::scale_image only rescales `img`; the matrices keep the original aspect (camera.rs:73-78)
    const float fw = s * (float)scene->width, fh = s * (float)scene->height;
    if (!(fw >= 1.0f) || !(fh >= 1.0f) || fw >= 4294967040.0f || fh >= 4294967040.0f) { rl_set_error("image scale leaves no pixels (or more than 2^32)"); return RL_ERR_INVALID_ARGUMENT; }
    const uint32_t w = (uint32_t)fw, h = (uint32_t)fh;
    scene->width = w;
    scene->height = h;
    return RL_OK;
}

int rl_scene_add_bitmap(rl_scene* scene, uint32_t w, uint32_t h, const float* rgb) {
    if (!scene || !rgb || w == 0 || h == 0) return RL_ERR_INVALID_ARGUMENT;
    HostBitmap b{w, h, std::vector<float>(rgb, rgb + (size_t)3 * w * h)};
    scene->bitmaps.push_back(std::move(b));
    return (int)scene->bitmaps.size() - 1;
}

// Mesh::new (src/geometry.rs:122-182)
int rl_scene_add_mesh(rl_scene* scene, const float* vertices, size_t n_vertices, const uint32_t* indices,
                      size_t n_triangles, const float* normals, const float* uv, const rl_bsdf_desc* bsdf,
                      const float* emission_rgb) {
    if (!scene || !vertices || !indices || !bsdf) return RL_ERR_INVALID_ARGUMENT;
    if (n_triangles == 0 || n_vertices == 0) return RL_ERR_INVALID_ARGUMENT;   // "Empty meshs": Mesh::new returns None
    for (size_t i = 0; i < 3 * n_triangles; i++)
        if (indices[i] >= n_vertices) return RL_ERR_INVALID_ARGUMENT;
    if (bsdf->type < RL_BSDF_DIFFUSE || bsdf->type > RL_BSDF_SUBSTRATE) { rl_set_error("unknown BSDF type"); return RL_ERR_INVALID_ARGUMENT; }
    if (bsdf->distribution < RL_MICROFACET_NONE || bsdf->distribution > RL_MICROFACET_GGX) { rl_set_error("unknown microfacet distribution"); return RL_ERR_INVALID_ARGUMENT; }
    // the device indexes textures by these ids without a bounds check: every enum / id of the five colour slots is validated here
    // (bitmaps may still be added after the mesh: the upper bound of bitmap_id is checked again when the emitters are built)
    for (const rl_color_desc* c : {&bsdf->diffuse, &bsdf->specular, &bsdf->transmittance, &bsdf->eta, &bsdf->k}) {
        if (c->type < RL_TEX_CONSTANT || c->type > RL_TEX_BITMAP) { rl_set_error("unknown texture type in a BSDF colour"); return RL_ERR_INVALID_ARGUMENT; }
        if (c->type == RL_TEX_BITMAP && c->bitmap_id < 0) { rl_set_error("RL_TEX_BITMAP colour without a bitmap id"); return RL_ERR_INVALID_ARGUMENT; }
    }
    HostMesh m;
    m.positions.resize(n_vertices);
    for (size_t i = 0; i < n_vertices; i++) m.positions[i] = {vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]};
    m.indices.assign(indices, indices + 3 * n_triangles);
    std::vector<float> areas(n_triangles);
    for (size_t t = 0; t < n_triangles; t++) {
        Vec3 v0 = m.positions[indices[3 * t]], v1 = m.positions[indices[3 * t + 1]], v2 = m.positions[indices[3 * t + 2]];
        areas[t] = vlen(vcross(vsub(v1, v0), vsub(v2, v0))) * 0.5f;
    }
    if (normals) {
        m.normals.resize(n_vertices);
        size_t wrong = 0;
        for (size_t i = 0; i < n_vertices; i++) {
            Vec3 n = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
            float l = vdot(n, n);
            if (l == 0.0f) wrong++;
            else if (l != 1.0f) n = vdiv(n, std::sqrt(l));
            m.normals[i] = n;
        }
        if (wrong > 0 && wrong == n_vertices) m.normals.clear();   // "All normal are wrong": normals = None
    }
    if (uv) m.uvs.assign(uv, uv + 2 * n_vertices);
    m.bsdf = *bsdf;
    if (emission_rgb) {
        m.is_light = true;
        for (int i = 0; i < 3; i++) m.emission[i] = emission_rgb[i];
    }
    build_cdf(areas, &m.cdf, &m.func_int);
    scene->meshes.push_back(std::move(m));
    scene->emitters_built = false;
    return (int)scene->meshes.size() - 1;
}

int rl_scene_set_medium(rl_scene* scene, const float sigma_a[3], const float sigma_s[3], int phase_type, float g) {
    if (!scene || !sigma_a || !sigma_s || (phase_type != RL_PHASE_ISOTROPIC && phase_type != RL_PHASE_HG)) return RL_ERR_INVALID_ARGUMENT;
    MediumRecord& m = scene->medium;
    m.enabled = 1;
    for (int i = 0; i < 3; i++) {
        m.sigma_a[i] = sigma_a[i];
        m.sigma_s[i] = sigma_s[i];
        float t = sigma_a[i] + sigma_s[i];
        m.sigma_t[i] = t * 1.0f;   // (sigma_a + sigma_s) * density_mult, density_mult = 1.0 (cli.rs:381-385)
    }
    m.phase = phase_type;
    m.g = g;
    return RL_OK;
}

// Scene::build_emitters(false) (src/scene.rs:53-123)
int rl_scene_build_emitters(rl_scene* scene) {
    if (!scene || !scene->has_camera) return RL_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < scene->meshes.size(); i++) {
        const rl_bsdf_desc& b = scene->meshes[i].bsdf;
        for (const rl_color_desc* c : {&b.diffuse, &b.specular, &b.transmittance, &b.eta, &b.k})
            if (c->type == RL_TEX_BITMAP && (c->bitmap_id < 0 || (size_t)c->bitmap_id >= scene->bitmaps.size())) {
                rl_set_error("mesh " + std::to_string(i) + " references bitmap " + std::to_string(c->bitmap_id) + " but the scene has " +
                             std::to_string(scene->bitmaps.size()) + " bitmaps");
                return RL_ERR_INVALID_ARGUMENT;
            }
    }
    Box3 box;
    for (const HostMesh& m : scene->meshes) {
        Box3 b;
        for (const Vec3& p : m.positions) b.grow(p);
        b.pad_degenerate(kEpsilon);
        box.grow(b);
    }
    box.grow(scene->cam_pos);
    Vec3 c = box.centre();
    scene->bsphere_center[0] = c.x; scene->bsphere_center[1] = c.y; scene->bsphere_center[2] = c.z;
    scene->bsphere_radius = vlen(vsub(c, box.hi));
    scene->emitters.clear();
    std::vector<float> flux;
    auto channel_max = [](const float* c) { return std::fmax(c[0], std::fmax(c[1], c[2])); };
    for (size_t i = 0; i < scene->meshes.size(); i++) {
        const HostMesh& m = scene->meshes[i];
        if (!m.is_light) continue;
        EmitterRecord e;
        std::memset(&e, 0, sizeof(e));
        e.kind = EMITTER_MESH; e.mesh = (int32_t)i;
        scene->emitters.push_back(e);
        // Mesh::flux = cdf.total() * e * PI (emitter.rs:591-599)
        float total = m.area_total();
        // a NaN / infinite total leaves NaNs in the normalised area cdf (a zero total does not: Distribution1DConstruct::normalize skips
        // the division, math.rs:417-441): the reference panics on them in sample_discrete (`partial_cmp(..).unwrap()`, math.rs:447-457)
        // — here the scene is refused instead of rendered with garbage indices
        if (!std::isfinite(total)) {
            rl_set_error("emissive mesh " + std::to_string(i) + " has no finite area (NaN / infinite vertices)");
            return RL_ERR_INVALID_ARGUMENT;
        }
        float ch[3];
        // Emitter::flux (emitter.rs:591-599): the constant colour, or Color::value(scale) for the uv-dependent kinds ("TODO" there)
        for (int k = 0; k < 3; k++) ch[k] = ((m.emission_type == 0 ? m.emission[k] : m.emission_scale) * total) * kPi;
        flux.push_back(channel_max(ch));
    }
    const float big_radius = scene->bsphere_radius * 1.1f;      // Emitter::preprocess: bsphere.radius *= 1.1
    auto finish = [&](EmitterRecord e) {
        for (int k = 0; k < 3; k++) e.center[k] = scene->bsphere_center[k];
        e.radius = big_radius;
        scene->emitters.push_back(e);
        float ch[3];
        float r2 = big_radius * big_radius;                      // radius.powi(2)
        if (e.kind == EMITTER_POINT) for (int k = 0; k < 3; k++) ch[k] = (e.c[k] * 4.0f) * kPi;            // intensity * 4 * PI
        else if (e.kind == EMITTER_DIRECTIONAL) for (int k = 0; k < 3; k++) ch[k] = e.c[k] * (kPi * r2);    // area * intensity
        else for (int k = 0; k < 3; k++) ch[k] = e.c[k] * (kPi * r2);                                      // PI * r^2 * c
        flux.push_back(channel_max(ch));
    };
    if (scene->has_env) {
        EmitterRecord e;
        std::memset(&e, 0, sizeof(e));
        e.kind = EMITTER_ENV; e.mesh = -1;
        for (int k = 0; k < 3; k++) e.c[k] = scene->env_color[k];
        if (scene->env_map.w) {
            // EnvironmentLightColor::new_texture + Distribution2D::from_bitmap (emitter.rs:340-353, math.rs:494-521):
            // rows weighted by sin(theta) at the pixel centre, one conditional cdf per row, marginal over the row means
            const HostBitmap& im = scene->env_map;
            scene->env_cond_cdf.clear(); scene->env_cond_func.clear();
            std::vector<float> marg, row(im.w), cdf;
            for (uint32_t y = 0; y < im.h; y++) {
                const float w = dm::sinf_det(((float)y + 0.5f) * kPi / (float)im.h);
                for (uint32_t x = 0; x < im.w; x++) {
                    const float* px = &im.rgb[3 * ((size_t)y * im.w + x)];
                    const float r = px[0] * w, g = px[1] * w, b = px[2] * w;
                    row[x] = (r * 0.212671f + g * 0.715160f) + b * 0.072169f;      // Color::luminance (structure.rs:173-176)
                }
                float fi;
                build_cdf(row, &cdf, &fi);
                scene->env_cond_cdf.insert(scene->env_cond_cdf.end(), cdf.begin(), cdf.end());
                scene->env_cond_func.insert(scene->env_cond_func.end(), row.begin(), row.end());
                marg.push_back(fi);
            }
            build_cdf(marg, &scene->env_marg_cdf, &scene->env_marg_func_int);
            // flux = Color::value(PI * radius^2 * marginal.func_int) (emitter.rs:524-530); finish() multiplies c by PI * r^2
            for (int k = 0; k < 3; k++) e.c[k] = 0.0f;
            for (int k = 0; k < 3; k++) e.center[k] = scene->bsphere_center[k];
            e.radius = big_radius;
            scene->emitters.push_back(e);
            flux.push_back((kPi * (big_radius * big_radius)) * scene->env_marg_func_int);
        } else finish(e);
    }
    for (const EmitterRecord& o : scene->other_emitters) finish(o);
    scene->emitters_cdf.clear();
    if (!flux.empty()) {
        float sum = 0.0f;
        for (float f : flux) sum += f;
        if (!std::isfinite(sum)) {      // same reason: the emitter-selection cdf would hold NaNs
            rl_set_error("the emitters' total flux is not finite (NaN / infinite emission, or non-finite vertices inflating the scene's bounding sphere)");
            return RL_ERR_INVALID_ARGUMENT;
        }
        float fi;
        build_cdf(flux, &scene->emitters_cdf, &fi);
    }
    scene->ats_root = -1; scene->ats_nodes.clear();
    if (scene->want_ats) {   // emitter_sampler.build_ats() (scene.rs:118-120)
        std::string err;
        int rc = build_light_tree(scene, &err);
        if (rc != RL_OK) { rl_set_error(err); return rc; }
    }
    scene->emitters_built = true;
    return RL_OK;
}

int rl_scene_enable_ats(rl_scene* scene, int build_ats) {
    if (!scene) return RL_ERR_INVALID_ARGUMENT;
    scene->want_ats = build_ats != 0;
    scene->emitters_built = false;
    return RL_OK;
}

int rl_scene_add_point_light(rl_scene* scene, const float position[3], const float intensity[3]) {
    if (!scene || !position || !intensity) return RL_ERR_INVALID_ARGUMENT;
    EmitterRecord e;
    std::memset(&e, 0, sizeof(e));
    e.kind = EMITTER_POINT; e.mesh = -1;
    for (int k = 0; k < 3; k++) { e.v[k] = position[k]; e.c[k] = intensity[k]; }
    scene->other_emitters.push_back(e);
    scene->emitters_built = false;
    return RL_OK;
}
int rl_scene_add_directional_light(rl_scene* scene, const float direction[3], const float intensity[3]) {
    if (!scene || !direction || !intensity) return RL_ERR_INVALID_ARGUMENT;
    EmitterRecord e;
    std::memset(&e, 0, sizeof(e));
    e.kind = EMITTER_DIRECTIONAL; e.mesh = -1;
    for (int k = 0; k < 3; k++) { e.v[k] = direction[k]; e.c[k] = intensity[k]; }
    scene->other_emitters.push_back(e);
    scene->emitters_built = false;
    return RL_OK;
}
int rl_scene_set_environment(rl_scene* scene, const float rgb[3]) {
    if (!scene || !rgb) return RL_ERR_INVALID_ARGUMENT;
    scene->has_env = true;
    for (int k = 0; k < 3; k++) scene->env_color[k] = rgb[k];
    scene->emitters_built = false;
    return RL_OK;
}

int rl_scene_set_environment_map(rl_scene* scene, uint32_t w, uint32_t h, const float* rgb) {
    if (!scene || !rgb || w == 0 || h == 0) return RL_ERR_INVALID_ARGUMENT;
    scene->has_env = true;
    scene->env_map.w = w; scene->env_map.h = h;
    scene->env_map.rgb.assign(rgb, rgb + (size_t)3 * w * h);
    scene->emitters_built = false;
    return RL_OK;
}

int rl_scene_image_size(const rl_scene* scene, uint32_t* w, uint32_t* h) {
    if (!scene || !w || !h) return RL_ERR_INVALID_ARGUMENT;
    *w = scene->width; *h = scene->height;
    return RL_OK;
}

int rl_scene_counts(const rl_scene* scene, uint64_t* n_meshes, uint64_t* n_triangles, uint64_t* n_emitters) {
    if (!scene) return RL_ERR_INVALID_ARGUMENT;
    uint64_t t = 0;
    for (const HostMesh& m : scene->meshes) t += m.n_tris();
    if (n_meshes) *n_meshes = scene->meshes.size();
    if (n_triangles) *n_triangles = t;
    if (n_emitters) *n_emitters = scene->emitters.size();
    return RL_OK;
}

// ---- sampler (src/samplers/independent.rs)
void rl_sampler_seed(rl_sampler* sampler, uint64_t seed, int variant) {
    Xoshiro x; x.seed(seed, variant);
    std::memcpy(sampler->s, x.s, sizeof(x.s));
}
uint64_t rl_sampler_next_u64(rl_sampler* sampler) {
    Xoshiro x; std::memcpy(x.s, sampler->s, sizeof(x.s));
    uint64_t v = x.next();
    std::memcpy(sampler->s, x.s, sizeof(x.s));
    return v;
}
float rl_sampler_next_f32(rl_sampler* sampler) {
    Xoshiro x; std::memcpy(x.s, sampler->s, sizeof(x.s));
    float v = x.next_f32();
    std::memcpy(sampler->s, x.s, sizeof(x.s));
    return v;
}

// ---- generate_img_blocks (src/integrators/mod.rs:351-374)
size_t rl_block_count(uint32_t width, uint32_t height) { return (size_t)((width + 15) / 16) * ((height + 15) / 16); }

int rl_generate_block_seeds(rl_sampler* master, uint32_t width, uint32_t height, uint64_t* seeds_out, size_t n_blocks) {
    if (!master || !seeds_out || n_blocks != rl_block_count(width, height)) return RL_ERR_INVALID_ARGUMENT;
    size_t k = 0;
    for (uint32_t ix = 0; ix < width; ix += 16)          // x-major creation order
        for (uint32_t iy = 0; iy < height; iy += 16)
            seeds_out[k++] = rl_sampler_next_u64(master);   // clone_box: seed_from_u64(self.rnd.next_u64())
    return RL_OK;
}

// the scene as one POD — the builder calls above in their canonical order
int rl_scene_create_from_desc(const rl_scene_desc* d, rl_scene** out) {
    if (!d || !out) return RL_ERR_INVALID_ARGUMENT;
    if ((d->n_meshes && !d->meshes) || (d->n_bitmaps && !d->bitmaps) || (d->n_lights && !d->lights)) return RL_ERR_INVALID_ARGUMENT;
    rl_scene* s = nullptr;
    int rc = rl_scene_create(&s);
    if (rc != RL_OK) return rc;
    auto fail = [&](int code) { rl_scene_destroy(s); return code < 0 ? code : RL_ERR_INVALID_ARGUMENT; };
    if ((rc = d->has_camera_matrices ? rl_scene_set_camera_matrices(s, d->width, d->height, d->sample_to_camera, d->to_world)
                                     : rl_scene_set_camera(s, d->width, d->height, d->fov_degrees, d->fov_axis, d->to_world, d->flip)) != RL_OK) return fail(rc);
    for (size_t i = 0; i < d->n_bitmaps; i++)
        if ((rc = rl_scene_add_bitmap(s, d->bitmaps[i].width, d->bitmaps[i].height, d->bitmaps[i].rgb)) < 0) return fail(rc);
    for (size_t i = 0; i < d->n_meshes; i++) {
        const rl_mesh_desc& m = d->meshes[i];
        if ((rc = rl_scene_add_mesh(s, m.vertices, m.n_vertices, m.indices, m.n_triangles, m.normals, m.uv, &m.bsdf, m.has_emission ? m.emission_rgb : nullptr)) < 0) return fail(rc);
        if (m.emission_type != RL_EMISSION_COLOR && (rc = rl_scene_set_mesh_emission(s, (uint32_t)i, m.emission_type, m.emission_scale, m.emission_bitmap_id)) != RL_OK) return fail(rc);
    }
    if (d->has_medium && (rc = rl_scene_set_medium(s, d->sigma_a, d->sigma_s, d->phase_type, d->g)) != RL_OK) return fail(rc);
    for (size_t i = 0; i < d->n_lights; i++) {
        const rl_light_desc& l = d->lights[i];
        if (l.kind != 0 && l.kind != 1) { rl_set_error("unknown light kind"); return fail(RL_ERR_INVALID_ARGUMENT); }
        if ((rc = (l.kind == 0 ? rl_scene_add_point_light : rl_scene_add_directional_light)(s, l.a, l.intensity)) != RL_OK) return fail(rc);
    }
    if (d->has_environment && (rc = rl_scene_set_environment(s, d->environment_rgb)) != RL_OK) return fail(rc);
    if (d->env_map_rgb && d->env_map_width && d->env_map_height &&
        (rc = rl_scene_set_environment_map(s, d->env_map_width, d->env_map_height, d->env_map_rgb)) != RL_OK) return fail(rc);
    if ((rc = rl_scene_enable_ats(s, d->build_ats ? 1 : 0)) != RL_OK) return fail(rc);
    if ((rc = rl_scene_build_emitters(s)) != RL_OK) return fail(rc);
    *out = s;
    return RL_OK;
}

void rl_path_params_default(rl_path_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->spp = 1;                               // Cli.nbsamples default (cli.rs:113-114)
    p->has_min_depth = 1; p->min_depth = 0;   // "0"
    p->has_max_depth = 0;                     // "inf" -> None
    p->has_rr_depth = 1; p->rr_depth = 0;     // "0"
    p->strategy = RL_STRATEGY_ALL;            // "all"
    p->single_scattering = 0;
    p->stream_mode = RL_STREAM_REFERENCE_ORDER;   // the drop-in default: the image `-r independent:SEED` renders in rustlight; per-sample streams are opt-in
    p->seed_variant = 0;
    p->shard_index = 0; p->shard_count = 1;
    p->pool_slots = 0;
}

}  // extern "C"
