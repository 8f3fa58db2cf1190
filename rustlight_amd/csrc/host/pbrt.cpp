// pbrt.cpp — loader for the PBRT-v3 subset rustlight's PBRT front end consumes
// (src/scene_loader.rs:77-315 + the material converter src/bsdfs/mod.rs:217-390).  The reference
// delegates parsing to the un-vendored `pbrt_rs` crate; this is a from-scratch tokenizer + a small
// graphics-state machine that produces the same `Scene` content:
//   * camera  = Camera::new(image_size, Fov::Y(fov), inverse(CTM at `Camera`), flip = false)
//   * meshes  = every `Shape "trianglemesh"` / `Shape "plymesh"` in file order, then the shapes of every
//               `ObjectInstance` (scene_loader.rs:170-204), points/normals transformed by instance matrix x shape CTM
//   * bsdf    = named / current material: matte, mirror, metal, glass, substrate; colours are constants or
//               `Texture "name" "spectrum" "imagemap"` bitmaps (Bitmap::read: .pfm / .png here)
//   * emission= `AreaLightSource "diffuse" "rgb L"` active in the current attribute scope
//   * lights  = LightSource point / distant / infinite (rgb L or mapname)
// `Include` is followed; other directives return RL_ERR_UNSUPPORTED.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "../kernels/wavefront.h"
#include "meshio.h"
#include "scene.h"

namespace rl {
namespace {

struct Token { enum Kind { Word, Str, Num, LBr, RBr, End } kind; std::string text; double num; };

struct Lexer {
    std::string src;
    size_t pos = 0;
    Token peeked; bool has_peek = false;
    Token next() {
        if (has_peek) { has_peek = false; return peeked; }
        while (pos < src.size()) {
            char c = src[pos];
            if (std::isspace((unsigned char)c)) { pos++; continue; }
            if (c == '#') { while (pos < src.size() && src[pos] != '\n') pos++; continue; }
            break;
        }
        if (pos >= src.size()) return {Token::End, "", 0};
        char c = src[pos];
        if (c == '[') { pos++; return {Token::LBr, "[", 0}; }
        if (c == ']') { pos++; return {Token::RBr, "]", 0}; }
        if (c == '"') {
            size_t e = src.find('"', pos + 1);
            if (e == std::string::npos) e = src.size();
            Token t{Token::Str, src.substr(pos + 1, e - pos - 1), 0};
            pos = e + 1;
            return t;
        }
        size_t s = pos;
        while (pos < src.size() && !std::isspace((unsigned char)src[pos]) && src[pos] != '[' && src[pos] != ']' && src[pos] != '"') pos++;
        std::string w = src.substr(s, pos - s);
        char* endp = nullptr;
        double v = std::strtod(w.c_str(), &endp);
        if (endp && *endp == 0 && !w.empty()) return {Token::Num, w, v};
        return {Token::Word, w, 0};
    }
    Token peek() { if (!has_peek) { peeked = next(); has_peek = true; } return peeked; }
};

struct Param { std::string type, name; std::vector<double> nums; std::vector<std::string> strs; };

// reads `"type name" value|[values]` pairs until the next directive
static bool read_params(Lexer& lx, std::vector<Param>* out) {
    out->clear();
    while (lx.peek().kind == Token::Str) {
        Token decl = lx.next();
        Param p;
        std::istringstream is(decl.text);
        is >> p.type >> p.name;
        Token t = lx.next();
        if (t.kind == Token::LBr) {
            for (;;) {
                Token v = lx.next();
                if (v.kind == Token::RBr) break;
                if (v.kind == Token::Num) p.nums.push_back(v.num);
                else if (v.kind == Token::Str || v.kind == Token::Word) p.strs.push_back(v.text);
                else return false;
            }
        } else if (t.kind == Token::Num) p.nums.push_back(t.num);
        else if (t.kind == Token::Str || t.kind == Token::Word) p.strs.push_back(t.text);
        else return false;
        out->push_back(std::move(p));
    }
    return true;
}
static const Param* find(const std::vector<Param>& ps, const char* name) {
    for (const Param& p : ps) if (p.name == name) return &p;
    return nullptr;
}

static rl_color_desc constant_color(float r, float g, float b) {
    rl_color_desc c;
    std::memset(&c, 0, sizeof(c));
    c.type = RL_TEX_CONSTANT;
    c.color0[0] = r; c.color0[1] = g; c.color0[2] = b;
    c.scale[0] = c.scale[1] = 1.0f;
    c.bitmap_id = -1;
    return c;
}
// named `Texture ... "imagemap"` bitmaps of the scene being loaded: name -> id from rl_scene_add_bitmap
static thread_local const std::map<std::string, int>* g_textures = nullptr;
static rl_color_desc color_param(const std::vector<Param>& ps, const char* name, float dr, float dg, float db) {
    const Param* p = find(ps, name);
    if (p && p->type == "texture" && !p->strs.empty() && g_textures) {   // Spectrum::Texture(name) -> BSDFColor::Bitmap (bsdfs/mod.rs:227-236)
        auto it = g_textures->find(p->strs[0]);
        if (it != g_textures->end()) { rl_color_desc c = constant_color(0, 0, 0); c.type = RL_TEX_BITMAP; c.bitmap_id = it->second; return c; }
    }
    if (p && p->nums.size() >= 3) return constant_color((float)p->nums[0], (float)p->nums[1], (float)p->nums[2]);
    if (p && p->nums.size() == 1) return constant_color((float)p->nums[0], (float)p->nums[0], (float)p->nums[0]);
    return constant_color(dr, dg, db);
}
static float float_param(const std::vector<Param>& ps, const char* name, float def) {
    const Param* p = find(ps, name);
    return (p && !p->nums.empty()) ? (float)p->nums[0] : def;
}

// pbrt roughness -> alpha (distribution_pbrt, src/bsdfs/mod.rs:254-293); host-only setup arithmetic
static float remap_roughness(float v, bool remap) {
    if (!remap) return v;
    float x = std::log(std::fmax(v, 1e-3f));
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

// bsdf_pbrt (src/bsdfs/mod.rs:295-390)
static rl_bsdf_desc make_material(const std::string& type, const std::vector<Param>& ps) {
    rl_bsdf_desc b;
    std::memset(&b, 0, sizeof(b));
    b.diffuse = b.specular = b.transmittance = constant_color(1, 1, 1);
    b.eta = constant_color(1, 1, 1);
    b.k = constant_color(0, 0, 0);
    b.glass_eta = 1.0f;
    const Param* remap_p = find(ps, "remaproughness");
    bool remap = !(remap_p && !remap_p->strs.empty() && remap_p->strs[0] == "false");
    float rough = float_param(ps, "roughness", 0.1f);
    float ur = float_param(ps, "uroughness", rough), vr = float_param(ps, "vroughness", rough);
    if (type == "matte") {
        b.type = RL_BSDF_DIFFUSE;
        b.diffuse = color_param(ps, "Kd", 0.5f, 0.5f, 0.5f);
    } else if (type == "mirror") {
        b.type = RL_BSDF_METAL;
        b.specular = color_param(ps, "Kr", 0.9f, 0.9f, 0.9f);
        b.distribution = RL_MICROFACET_NONE;
    } else if (type == "metal") {
        b.type = RL_BSDF_METAL;
        b.eta = color_param(ps, "eta", 0.2004376970f, 0.9240334304f, 1.1022119527f);
        b.k = color_param(ps, "k", 3.9129485033f, 2.4528477015f, 2.1421879552f);
        b.distribution = RL_MICROFACET_GGX;
        b.alpha_u = remap_roughness(ur, remap); b.alpha_v = remap_roughness(vr, remap);
    } else if (type == "glass") {
        b.type = RL_BSDF_GLASS;
        b.specular = color_param(ps, "Kr", 1, 1, 1);
        b.transmittance = color_param(ps, "Kt", 1, 1, 1);
        float eta = float_param(ps, "eta", float_param(ps, "index", 1.5f));
        b.glass_eta = eta / 1.0f;   // BSDFGlass::eta(eta, 1.0)
    } else if (type == "substrate") {
        b.type = RL_BSDF_SUBSTRATE;
        b.diffuse = color_param(ps, "Kd", 0.5f, 0.5f, 0.5f);
        b.specular = color_param(ps, "Ks", 0.5f, 0.5f, 0.5f);
        b.distribution = RL_MICROFACET_GGX;
        b.alpha_u = remap_roughness(ur, remap); b.alpha_v = remap_roughness(vr, remap);
    } else {   // unknown material: BSDFDiffuse(0.8) (bsdfs/mod.rs:384-389)
        b.type = RL_BSDF_DIFFUSE;
        b.diffuse = constant_color(0.8f, 0.8f, 0.8f);
    }
    return b;
}

static rl_bsdf_desc default_material() {   // no material bound: BSDFDiffuse(0.5) (scene_loader.rs:124-132)
    std::vector<Param> none;
    rl_bsdf_desc b = make_material("matte", none);
    return b;
}

struct GState {
    Mat4 ctm = Mat4::identity();
    bool has_material = false; rl_bsdf_desc material;
    bool has_emission = false; float emission[3] = {0, 0, 0};
    bool reverse_orientation = false;
};

static Mat4 look_at(const double* v) {
    // pbrt LookAt: camera-to-world from eye/look/up, returns world-to-camera (its inverse)
    Vec3 eye{(float)v[0], (float)v[1], (float)v[2]}, look{(float)v[3], (float)v[4], (float)v[5]}, up{(float)v[6], (float)v[7], (float)v[8]};
    Vec3 dir = vnormalize(vsub(look, eye));
    Vec3 right = vnormalize(vcross(vnormalize(up), dir));
    Vec3 new_up = vcross(dir, right);
    Mat4 c2w = Mat4::identity();
    c2w.m[0][0] = right.x; c2w.m[0][1] = right.y; c2w.m[0][2] = right.z;
    c2w.m[1][0] = new_up.x; c2w.m[1][1] = new_up.y; c2w.m[1][2] = new_up.z;
    c2w.m[2][0] = dir.x; c2w.m[2][1] = dir.y; c2w.m[2][2] = dir.z;
    c2w.m[3][0] = eye.x; c2w.m[3][1] = eye.y; c2w.m[3][2] = eye.z;
    Mat4 w2c;
    c2w.inverse(&w2c);
    return w2c;
}

// a shape as parsed: object-space data + the CTM / material / emission in force at its `Shape` directive
struct RawShape {
    std::vector<float> P, N, UV;
    std::vector<uint32_t> idx;
    Mat4 ctm = Mat4::identity();
    bool reverse_orientation = false;
    rl_bsdf_desc bsdf;
    bool has_emission = false; float emission[3] = {0, 0, 0};
};

// PBRTSceneLoader::transform_mesh (scene_loader.rs:88-157): mat = instance matrix * shape matrix
static int emit_shape(rl_scene* scene, const RawShape& r, const Mat4& instance, bool use_shading_normals) {
    const Mat4 mat = instance.times(r.ctm);
    const size_t nv = r.P.size() / 3;
    std::vector<float> pos(3 * nv), nrm;
    for (size_t i = 0; i < nv; i++) {   // mat.transform_point(p) (scene_loader.rs:118-121)
        Vec3 p = mat.xform_point({r.P[3 * i], r.P[3 * i + 1], r.P[3 * i + 2]});
        pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z;
    }
    if (use_shading_normals && r.N.size() == 3 * nv) {
        nrm.resize(3 * nv);
        for (size_t i = 0; i < nv; i++) {   // mat.transform_vector(+-n) (scene_loader.rs:101-116)
            Vec3 n{r.N[3 * i], r.N[3 * i + 1], r.N[3 * i + 2]};
            if (r.reverse_orientation) n = {-n.x, -n.y, -n.z};
            n = mat.xform_vector(n);
            nrm[3 * i] = n.x; nrm[3 * i + 1] = n.y; nrm[3 * i + 2] = n.z;
        }
    }
    return rl_scene_add_mesh(scene, pos.data(), nv, r.idx.data(), r.idx.size() / 3, nrm.empty() ? nullptr : nrm.data(),
                             r.UV.size() == 2 * nv ? r.UV.data() : nullptr, &r.bsdf, r.has_emission ? r.emission : nullptr);
}

}  // namespace

int load_pbrt(const char* path, bool use_shading_normals, rl_scene** out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { *err = std::string("cannot open ") + path; return RL_ERR_IO; }
    std::stringstream ss;
    ss << f.rdbuf();
    Lexer lx;
    lx.src = ss.str();
    // files named by the scene are relative to its directory (`wk` in scene_loader.rs:164-166)
    const std::string scene_path(path);
    const size_t slash = scene_path.find_last_of('/');
    const std::string base_dir = slash == std::string::npos ? std::string(".") : scene_path.substr(0, slash);
    auto join_path = [](const std::string& dir, const std::string& name) { return (!name.empty() && name[0] == '/') ? name : dir + "/" + name; };
    std::vector<GState> stack;
    GState gs;
    std::map<std::string, rl_bsdf_desc> named;
    std::map<std::string, int> textures;                       // Texture "name" ... "imagemap" -> bitmap id
    std::map<std::string, std::vector<RawShape>> objects;      // ObjectBegin "name" ... ObjectEnd
    std::vector<std::pair<std::string, Mat4>> instances;       // ObjectInstance "name" with the CTM in force
    std::string cur_object; bool in_object = false;
    std::map<std::string, Mat4> coord_systems;                 // CoordinateSystem "name" / CoordSysTransform "name"
    g_textures = &textures;
    rl_scene* scene = new rl_scene();
    uint32_t width = 512, height = 512;
    float fov = 90.0f;
    bool have_camera = false;
    Mat4 world_to_camera = Mat4::identity();
    std::vector<Param> ps;
    auto fail = [&](int code, const std::string& msg) { *err = msg; delete scene; g_textures = nullptr; return code; };
    for (;;) {
        Token t = lx.next();
        if (t.kind == Token::End) break;
        if (t.kind != Token::Word) return fail(RL_ERR_PARSE, "unexpected token '" + t.text + "'");
        const std::string& d = t.text;
        if (d == "Transform" || d == "ConcatTransform") {
            if (lx.next().kind != Token::LBr) return fail(RL_ERR_PARSE, d + ": expected [");
            float m[16]; int n = 0;
            for (;;) { Token v = lx.next(); if (v.kind == Token::RBr) break; if (v.kind != Token::Num || n >= 16) return fail(RL_ERR_PARSE, d + ": bad matrix"); m[n++] = (float)v.num; }
            if (n != 16) return fail(RL_ERR_PARSE, d + ": need 16 numbers");
            Mat4 mm = Mat4::from_cols(m);
            gs.ctm = d == "Transform" ? mm : gs.ctm.times(mm);
        } else if (d == "Identity") {
            gs.ctm = Mat4::identity();
        } else if (d == "Translate" || d == "Scale") {
            double v[3];
            for (int i = 0; i < 3; i++) { Token n = lx.next(); if (n.kind != Token::Num) return fail(RL_ERR_PARSE, d + ": need 3 numbers"); v[i] = n.num; }
            gs.ctm = gs.ctm.times(d == "Translate" ? Mat4::translate((float)v[0], (float)v[1], (float)v[2]) : Mat4::scale((float)v[0], (float)v[1], (float)v[2]));
        } else if (d == "Rotate") {                  // Rotate angle x y z (degrees about an axis through the origin)
            double v[4];
            for (int i = 0; i < 4; i++) { Token n = lx.next(); if (n.kind != Token::Num) return fail(RL_ERR_PARSE, "Rotate: need 4 numbers"); v[i] = n.num; }
            const Vec3 a = vnormalize({(float)v[1], (float)v[2], (float)v[3]});
            const float ang = (float)v[0] * 3.14159265358979323846f / 180.0f, sn = std::sin(ang), cs = std::cos(ang);
            Mat4 r = Mat4::identity();
            r.m[0][0] = a.x * a.x + (1.0f - a.x * a.x) * cs; r.m[1][0] = a.x * a.y * (1.0f - cs) - a.z * sn;   r.m[2][0] = a.x * a.z * (1.0f - cs) + a.y * sn;
            r.m[0][1] = a.x * a.y * (1.0f - cs) + a.z * sn;   r.m[1][1] = a.y * a.y + (1.0f - a.y * a.y) * cs; r.m[2][1] = a.y * a.z * (1.0f - cs) - a.x * sn;
            r.m[0][2] = a.x * a.z * (1.0f - cs) - a.y * sn;   r.m[1][2] = a.y * a.z * (1.0f - cs) + a.x * sn;   r.m[2][2] = a.z * a.z + (1.0f - a.z * a.z) * cs;
            gs.ctm = gs.ctm.times(r);
        } else if (d == "CoordinateSystem") {
            Token name = lx.next();
            coord_systems[name.text] = gs.ctm;
        } else if (d == "CoordSysTransform") {
            Token name = lx.next();
            auto it = coord_systems.find(name.text);
            if (it != coord_systems.end()) gs.ctm = it->second;
        } else if (d == "MediumInterface") {         // participating media come from the CLI's -m, not from the scene file
            while (lx.peek().kind == Token::Str) lx.next();
        } else if (d == "MakeNamedMedium") {
            lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "MakeNamedMedium: bad parameters");
        } else if (d == "LookAt") {
            double v[9];
            for (int i = 0; i < 9; i++) { Token n = lx.next(); if (n.kind != Token::Num) return fail(RL_ERR_PARSE, "LookAt: need 9 numbers"); v[i] = n.num; }
            gs.ctm = gs.ctm.times(look_at(v));
        } else if (d == "Camera") {
            Token ty = lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "Camera: bad parameters");
            if (ty.text != "perspective") return fail(RL_ERR_UNSUPPORTED, "Camera: only perspective is supported");
            fov = float_param(ps, "fov", 90.0f);
            world_to_camera = gs.ctm;
            have_camera = true;
        } else if (d == "Film") {
            lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "Film: bad parameters");
            width = (uint32_t)float_param(ps, "xresolution", 512.0f);
            height = (uint32_t)float_param(ps, "yresolution", 512.0f);
        } else if (d == "Sampler" || d == "PixelFilter" || d == "Integrator" || d == "Accelerator") {
            lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, d + ": bad parameters");
        } else if (d == "WorldBegin") {
            gs = GState();
            stack.clear();
        } else if (d == "WorldEnd") {
        } else if (d == "AttributeBegin" || d == "TransformBegin") {
            stack.push_back(gs);
        } else if (d == "AttributeEnd" || d == "TransformEnd") {
            if (stack.empty()) return fail(RL_ERR_PARSE, d + " without Begin");
            if (d == "TransformEnd") { Mat4 keep = stack.back().ctm; stack.pop_back(); gs.ctm = keep; }
            else { gs = stack.back(); stack.pop_back(); }
        } else if (d == "ReverseOrientation") {
            gs.reverse_orientation = !gs.reverse_orientation;
        } else if (d == "MakeNamedMaterial") {
            Token name = lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "MakeNamedMaterial: bad parameters");
            const Param* ty = find(ps, "type");
            named[name.text] = make_material(ty && !ty->strs.empty() ? ty->strs[0] : "matte", ps);
        } else if (d == "NamedMaterial") {
            Token name = lx.next();
            auto it = named.find(name.text);
            if (it != named.end()) { gs.material = it->second; gs.has_material = true; }
            else gs.has_material = false;
        } else if (d == "Material") {
            Token ty = lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "Material: bad parameters");
            gs.material = make_material(ty.text, ps);
            gs.has_material = true;
        } else if (d == "AreaLightSource") {
            lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "AreaLightSource: bad parameters");
            const Param* L = find(ps, "L");
            gs.has_emission = true;
            for (int i = 0; i < 3; i++) gs.emission[i] = (L && L->nums.size() >= 3) ? (float)L->nums[i] : 1.0f;
        } else if (d == "LightSource") {   // scene_loader.rs:177-224
            Token ty = lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "LightSource: bad parameters");
            rl_color_desc sc = color_param(ps, "scale", 1, 1, 1);
            if (ty.text == "point") {
                rl_color_desc I = color_param(ps, "I", 1, 1, 1);
                const Param* from = find(ps, "from");
                Vec3 pos = gs.ctm.xform_point({from && from->nums.size() >= 3 ? (float)from->nums[0] : 0.0f, from && from->nums.size() >= 3 ? (float)from->nums[1] : 0.0f,
                                               from && from->nums.size() >= 3 ? (float)from->nums[2] : 0.0f});
                float p3[3] = {pos.x, pos.y, pos.z}, i3[3] = {I.color0[0] * sc.color0[0], I.color0[1] * sc.color0[1], I.color0[2] * sc.color0[2]};
                rl_scene_add_point_light(scene, p3, i3);
            } else if (ty.text == "distant") {
                rl_color_desc Lc = color_param(ps, "L", 1, 1, 1);
                const Param* from = find(ps, "from");
                const Param* to = find(ps, "to");
                Vec3 f{0, 0, 0}, t{0, 0, 1};
                if (from && from->nums.size() >= 3) f = {(float)from->nums[0], (float)from->nums[1], (float)from->nums[2]};
                if (to && to->nums.size() >= 3) t = {(float)to->nums[0], (float)to->nums[1], (float)to->nums[2]};
                Vec3 dir = vnormalize(vsub(t, f));   // (to - from).normalize()
                float d3[3] = {dir.x, dir.y, dir.z}, i3[3] = {Lc.color0[0] * sc.color0[0], Lc.color0[1] * sc.color0[1], Lc.color0[2] * sc.color0[2]};
                rl_scene_add_directional_light(scene, d3, i3);
            } else if (ty.text == "infinite") {
                if (const Param* mp = find(ps, "mapname")) {
                    // Spectrum::Mapname: Bitmap::read(wk.join(name)) -> EnvironmentLightColor::new_texture; scale must be 1 (scene_loader.rs:259-271)
                    if (mp->strs.empty()) return fail(RL_ERR_PARSE, "LightSource infinite: mapname needs a file name");
                    if (sc.color0[0] != 1.0f || sc.color0[1] != 1.0f || sc.color0[2] != 1.0f) return fail(RL_ERR_UNSUPPORTED, "LightSource infinite: scale must be 1 with a mapname");
                    const std::string file = join_path(base_dir, mp->strs[0]);
                    HostBitmap env; std::string ierr;                    // Bitmap::read: .pfm | .exr | LDR by extension
                    int irc = read_image(file, &env, &ierr);
                    if (irc != RL_OK) return fail(irc, "LightSource infinite: " + ierr);
                    const uint32_t ew = env.w, eh = env.h; const std::vector<float>& texels = env.rgb;
                    rl_scene_set_environment_map(scene, ew, eh, texels.data());
                    continue;
                }
                rl_color_desc Lc = color_param(ps, "L", 1, 1, 1);
                float e3[3] = {Lc.color0[0] * sc.color0[0], Lc.color0[1] * sc.color0[1], Lc.color0[2] * sc.color0[2]};
                rl_scene_set_environment(scene, e3);
            } else return fail(RL_ERR_UNSUPPORTED, "LightSource \"" + ty.text + "\" is not supported");
        } else if (d == "Shape") {
            Token ty = lx.next();
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "Shape: bad parameters");
            RawShape r;
            if (ty.text == "trianglemesh") {
                const Param* P = find(ps, "P");
                const Param* I = find(ps, "indices");
                const Param* N = find(ps, "N");
                const Param* UV = find(ps, "uv");
                if (!UV) UV = find(ps, "st");
                if (!P || !I || P->nums.size() % 3 || I->nums.size() % 3) return fail(RL_ERR_PARSE, "Shape trianglemesh: bad P / indices");
                const size_t nv = P->nums.size() / 3;
                r.P.assign(P->nums.begin(), P->nums.end());
                if (N && N->nums.size() == 3 * nv) r.N.assign(N->nums.begin(), N->nums.end());
                if (UV && UV->nums.size() == 2 * nv) r.UV.assign(UV->nums.begin(), UV->nums.end());
                r.idx.resize(I->nums.size());
                for (size_t i = 0; i < r.idx.size(); i++) r.idx[i] = (uint32_t)I->nums[i];
            } else if (ty.text == "plymesh") {   // Shape::Ply -> read_ply(..).to_trimesh() (scene_loader.rs:88-93)
                const Param* fn = find(ps, "filename");
                if (!fn || fn->strs.empty()) return fail(RL_ERR_PARSE, "Shape plymesh: filename missing");
                LoadedMesh m;
                std::string e2;
                if (int rc = read_ply(join_path(base_dir, fn->strs[0]), &m, &e2)) return fail(rc, e2);
                r.P = std::move(m.pos); r.N = std::move(m.nrm); r.UV = std::move(m.uv); r.idx = std::move(m.idx);
            } else return fail(RL_ERR_UNSUPPORTED, "Shape \"" + ty.text + "\" is not supported (trianglemesh, plymesh)");
            r.ctm = gs.ctm;
            r.reverse_orientation = gs.reverse_orientation;
            r.bsdf = gs.has_material ? gs.material : default_material();
            r.has_emission = gs.has_emission;
            for (int k = 0; k < 3; k++) r.emission[k] = gs.emission[k];
            if (in_object) objects[cur_object].push_back(std::move(r));
            else if (emit_shape(scene, r, Mat4::identity(), use_shading_normals) < 0) return fail(RL_ERR_PARSE, "Shape: invalid mesh");
        } else if (d == "Texture") {             // named imagemap textures (scene_info.textures; bsdfs/mod.rs:227-236)
            Token name = lx.next(); Token kind = lx.next(); Token cls = lx.next();
            (void)kind;
            if (!read_params(lx, &ps)) return fail(RL_ERR_PARSE, "Texture: bad parameters");
            if (cls.text != "imagemap") return fail(RL_ERR_UNSUPPORTED, "Texture class \"" + cls.text + "\" is not supported (imagemap only)");
            const Param* fn = find(ps, "filename");
            if (!fn || fn->strs.empty()) return fail(RL_ERR_PARSE, "Texture imagemap: filename missing");
            HostBitmap img;
            std::string e2;
            if (int rc = read_image(join_path(base_dir, fn->strs[0]), &img, &e2)) return fail(rc, e2);
            int id = rl_scene_add_bitmap(scene, img.w, img.h, img.rgb.data());
            if (id < 0) return fail(id, "Texture imagemap: bitmap rejected");
            textures[name.text] = id;
        } else if (d == "Include") {             // the included text is parsed in place
            Token file = lx.next();
            std::ifstream inc(join_path(base_dir, file.text), std::ios::binary);
            if (!inc) return fail(RL_ERR_IO, "cannot open Include " + file.text);
            std::stringstream is2;
            is2 << inc.rdbuf();
            lx.has_peek = false;
            lx.src.insert(lx.pos, "\n" + is2.str() + "\n");
        } else if (d == "ObjectBegin") {
            Token name = lx.next();
            if (in_object) return fail(RL_ERR_PARSE, "nested ObjectBegin");
            stack.push_back(gs);
            in_object = true; cur_object = name.text; objects[cur_object];
        } else if (d == "ObjectEnd") {
            if (!in_object || stack.empty()) return fail(RL_ERR_PARSE, "ObjectEnd without ObjectBegin");
            in_object = false;
            gs = stack.back(); stack.pop_back();
        } else if (d == "ObjectInstance") {
            Token name = lx.next();
            instances.emplace_back(name.text, gs.ctm);
        } else {
            return fail(RL_ERR_UNSUPPORTED, "directive '" + d + "' is not supported");
        }
    }
    for (const auto& inst : instances) {   // scene_info.instances, after all plain shapes (scene_loader.rs:186-204)
        auto it = objects.find(inst.first);
        if (it == objects.end()) return fail(RL_ERR_PARSE, "ObjectInstance of unknown object " + inst.first);
        for (const RawShape& r : it->second)
            if (emit_shape(scene, r, inst.second, use_shading_normals) < 0) return fail(RL_ERR_PARSE, "ObjectInstance: invalid mesh");
    }
    g_textures = nullptr;
    if (!have_camera) return fail(RL_ERR_PARSE, "The camera is not set!");
    Mat4 to_world;
    if (!world_to_camera.inverse(&to_world)) return fail(RL_ERR_PARSE, "singular camera transform");
    float cols[16];
    to_world.to_cols(cols);
    int rc = rl_scene_set_camera(scene, width, height, fov, 1 /* Fov::Y */, cols, 0 /* flip = false */);
    if (rc != RL_OK) return fail(rc, "invalid camera");
    *out = scene;
    return RL_OK;
}

}  // namespace rl

extern "C" int rl_scene_load_pbrt(const char* path, int use_shading_normals, rl_scene** out) {
    if (!path || !out) return RL_ERR_INVALID_ARGUMENT;
    std::string err;
    int rc;
    try { rc = rl::load_pbrt(path, use_shading_normals != 0, out, &err); }
    catch (const std::exception& e) { rc = RL_ERR_PARSE; err = std::string(path) + ": malformed scene (" + e.what() + ")"; }   // nothing is thrown across the C ABI
    if (rc != RL_OK) rl_set_error(err);
    return rc;
}
