// mitsuba.cpp — loader for the Mitsuba 0.5/0.6 XML subset rustlight's Mitsuba front end consumes
// (src/scene_loader.rs:318-795 + the material converter bsdf_mts, src/bsdfs/mod.rs:392-612).  The reference
// delegates parsing to the un-vendored `mitsuba_rs` crate; this is a from-scratch XML reader plus the same scene
// assembly:
//   * camera  = Camera::new((film.width, film.height), Fov::X|Y(fov) by fovAxis, sensor toWorld, flip = true)
//   * shapes  = obj | ply | serialized | rectangle | sphere (32 x 32 UV tessellation, scene_loader.rs:598-629), each
//               with its bsdf (inline or <ref>), optional area emitter radiance, toWorld applied as apply_transform does
//               (normals: transform_vector then renormalise; points: transform_point);
//   * bsdf    = twosided(unwrapped) | diffuse | phong | dielectric | plastic / roughplastic | conductor / roughconductor,
//               anything else BSDFDiffuse(0.8); colours: rgb / spectrum / srgb constants, bitmap / checkerboard /
//               gridtexture textures;
//   * emitters= point; medium = the first homogeneous medium (sigmaS, sigmaA x scale, isotropic | hg phase).
// Shapes are kept in file order (the reference iterates a HashMap of named shapes first, i.e. in no defined order).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "../kernels/wavefront.h"
#include "meshio.h"

namespace rl {
namespace {

// ---------------------------------------------------------------------------------------------- XML
struct XmlNode {
    std::string tag;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<XmlNode>> children;
    const std::string& get(const std::string& k, const std::string& def = empty()) const { auto it = attr.find(k); return it == attr.end() ? def : it->second; }
    bool has(const std::string& k) const { return attr.count(k) != 0; }
    static const std::string& empty() { static const std::string e; return e; }
};

struct XmlParser {
    const std::string& s; size_t p = 0; std::string err;
    void skip_ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) p++; }
    bool skip_misc() {   // whitespace, comments, <?...?>, <!DOCTYPE ...>
        for (;;) {
            skip_ws();
            if (s.compare(p, 4, "<!--") == 0) { size_t e = s.find("-->", p + 4); if (e == std::string::npos) { err = "unterminated comment"; return false; } p = e + 3; }
            else if (s.compare(p, 2, "<?") == 0) { size_t e = s.find("?>", p + 2); if (e == std::string::npos) { err = "unterminated declaration"; return false; } p = e + 2; }
            else if (s.compare(p, 2, "<!") == 0) { size_t e = s.find('>', p + 2); if (e == std::string::npos) { err = "unterminated declaration"; return false; } p = e + 1; }
            else return true;
        }
    }
    static std::string unescape(const std::string& v) {
        std::string o;
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i] != '&') { o.push_back(v[i]); continue; }
            if (v.compare(i, 4, "&lt;") == 0) { o.push_back('<'); i += 3; }
            else if (v.compare(i, 4, "&gt;") == 0) { o.push_back('>'); i += 3; }
            else if (v.compare(i, 5, "&amp;") == 0) { o.push_back('&'); i += 4; }
            else if (v.compare(i, 6, "&quot;") == 0) { o.push_back('"'); i += 5; }
            else if (v.compare(i, 6, "&apos;") == 0) { o.push_back('\''); i += 5; }
            else o.push_back('&');
        }
        return o;
    }
    std::unique_ptr<XmlNode> element() {
        if (!skip_misc()) return nullptr;
        if (p >= s.size() || s[p] != '<') { err = "expected '<'"; return nullptr; }
        p++;
        std::unique_ptr<XmlNode> n(new XmlNode());
        while (p < s.size() && !std::isspace((unsigned char)s[p]) && s[p] != '>' && s[p] != '/') n->tag.push_back(s[p++]);
        for (;;) {
            skip_ws();
            if (p >= s.size()) { err = "unterminated tag <" + n->tag; return nullptr; }
            if (s[p] == '/') { if (p + 1 < s.size() && s[p + 1] == '>') { p += 2; return n; } err = "stray '/'"; return nullptr; }
            if (s[p] == '>') { p++; break; }
            std::string k;
            while (p < s.size() && s[p] != '=' && !std::isspace((unsigned char)s[p])) k.push_back(s[p++]);
            skip_ws();
            if (p >= s.size() || s[p] != '=') { err = "attribute without value in <" + n->tag; return nullptr; }
            p++; skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) { err = "unquoted attribute in <" + n->tag; return nullptr; }
            const char q = s[p++];
            size_t e = s.find(q, p);
            if (e == std::string::npos) { err = "unterminated attribute in <" + n->tag; return nullptr; }
            n->attr[k] = unescape(s.substr(p, e - p));
            p = e + 1;
        }
        for (;;) {   // children until </tag>
            if (!skip_misc()) return nullptr;
            if (p >= s.size()) { err = "missing </" + n->tag + ">"; return nullptr; }
            if (s[p] != '<') { while (p < s.size() && s[p] != '<') p++; continue; }   // text content is not used by the format
            if (s.compare(p, 2, "</") == 0) {
                size_t e = s.find('>', p);
                if (e == std::string::npos) { err = "unterminated closing tag"; return nullptr; }
                p = e + 1;
                return n;
            }
            std::unique_ptr<XmlNode> c = element();
            if (!c) return nullptr;
            n->children.push_back(std::move(c));
        }
    }
};

// ---------------------------------------------------------------------------------------------- values
std::vector<float> numbers(const std::string& v) {
    std::vector<float> out;
    const char* c = v.c_str();
    while (*c) {
        while (*c && (std::isspace((unsigned char)*c) || *c == ',')) c++;
        if (!*c) break;
        char* e = nullptr;
        float f = std::strtof(c, &e);
        if (e == c) break;
        out.push_back(f);
        c = e;
    }
    return out;
}

struct Loader {
    std::string base_dir;
    bool use_shading_normals = true;
    rl_scene* scene = nullptr;
    std::map<std::string, std::string> defaults;          // <default name value>
    std::map<std::string, const XmlNode*> bsdf_by_id;
    std::string err;
    std::vector<std::unique_ptr<XmlNode>> included;       // keeps <include>d trees alive

    std::string subst(const std::string& v) const {       // $name -> default
        if (v.find('$') == std::string::npos) return v;
        std::string o;
        for (size_t i = 0; i < v.size();) {
            if (v[i] != '$') { o.push_back(v[i++]); continue; }
            size_t j = i + 1;
            while (j < v.size() && (std::isalnum((unsigned char)v[j]) || v[j] == '_')) j++;
            auto it = defaults.find(v.substr(i + 1, j - i - 1));
            o += it == defaults.end() ? v.substr(i, j - i) : it->second;
            i = j;
        }
        return o;
    }
    std::string value(const XmlNode& n, const char* key = "value") const { return subst(n.get(key)); }
    const XmlNode* child(const XmlNode& n, const std::string& name) const {
        for (const auto& c : n.children) if (c->get("name") == name) return c.get();
        return nullptr;
    }
    float float_prop(const XmlNode& n, const std::string& name, float def) const {
        const XmlNode* c = child(n, name);
        if (!c || (c->tag != "float" && c->tag != "integer")) return def;
        std::vector<float> v = numbers(value(*c));
        return v.empty() ? def : v[0];
    }
    std::string string_prop(const XmlNode& n, const std::string& name, const std::string& def) const {
        const XmlNode* c = child(n, name);
        return (c && c->tag == "string") ? value(*c) : def;
    }
    bool bool_prop(const XmlNode& n, const std::string& name, bool def) const {
        const XmlNode* c = child(n, name);
        return (c && c->tag == "boolean") ? value(*c) == "true" : def;
    }
    // RGB / spectrum constants -> as_rgb()
    bool rgb_value(const XmlNode& c, float out[3]) const {
        if (c.tag == "rgb" || c.tag == "spectrum" || c.tag == "srgb" || c.tag == "color") {
            std::string v = value(c);
            if (c.tag == "srgb" && !v.empty() && v[0] == '#' && v.size() == 7) {
                for (int k = 0; k < 3; k++) out[k] = (float)std::strtol(v.substr(1 + 2 * k, 2).c_str(), nullptr, 16) / 255.0f;
            } else {
                std::vector<float> f = numbers(v);
                if (f.size() >= 3) { out[0] = f[0]; out[1] = f[1]; out[2] = f[2]; }
                else if (f.size() == 1) out[0] = out[1] = out[2] = f[0];
                else return false;
            }
            if (c.tag == "srgb") for (int k = 0; k < 3; k++) out[k] = out[k] <= 0.04045f ? out[k] / 12.92f : std::pow((out[k] + 0.055f) / 1.055f, 2.4f);
            return true;
        }
        return false;
    }
    void rgb_prop(const XmlNode& n, const std::string& name, float def, float out[3]) const {
        out[0] = out[1] = out[2] = def;
        const XmlNode* c = child(n, name);
        if (c) rgb_value(*c, out);
    }

    // Transform: operations compose in document order, each one applied after the previous (M = op * M)
    Mat4 transform(const XmlNode& t) const {
        Mat4 m = Mat4::identity();
        for (const auto& c : t.children) {
            Mat4 op = Mat4::identity();
            if (c->tag == "matrix") {
                std::vector<float> v = numbers(value(*c));
                if (v.size() == 16) for (int r = 0; r < 4; r++) for (int col = 0; col < 4; col++) op.m[col][r] = v[4 * r + col];   // row-major text
                else if (v.size() == 9) for (int r = 0; r < 3; r++) for (int col = 0; col < 3; col++) op.m[col][r] = v[3 * r + col];
            } else if (c->tag == "translate") {
                op = Mat4::translate(attr_f(*c, "x", 0), attr_f(*c, "y", 0), attr_f(*c, "z", 0));
            } else if (c->tag == "scale") {
                if (c->has("value")) { float s = attr_f(*c, "value", 1); op = Mat4::scale(s, s, s); }
                else op = Mat4::scale(attr_f(*c, "x", 1), attr_f(*c, "y", 1), attr_f(*c, "z", 1));
            } else if (c->tag == "rotate") {
                Vec3 a = vnormalize({attr_f(*c, "x", 0), attr_f(*c, "y", 0), attr_f(*c, "z", 0)});
                float ang = attr_f(*c, "angle", 0) * 3.14159265358979323846f / 180.0f;
                float s = std::sin(ang), co = std::cos(ang), t1 = 1.0f - co;
                op.m[0][0] = t1 * a.x * a.x + co;       op.m[1][0] = t1 * a.x * a.y - s * a.z; op.m[2][0] = t1 * a.x * a.z + s * a.y;
                op.m[0][1] = t1 * a.x * a.y + s * a.z;  op.m[1][1] = t1 * a.y * a.y + co;      op.m[2][1] = t1 * a.y * a.z - s * a.x;
                op.m[0][2] = t1 * a.x * a.z - s * a.y;  op.m[1][2] = t1 * a.y * a.z + s * a.x; op.m[2][2] = t1 * a.z * a.z + co;
            } else if (c->tag == "lookat" || c->tag == "lookAt") {
                std::vector<float> o = numbers(value(*c, "origin")), tg = numbers(value(*c, "target")), up = numbers(value(*c, "up"));
                if (up.size() < 3) up = {0, 1, 0};
                if (o.size() >= 3 && tg.size() >= 3) {
                    Vec3 dir = vnormalize(vsub({tg[0], tg[1], tg[2]}, {o[0], o[1], o[2]}));
                    Vec3 left = vnormalize(vcross({up[0], up[1], up[2]}, dir));
                    Vec3 nup = vcross(dir, left);
                    op.m[0][0] = left.x; op.m[0][1] = left.y; op.m[0][2] = left.z;
                    op.m[1][0] = nup.x;  op.m[1][1] = nup.y;  op.m[1][2] = nup.z;
                    op.m[2][0] = dir.x;  op.m[2][1] = dir.y;  op.m[2][2] = dir.z;
                    op.m[3][0] = o[0];   op.m[3][1] = o[1];   op.m[3][2] = o[2];
                }
            } else continue;
            m = op.times(m);
        }
        return m;
    }
    float attr_f(const XmlNode& n, const char* k, float def) const { std::vector<float> v = numbers(value(n, k)); return v.empty() ? def : v[0]; }

    // ---- BSDFColor (bsdf_texture_match_mts, bsdfs/mod.rs:403-455)
    static rl_color_desc constant(float r, float g, float b) {
        rl_color_desc c;
        std::memset(&c, 0, sizeof(c));
        c.type = RL_TEX_CONSTANT; c.color0[0] = r; c.color0[1] = g; c.color0[2] = b; c.scale[0] = c.scale[1] = 1.0f; c.bitmap_id = -1;
        return c;
    }
    bool color_prop(const XmlNode& n, const std::string& name, float def, rl_color_desc* out, float avg[3]) {
        *out = constant(def, def, def);
        avg[0] = avg[1] = avg[2] = def;
        const XmlNode* c = child(n, name);
        if (!c) return true;
        float v[3];
        if (rgb_value(*c, v)) { *out = constant(v[0], v[1], v[2]); std::memcpy(avg, v, sizeof(v)); return true; }
        if (c->tag != "texture") return true;
        const std::string ty = c->get("type");
        if (ty == "bitmap") {
            HostBitmap img;
            std::string file = string_prop(*c, "filename", "");
            if (file.empty() || file[0] != '/') file = base_dir + "/" + file;
            if (read_image(file, &img, &err) != RL_OK) return false;
            const float gamma = float_prop(*c, "gamma", 1.0f);
            if (gamma != 1.0f) for (float& t : img.rgb) t = std::pow(t, 1.0f / gamma);          // img.gamma(1.0 / gamma)
            int id = rl_scene_add_bitmap(scene, img.w, img.h, img.rgb.data());
            if (id < 0) { err = "bitmap texture rejected"; return false; }
            out->type = RL_TEX_BITMAP; out->bitmap_id = id;
            float s[3] = {0, 0, 0};                                                                 // Bitmap::average
            for (size_t i = 0; i < img.rgb.size(); i += 3) { s[0] += img.rgb[i]; s[1] += img.rgb[i + 1]; s[2] += img.rgb[i + 2]; }
            const float inv = 1.0f / (float)(img.rgb.size() / 3);
            for (int k = 0; k < 3; k++) avg[k] = s[k] * inv;
            return true;
        }
        if (ty == "checkerboard" || ty == "gridtexture") {
            float c0[3], c1[3];
            rgb_prop(*c, "color0", ty == "checkerboard" ? 0.4f : 0.2f, c0);
            rgb_prop(*c, "color1", ty == "checkerboard" ? 0.2f : 0.4f, c1);
            *out = constant(c0[0], c0[1], c0[2]);
            out->type = ty == "checkerboard" ? RL_TEX_CHECKERBOARD : RL_TEX_GRID;
            for (int k = 0; k < 3; k++) out->color1[k] = c1[k];
            out->offset[0] = float_prop(*c, "uoffset", 0.0f); out->offset[1] = float_prop(*c, "voffset", 0.0f);
            out->scale[0] = float_prop(*c, "uscale", 1.0f); out->scale[1] = float_prop(*c, "vscale", 1.0f);
            out->line_width = float_prop(*c, "lineWidth", 0.01f);
            for (int k = 0; k < 3; k++) avg[k] = 0.5f * (c0[k] + c1[k]);
            return true;
        }
        err = "Mitsuba texture type not supported: " + ty;
        return false;
    }
    // distribution_mts (bsdfs/mod.rs:457-497): Some only for the rough variants
    void distribution(const XmlNode& n, bool rough, rl_bsdf_desc* b) const {
        if (!rough) { b->distribution = RL_MICROFACET_NONE; return; }
        const std::string d = string_prop(n, "distribution", "beckmann");
        b->distribution = d == "ggx" ? RL_MICROFACET_GGX : RL_MICROFACET_BECKMANN;
        float a = float_prop(n, "alpha", 0.1f);
        b->alpha_u = float_prop(n, "alphaU", a); b->alpha_v = float_prop(n, "alphaV", a);
    }
    static float luminance(const float c[3]) { return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }

    // bsdf_mts (bsdfs/mod.rs:499-612)
    bool make_bsdf(const XmlNode* n, rl_bsdf_desc* b) {
        std::memset(b, 0, sizeof(*b));
        b->diffuse = b->specular = b->transmittance = b->eta = constant(1, 1, 1);
        b->k = constant(0, 0, 0);
        b->glass_eta = 1.0f;
        b->type = RL_BSDF_DIFFUSE;
        if (!n) { b->diffuse = constant(0.8f, 0.8f, 0.8f); return true; }                          // no bsdf: BSDFDiffuse(0.8)
        if (n->tag == "ref") { auto it = bsdf_by_id.find(n->get("id")); return make_bsdf(it == bsdf_by_id.end() ? nullptr : it->second, b); }
        const std::string ty = n->get("type");
        float avg_d[3], avg_s[3];
        if (ty == "twosided") {
            for (const auto& c : n->children) if (c->tag == "bsdf" || c->tag == "ref") return make_bsdf(c.get(), b);
            return make_bsdf(nullptr, b);
        } else if (ty == "diffuse") {
            return color_prop(*n, "reflectance", 0.5f, &b->diffuse, avg_d);
        } else if (ty == "phong") {
            b->type = RL_BSDF_PHONG;
            if (!color_prop(*n, "specularReflectance", 0.2f, &b->specular, avg_s) || !color_prop(*n, "diffuseReflectance", 0.5f, &b->diffuse, avg_d)) return false;
            b->exponent = float_prop(*n, "exponent", 30.0f);
            const float d = luminance(avg_d), s = luminance(avg_s);
            if (d + s == 0.0f) { err = "phong: diffuse + specular reflectance is zero"; return false; }
            b->weight_specular = s / (d + s);
            return true;
        } else if (ty == "dielectric") {
            b->type = RL_BSDF_GLASS;
            if (!color_prop(*n, "specularReflectance", 1.0f, &b->specular, avg_s) || !color_prop(*n, "specularTransmittance", 1.0f, &b->transmittance, avg_d)) return false;
            b->glass_eta = float_prop(*n, "intIOR", 1.5046f) / float_prop(*n, "extIOR", 1.000277f);   // BSDFGlass::eta(int, ext)
            return true;
        } else if (ty == "plastic" || ty == "roughplastic") {
            b->type = RL_BSDF_SUBSTRATE;
            if (!color_prop(*n, "specularReflectance", 1.0f, &b->specular, avg_s) || !color_prop(*n, "diffuseReflectance", 0.5f, &b->diffuse, avg_d)) return false;
            distribution(*n, ty == "roughplastic", b);
            return true;
        } else if (ty == "conductor" || ty == "roughconductor") {
            b->type = RL_BSDF_METAL;
            if (!color_prop(*n, "specularReflectance", 1.0f, &b->specular, avg_s)) return false;
            float eta[3], k[3];
            const float ext = float_prop(*n, "extEta", 1.000277f);
            const float cu_eta[3] = {0.2004376970f, 0.9240334304f, 1.1022119527f}, cu_k[3] = {3.9129485033f, 2.4528477015f, 2.1421879552f};
            const XmlNode* ce = child(*n, "eta"); const XmlNode* ck = child(*n, "k");
            if (!(ce && rgb_value(*ce, eta))) std::memcpy(eta, cu_eta, sizeof(eta));
            if (!(ck && rgb_value(*ck, k))) std::memcpy(k, cu_k, sizeof(k));
            b->eta = constant(eta[0] / ext, eta[1] / ext, eta[2] / ext);
            b->k = constant(k[0] / ext, k[1] / ext, k[2] / ext);
            distribution(*n, ty == "roughconductor", b);
            return true;
        }
        b->diffuse = constant(0.8f, 0.8f, 0.8f);                                                   // unknown type: BSDFDiffuse(0.8)
        return true;
    }

    // ---- shapes
    bool add_mesh(LoadedMesh& m, const XmlNode& shape, bool face_normals, const rl_bsdf_desc* forced_bsdf = nullptr) {
        const XmlNode* bs = nullptr; const XmlNode* em = nullptr; const XmlNode* tw = nullptr;
        for (const auto& c : shape.children) {
            if (c->tag == "bsdf" || (c->tag == "ref" && bsdf_by_id.count(c->get("id")))) bs = c.get();
            else if (c->tag == "emitter") em = c.get();
            else if (c->tag == "transform" && c->get("name") == "toWorld") tw = c.get();
        }
        rl_bsdf_desc b;
        if (forced_bsdf) b = *forced_bsdf;
        else if (!make_bsdf(bs, &b)) return false;
        if (face_normals || !use_shading_normals) m.nrm.clear();
        // apply_transform (scene_loader.rs:341-378)
        const size_t nv = m.pos.size() / 3;
        const bool has_t = tw != nullptr;
        const Mat4 mat = has_t ? transform(*tw) : Mat4::identity();
        for (size_t i = 0; i < m.nrm.size() / 3; i++) {
            Vec3 n{m.nrm[3 * i], m.nrm[3 * i + 1], m.nrm[3 * i + 2]};
            if (has_t) n = mat.xform_vector(n);
            const float l = n.x * n.x + n.y * n.y + n.z * n.z;
            if (l != 0.0f && l != 1.0f) { const float s = std::sqrt(l); n = {n.x / s, n.y / s, n.z / s}; }
            m.nrm[3 * i] = n.x; m.nrm[3 * i + 1] = n.y; m.nrm[3 * i + 2] = n.z;
        }
        if (has_t) for (size_t i = 0; i < nv; i++) {
            Vec3 p = mat.xform_point({m.pos[3 * i], m.pos[3 * i + 1], m.pos[3 * i + 2]});
            m.pos[3 * i] = p.x; m.pos[3 * i + 1] = p.y; m.pos[3 * i + 2] = p.z;
        }
        float radiance[3]; bool emissive = false;
        if (em) { rgb_prop(*em, "radiance", 1.0f, radiance); emissive = true; }
        int rc = rl_scene_add_mesh(scene, m.pos.data(), nv, m.idx.data(), m.idx.size() / 3, m.nrm.empty() ? nullptr : m.nrm.data(),
                                   m.uv.empty() ? nullptr : m.uv.data(), &b, emissive ? radiance : nullptr);
        if (rc < 0) { err = "shape rejected by rl_scene_add_mesh"; return false; }
        return true;
    }
    std::string resolve(const std::string& f) const { return (!f.empty() && f[0] == '/') ? f : base_dir + "/" + f; }

    bool shape(const XmlNode& s) {
        const std::string ty = s.get("type");
        const bool face_normals = bool_prop(s, "faceNormals", false);
        if (ty == "ply") {
            LoadedMesh m;
            if (read_ply(resolve(string_prop(s, "filename", "")), &m, &err) != RL_OK) return false;
            return add_mesh(m, s, face_normals);
        }
        if (ty == "serialized") {
            LoadedMesh m;
            if (read_serialized(resolve(string_prop(s, "filename", "")), (int)float_prop(s, "shapeIndex", 0.0f), &m, &err) != RL_OK) return false;
            return add_mesh(m, s, face_normals);
        }
        if (ty == "obj") {
            std::vector<LoadedMesh> ms;
            if (read_obj(resolve(string_prop(s, "filename", "")), &ms, &err) != RL_OK) return false;
            const bool flip = bool_prop(s, "flipTexCoords", false);
            for (LoadedMesh& m : ms) {
                if (flip) for (size_t i = 1; i < m.uv.size(); i += 2) m.uv[i] = 1.0f - m.uv[i];
                if (!add_mesh(m, s, face_normals)) return false;      // the shape's bsdf replaces the MTL one (scene_loader.rs:452-461)
            }
            return true;
        }
        if (ty == "rectangle") {   // scene_loader.rs:537-556
            LoadedMesh m;
            m.pos = {-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0};
            m.uv = {0, 0, 1, 0, 1, 1, 0, 1};
            m.nrm = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
            m.idx = {0, 1, 2, 2, 3, 0};
            return add_mesh(m, s, false);
        }
        if (ty == "sphere") {      // scene_loader.rs:598-629: 32 x 32 UV sphere; host-only setup arithmetic
            float center[3] = {0, 0, 0};
            if (const XmlNode* c = child(s, "center")) {
                if (c->has("x") || c->has("y") || c->has("z")) { center[0] = attr_f(*c, "x", 0); center[1] = attr_f(*c, "y", 0); center[2] = attr_f(*c, "z", 0); }
                else { std::vector<float> v = numbers(value(*c)); if (v.size() >= 3) { center[0] = v[0]; center[1] = v[1]; center[2] = v[2]; } }
            }
            const float radius = float_prop(s, "radius", 1.0f), pi = 3.14159265358979323846f;
            LoadedMesh m;
            const int N = 32;
            for (int i = 0; i < N; i++) {
                const float theta = (float)i / (float)(N - 1) * pi;
                for (int j = 0; j < N; j++) {
                    const float phi = (float)j / (float)(N - 1) * 2.0f * pi;
                    const float x = radius * std::sin(theta) * std::cos(phi), y = radius * std::sin(theta) * std::sin(phi), z = radius * std::cos(theta);
                    m.pos.insert(m.pos.end(), {x + center[0], y + center[1], z + center[2]});
                    Vec3 n = vnormalize({x, y, z});
                    m.nrm.insert(m.nrm.end(), {n.x, n.y, n.z});
                    m.uv.insert(m.uv.end(), {theta / pi, phi / (2.0f * pi)});
                }
            }
            for (int i = 0; i < N - 1; i++)
                for (int j = 0; j < N - 1; j++) {
                    const uint32_t i0 = (uint32_t)(i * N + j), i1 = i0 + 1, i2 = (uint32_t)((i + 1) * N + j + 1), i3 = (uint32_t)((i + 1) * N + j);
                    m.idx.insert(m.idx.end(), {i0, i1, i2, i2, i3, i0});
                }
            return add_mesh(m, s, false);
        }
        return true;   // "Ignoring shape" (scene_loader.rs:666-669)
    }

    bool walk(const XmlNode& root) {
        // pass 1: defaults, includes and named bsdfs (a <ref> may precede its target in the file)
        for (const auto& c : root.children) {
            if (c->tag == "default") defaults[c->get("name")] = c->get("value");
            else if (c->tag == "bsdf" && c->has("id")) bsdf_by_id[c->get("id")] = c.get();
        }
        bool have_sensor = false;
        for (const auto& c : root.children) {
            if (c->tag == "include") {
                std::string src, file = resolve(value(*c, "filename"));
                std::ifstream f(file, std::ios::binary);
                if (!f) { err = "cannot open include " + file; return false; }
                std::stringstream ss; ss << f.rdbuf(); src = ss.str();
                XmlParser xp{src};
                std::unique_ptr<XmlNode> inc = xp.element();
                if (!inc) { err = file + ": " + xp.err; return false; }
                const XmlNode* keep = inc.get();
                included.push_back(std::move(inc));
                if (!walk(*keep)) return false;
            } else if (c->tag == "sensor") {
                if (have_sensor) { err = "more than one sensor"; return false; }   // assert_eq!(mts.sensors.len(), 1)
                have_sensor = true;
                uint32_t w = 768, h = 576;
                for (const auto& f : c->children) if (f->tag == "film") { w = (uint32_t)float_prop(*f, "width", 768.0f); h = (uint32_t)float_prop(*f, "height", 576.0f); }
                const float fov = float_prop(*c, "fov", 45.0f);
                const std::string axis = string_prop(*c, "fovAxis", "x");
                if (axis != "x" && axis != "y") { err = "Unsupport Fov axis definition: " + axis; return false; }
                Mat4 tw = Mat4::identity();
                for (const auto& t : c->children) if (t->tag == "transform" && t->get("name") == "toWorld") tw = transform(*t);
                float cols[16];
                tw.to_cols(cols);
                if (rl_scene_set_camera(scene, w, h, fov, axis == "y" ? 1 : 0, cols, 1 /* flip = true */) != RL_OK) { err = "invalid sensor"; return false; }
            } else if (c->tag == "shape") {
                if (!shape(*c)) return false;
            } else if (c->tag == "emitter") {
                if (c->get("type") == "point") {   // scene_loader.rs:680-697
                    float pos[3] = {0, 0, 0}, inten[3];
                    if (const XmlNode* p = child(*c, "position")) { pos[0] = attr_f(*p, "x", 0); pos[1] = attr_f(*p, "y", 0); pos[2] = attr_f(*p, "z", 0); }
                    for (const auto& t : c->children) if (t->tag == "transform" && t->get("name") == "toWorld") { Vec3 q = transform(*t).xform_point({pos[0], pos[1], pos[2]}); pos[0] = q.x; pos[1] = q.y; pos[2] = q.z; }
                    rgb_prop(*c, "intensity", 1.0f, inten);
                    rl_scene_add_point_light(scene, pos, inten);
                }   // other emitters: "Ignoring emitter"
            } else if (c->tag == "medium" && c->get("type") == "homogeneous" && !have_medium) {   // scene_loader.rs:734-780: the first medium
                have_medium = true;
                float ss[3], sa[3];
                rgb_prop(*c, "sigmaS", 1.0f, ss); rgb_prop(*c, "sigmaA", 1.0f, sa);
                const float sc = float_prop(*c, "scale", 1.0f);
                for (int k = 0; k < 3; k++) { ss[k] *= sc; sa[k] *= sc; }
                int phase = RL_PHASE_ISOTROPIC; float g = 0.0f;
                for (const auto& p : c->children) if (p->tag == "phase" && p->get("type") == "hg") { phase = RL_PHASE_HG; g = float_prop(*p, "g", 0.8f); }
                if (rl_scene_set_medium(scene, sa, ss, phase, g) != RL_OK) { err = "invalid medium"; return false; }
            }
        }
        if (&root == top && !have_sensor) { err = "no sensor in the scene"; return false; }
        return true;
    }
    bool have_medium = false;
    const XmlNode* top = nullptr;
};

}  // namespace

int load_mitsuba(const char* path, bool use_shading_normals, rl_scene** out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { *err = std::string("cannot open ") + path; return RL_ERR_IO; }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string src = ss.str();
    XmlParser xp{src};
    std::unique_ptr<XmlNode> root = xp.element();
    if (!root) { *err = std::string(path) + ": " + xp.err; return RL_ERR_PARSE; }
    if (root->tag != "scene") { *err = std::string(path) + ": root element is not <scene>"; return RL_ERR_PARSE; }
    Loader ld;
    const std::string p(path);
    const size_t slash = p.find_last_of('/');
    ld.base_dir = slash == std::string::npos ? std::string(".") : p.substr(0, slash);
    ld.use_shading_normals = use_shading_normals;
    ld.scene = new rl_scene();
    ld.top = root.get();
    if (!ld.walk(*root)) { *err = ld.err; delete ld.scene; return RL_ERR_PARSE; }
    *out = ld.scene;
    return RL_OK;
}

}  // namespace rl

extern "C" int rl_scene_load_mitsuba(const char* path, int use_shading_normals, rl_scene** out) {
    if (!path || !out) return RL_ERR_INVALID_ARGUMENT;
    std::string err;
    int rc;
    try { rc = rl::load_mitsuba(path, use_shading_normals != 0, out, &err); }
    catch (const std::exception& e) { rc = RL_ERR_PARSE; err = std::string(path) + ": malformed scene (" + e.what() + ")"; }   // nothing is thrown across the C ABI
    if (rc != RL_OK) rl_set_error(err);
    return rc;
}

// SceneLoaderManager::load (src/scene_loader.rs:27-58): the loader is chosen by the file extension
extern "C" int rl_scene_load(const char* path, int use_shading_normals, rl_scene** out) {
    if (!path || !out) return RL_ERR_INVALID_ARGUMENT;
    const std::string p(path);
    const size_t dot = p.find_last_of('.');
    const std::string ext = dot == std::string::npos ? std::string() : p.substr(dot + 1);
    if (ext == "pbrt") return rl_scene_load_pbrt(path, use_shading_normals, out);
    if (ext == "xml") return rl_scene_load_mitsuba(path, use_shading_normals, out);
    rl_set_error("Impossible to found scene loader for " + ext + " extension");
    return RL_ERR_UNSUPPORTED;
}
