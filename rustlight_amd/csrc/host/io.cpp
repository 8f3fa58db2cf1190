// io.cpp — image output and host-only debug hooks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../kernels/wavefront.h"
#include "meshio.h"
#include "scene.h"

// Bitmap::read_pfm (src/structure.rs:563-607): "PF\n", "W H\n", "-1.0\n" (little endian only), rows stored bottom-up
namespace rl {
int read_pfm(const char* path, uint32_t* w, uint32_t* h, std::vector<float>* rgb) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return RL_ERR_IO;
    char line[128];
    unsigned ww = 0, hh = 0; float enc = 0.0f;
    bool ok = std::fgets(line, sizeof(line), f) && std::strcmp(line, "PF\n") == 0
           && std::fgets(line, sizeof(line), f) && std::sscanf(line, "%u %u", &ww, &hh) == 2
           && std::fgets(line, sizeof(line), f) && std::sscanf(line, "%f", &enc) == 1 && enc == -1.0f && ww && hh && (uint64_t)ww * hh <= (1ull << 28);
    if (!ok) { std::fclose(f); return RL_ERR_PARSE; }
    rgb->assign((size_t)3 * ww * hh, 0.0f);
    std::vector<float> row(3 * (size_t)ww);
    for (unsigned y = 0; y < hh && ok; y++) {
        ok = std::fread(row.data(), sizeof(float), row.size(), f) == row.size();
        if (ok) std::memcpy(&(*rgb)[3 * (size_t)(hh - y - 1) * ww], row.data(), row.size() * sizeof(float));
    }
    std::fclose(f);
    if (!ok) return RL_ERR_IO;
    *w = ww; *h = hh;
    return RL_OK;
}
}  // namespace rl

using namespace rl;

extern "C" {

int rl_load_pfm(const char* path, uint32_t* width, uint32_t* height, float* rgb, size_t capacity_floats) {
    if (!path || !width || !height) return RL_ERR_INVALID_ARGUMENT;
    std::vector<float> data;
    int rc;
    try { rc = rl::read_pfm(path, width, height, &data); }
    catch (const std::exception&) { rc = RL_ERR_PARSE; }
    if (rc != RL_OK) return rc;
    if (rgb) {   // rgb == NULL: size query
        if (capacity_floats < data.size()) return RL_ERR_INVALID_ARGUMENT;
        std::memcpy(rgb, data.data(), data.size() * sizeof(float));
    }
    return RL_OK;
}

// Bitmap::read (src/structure.rs:670-683): .pfm or .png by extension, same calling convention as rl_load_pfm
int rl_load_image(const char* path, uint32_t* width, uint32_t* height, float* rgb, size_t capacity_floats) {
    if (!path || !width || !height) return RL_ERR_INVALID_ARGUMENT;
    rl::HostBitmap img;
    std::string err;
    int rc;
    try { rc = rl::read_image(path, &img, &err); }
    catch (const std::exception& e) { rc = RL_ERR_PARSE; err = std::string(path) + ": malformed image (" + e.what() + ")"; }
    if (rc != RL_OK) { rl_set_error(err); return rc; }
    *width = img.w; *height = img.h;
    if (rgb) {
        if (capacity_floats < img.rgb.size()) return RL_ERR_INVALID_ARGUMENT;
        std::memcpy(rgb, img.rgb.data(), img.rgb.size() * sizeof(float));
    }
    return RL_OK;
}

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

// ---- PNG (stored deflate blocks; no external zlib needed)
static uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
static void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static bool write_chunk(FILE* f, const char* type, const std::vector<unsigned char>& data) {
    std::vector<unsigned char> buf;
    put_be32(buf, (uint32_t)data.size());
    buf.insert(buf.end(), type, type + 4);
    buf.insert(buf.end(), data.begin(), data.end());
    uint32_t crc = crc32_update(0, buf.data() + 4, buf.size() - 4);
    put_be32(buf, crc);
    return std::fwrite(buf.data(), 1, buf.size(), f) == buf.size();
}
// Color::to_rgba (structure.rs:161-168): (c.min(1.0).powf(1/2.2) * 255.0) as u8 — `as u8` saturates, NaN -> 0
static unsigned char to_u8(float c) {
    float v = std::pow(std::fmin(c, 1.0f), 1.0f / 2.2f) * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (unsigned char)v;
}
int rl_save_png(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || width == 0 || height == 0) return RL_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RL_ERR_IO;
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    ok = ok && write_chunk(f, "IHDR", ihdr);
    std::vector<unsigned char> raw;
    raw.reserve((size_t)height * (3 * width + 1));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        for (uint32_t x = 0; x < 3 * width; x++) raw.push_back(to_u8(rgb[(size_t)3 * width * y + x]));
    }
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size(); off += 65535) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    }
    for (unsigned char c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    put_be32(z, (b << 16) | a);
    ok = ok && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", {});
    std::fclose(f);
    return ok ? RL_OK : RL_ERR_IO;
}

// ---- OpenEXR: single-part scanline file, NO_COMPRESSION, FLOAT channels (stored alphabetically: B, G, R)
int rl_save_exr(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || width == 0 || height == 0) return RL_ERR_INVALID_ARGUMENT;
    std::vector<unsigned char> h;
    auto put32 = [&](std::vector<unsigned char>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((x >> (8 * i)) & 0xff); };
    auto putf = [&](std::vector<unsigned char>& v, float x) { uint32_t u; std::memcpy(&u, &x, 4); put32(v, u); };
    auto puts = [&](std::vector<unsigned char>& v, const char* s) { while (*s) v.push_back((unsigned char)*s++); v.push_back(0); };
    auto attr = [&](const char* name, const char* type, const std::vector<unsigned char>& val) { puts(h, name); puts(h, type); put32(h, (uint32_t)val.size()); h.insert(h.end(), val.begin(), val.end()); };
    put32(h, 20000630u);   // magic
    put32(h, 2u);          // version 2, scanline, single part
    { std::vector<unsigned char> v; for (const char* c : {"B", "G", "R"}) { puts(v, c); put32(v, 2u /*FLOAT*/); v.push_back(0); v.push_back(0); v.push_back(0); v.push_back(0); put32(v, 1); put32(v, 1); } v.push_back(0); attr("channels", "chlist", v); }
    { std::vector<unsigned char> v; v.push_back(0); attr("compression", "compression", v); }
    { std::vector<unsigned char> v; put32(v, 0); put32(v, 0); put32(v, width - 1); put32(v, height - 1); attr("dataWindow", "box2i", v); attr("displayWindow", "box2i", v); }
    { std::vector<unsigned char> v; v.push_back(0); attr("lineOrder", "lineOrder", v); }
    { std::vector<unsigned char> v; putf(v, 1.0f); attr("pixelAspectRatio", "float", v); }
    { std::vector<unsigned char> v; putf(v, 0.0f); putf(v, 0.0f); attr("screenWindowCenter", "v2f", v); }
    { std::vector<unsigned char> v; putf(v, 1.0f); attr("screenWindowWidth", "float", v); }
    h.push_back(0);
    const uint64_t line_bytes = (uint64_t)width * 12;
    const uint64_t table_off = h.size();
    uint64_t off = table_off + (uint64_t)8 * height;
    for (uint32_t y = 0; y < height; y++) { uint64_t o = off + (uint64_t)y * (8 + line_bytes); for (int i = 0; i < 8; i++) h.push_back((o >> (8 * i)) & 0xff); }
    FILE* f = std::fopen(path, "wb");
    if (!f) return RL_ERR_IO;
    bool ok = std::fwrite(h.data(), 1, h.size(), f) == h.size();
    std::vector<unsigned char> line;
    for (uint32_t y = 0; y < height && ok; y++) {
        line.clear();
        put32(line, y); put32(line, (uint32_t)line_bytes);
        for (int c = 2; c >= 0; c--)       // B, G, R planes
            for (uint32_t x = 0; x < width; x++) putf(line, rgb[(size_t)3 * ((size_t)y * width + x) + c]);
        ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
    }
    std::fclose(f);
    return ok ? RL_OK : RL_ERR_IO;
}

int rl_save_image(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path) return RL_ERR_INVALID_ARGUMENT;
    const char* dot = std::strrchr(path, '.');
    if (!dot) return RL_ERR_INVALID_ARGUMENT;            // "No file extension provided"
    if (!std::strcmp(dot, ".pfm")) return rl_save_pfm(path, rgb, width, height);
    if (!std::strcmp(dot, ".png")) return rl_save_png(path, rgb, width, height);
    if (!std::strcmp(dot, ".exr")) return rl_save_exr(path, rgb, width, height);
    return RL_ERR_UNSUPPORTED;                            // "Unknow output file extension"
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

// EmitterSampler's cdf over flux().channel_max() (scene.rs:103-111), for loader / emitter tests
int rl_debug_emitters_cdf(const rl_scene* scene, uint64_t* n_entries, float* cdf) {
    if (!scene || !n_entries || !scene->emitters_built) return RL_ERR_INVALID_ARGUMENT;
    if (cdf) { if (*n_entries < scene->emitters_cdf.size()) return RL_ERR_INVALID_ARGUMENT; std::memcpy(cdf, scene->emitters_cdf.data(), scene->emitters_cdf.size() * sizeof(float)); }
    *n_entries = scene->emitters_cdf.size();
    return RL_OK;
}

// the `-x ats` light tree as built by rl_scene_build_emitters: 16 words per node (struct LightNode) + the light proxies
int rl_debug_ats(const rl_scene* scene, uint64_t* n_nodes, float* nodes16, uint64_t* n_lights, int32_t* light_emitter, int32_t* light_prim) {
    if (!scene || !n_nodes || !n_lights || !scene->emitters_built) return RL_ERR_INVALID_ARGUMENT;
    if (nodes16) std::memcpy(nodes16, scene->ats_nodes.data(), scene->ats_nodes.size() * sizeof(rl::LightNode));
    if (light_emitter && light_prim) for (size_t i = 0; i < scene->ats_light_emitter.size(); i++) { light_emitter[i] = scene->ats_light_emitter[i]; light_prim[i] = scene->ats_light_prim[i]; }
    *n_nodes = scene->ats_nodes.size(); *n_lights = scene->ats_light_emitter.size();
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
