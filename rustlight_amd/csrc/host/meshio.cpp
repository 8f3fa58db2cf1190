// meshio.cpp — mesh and image file readers behind the scene loaders.
//
// rustlight delegates these formats to un-vendored crates (tobj 4 for OBJ, pbrt_rs::ply / mitsuba_rs::ply for PLY,
// mitsuba_rs::serialized, image for LDR bitmaps; Cargo.toml:19-30), so each reader restates the published format and
// produces what the reference's call sites consume (geometry.rs:13-96, scene_loader.rs:88-93, 392-440, 497-535,
// structure.rs:563-683):
//   * OBJ   — `o`/`g` (and a material change) start a model, polygons are fan-triangulated, vertices are re-indexed
//             per model in order of first use; the MTL gives `Kd` / `map_Kd` for a BSDFDiffuse (missing -> black);
//   * PLY   — ascii / binary_little_endian / binary_big_endian; vertex x y z [nx ny nz] [u v | s t], face lists
//             (fan-triangulated);
//   * serialized — Mitsuba 0.5 `.serialized` (format 0x041C, versions 3 and 4, zlib stream per mesh, offset table);
//   * images — .pfm (Bitmap::read_pfm) and 8/16-bit non-interlaced .png (value / 255, no gamma: read_ldr_image).
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "meshio.h"

namespace rl {
namespace {

bool slurp(const std::string& path, std::string* out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    *out = ss.str();
    return true;
}
std::string dir_of(const std::string& path) {
    size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string(".") : path.substr(0, s);
}

// ---------------------------------------------------------------------------------------------- OBJ / MTL
struct MtlEntry { bool has_kd = false; float kd[3] = {0, 0, 0}; std::string map_kd; };

void read_mtl(const std::string& path, std::map<std::string, MtlEntry>* out) {
    std::string src;
    if (!slurp(path, &src)) return;
    std::istringstream is(src);
    std::string line, cur;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key) || key[0] == '#') continue;
        if (key == "newmtl") { ls >> cur; (*out)[cur] = MtlEntry(); }
        else if (key == "Kd" && !cur.empty()) { MtlEntry& m = (*out)[cur]; if (ls >> m.kd[0] >> m.kd[1] >> m.kd[2]) m.has_kd = true; }
        else if (key == "map_Kd" && !cur.empty()) { std::string rest; std::getline(ls, rest); size_t b = rest.find_first_not_of(" \t"); size_t e = rest.find_last_not_of(" \t\r"); if (b != std::string::npos) (*out)[cur].map_kd = rest.substr(b, e - b + 1); }
    }
}

struct ObjBuilder {
    LoadedMesh mesh;
    std::map<std::tuple<int, int, int>, uint32_t> remap;
    bool any_vt = false, any_vn = false, all_vt = true, all_vn = true;
};

}  // namespace

int read_obj(const std::string& path, std::vector<LoadedMesh>* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    std::vector<float> V, VT, VN;
    std::map<std::string, MtlEntry> mtl;
    std::vector<ObjBuilder> models;
    std::string cur_name = "unnamed_object", cur_mtl;
    bool cur_has_mtl = false;
    auto start_model = [&]() {
        models.emplace_back();
        models.back().mesh.name = cur_name;
        if (cur_has_mtl) {
            auto it = mtl.find(cur_mtl);
            LoadedMesh& m = models.back().mesh;
            m.has_material = true;
            if (it != mtl.end()) { m.has_kd = it->second.has_kd; std::memcpy(m.kd, it->second.kd, sizeof(m.kd)); if (!it->second.map_kd.empty()) m.kd_map = dir_of(path) + "/" + it->second.map_kd; }
        }
    };
    std::istringstream is(src);
    std::string line;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key) || key[0] == '#') continue;
        if (key == "v") { float a, b, c; ls >> a >> b >> c; V.insert(V.end(), {a, b, c}); }
        else if (key == "vt") { float a = 0, b = 0; ls >> a >> b; VT.insert(VT.end(), {a, b}); }
        else if (key == "vn") { float a, b, c; ls >> a >> b >> c; VN.insert(VN.end(), {a, b, c}); }
        else if (key == "mtllib") { std::string f; ls >> f; read_mtl(dir_of(path) + "/" + f, &mtl); }
        else if (key == "o" || key == "g") {
            std::string n; std::getline(ls, n);
            size_t b = n.find_first_not_of(" \t"), e = n.find_last_not_of(" \t\r");
            cur_name = b == std::string::npos ? std::string("unnamed_object") : n.substr(b, e - b + 1);
            if (models.empty() || !models.back().mesh.idx.empty()) start_model(); else models.back().mesh.name = cur_name;
        } else if (key == "usemtl") {
            ls >> cur_mtl; cur_has_mtl = true;
            if (models.empty() || !models.back().mesh.idx.empty()) start_model();
            else { models.pop_back(); start_model(); }
        } else if (key == "f") {
            if (models.empty()) start_model();
            ObjBuilder& mb = models.back();
            std::vector<uint32_t> poly;
            std::string tok;
            while (ls >> tok) {
                int vi = 0, ti = 0, ni = 0;
                const char* p = tok.c_str();
                vi = (int)std::strtol(p, const_cast<char**>(&p), 10);
                if (*p == '/') { p++; if (*p != '/') ti = (int)std::strtol(p, const_cast<char**>(&p), 10); if (*p == '/') { p++; ni = (int)std::strtol(p, const_cast<char**>(&p), 10); } }
                if (vi < 0) vi = (int)(V.size() / 3) + vi + 1;
                if (ti < 0) ti = (int)(VT.size() / 2) + ti + 1;
                if (ni < 0) ni = (int)(VN.size() / 3) + ni + 1;
                if (vi <= 0 || (size_t)vi > V.size() / 3 || (size_t)ti > VT.size() / 2 || (size_t)ni > VN.size() / 3) { *err = "OBJ index out of range in " + path; return RL_ERR_PARSE; }
                auto key3 = std::make_tuple(vi, ti, ni);
                auto it = mb.remap.find(key3);
                uint32_t id;
                if (it == mb.remap.end()) {
                    id = (uint32_t)(mb.mesh.pos.size() / 3);
                    mb.remap[key3] = id;
                    mb.mesh.pos.insert(mb.mesh.pos.end(), {V[3 * (vi - 1)], V[3 * (vi - 1) + 1], V[3 * (vi - 1) + 2]});
                    if (ti) { mb.any_vt = true; mb.mesh.uv.insert(mb.mesh.uv.end(), {VT[2 * (ti - 1)], VT[2 * (ti - 1) + 1]}); } else { mb.all_vt = false; mb.mesh.uv.insert(mb.mesh.uv.end(), {0.0f, 0.0f}); }
                    if (ni) { mb.any_vn = true; mb.mesh.nrm.insert(mb.mesh.nrm.end(), {VN[3 * (ni - 1)], VN[3 * (ni - 1) + 1], VN[3 * (ni - 1) + 2]}); } else { mb.all_vn = false; mb.mesh.nrm.insert(mb.mesh.nrm.end(), {0.0f, 0.0f, 0.0f}); }
                } else id = it->second;
                poly.push_back(id);
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) mb.mesh.idx.insert(mb.mesh.idx.end(), {poly[0], poly[k], poly[k + 1]});
        }
    }
    for (ObjBuilder& mb : models) {
        if (mb.mesh.idx.empty()) continue;
        if (!(mb.any_vt && mb.all_vt)) mb.mesh.uv.clear();
        if (!(mb.any_vn && mb.all_vn)) mb.mesh.nrm.clear();
        out->push_back(std::move(mb.mesh));
    }
    if (out->empty()) { *err = "no faces in " + path; return RL_ERR_PARSE; }
    return RL_OK;
}

// ---------------------------------------------------------------------------------------------- PLY
namespace {
struct PlyProp { std::string name; int type = 0; bool is_list = false; int count_type = 0; };   // type: bytes, sign encodes float
int ply_type(const std::string& t) {   // size in bytes; floats are negative
    if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
    if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
    if (t == "int" || t == "int32" || t == "uint" || t == "uint32") return 4;
    if (t == "float" || t == "float32") return -4;
    if (t == "double" || t == "float64") return -8;
    return 0;
}
bool ply_signed(const std::string& t) { return t == "char" || t == "int8" || t == "short" || t == "int16" || t == "int" || t == "int32"; }
struct PlyReader {
    const std::string& src; size_t pos; int fmt;   // 0 ascii, 1 little, 2 big
    bool ok = true;
    double read(int type, bool sgn) {
        if (fmt == 0) {
            while (pos < src.size() && std::isspace((unsigned char)src[pos])) pos++;
            char* e = nullptr;
            double v = std::strtod(src.c_str() + pos, &e);
            if (e == src.c_str() + pos) { ok = false; return 0.0; }
            pos = (size_t)(e - src.c_str());
            return v;
        }
        int n = type < 0 ? -type : type;
        if (pos + (size_t)n > src.size()) { ok = false; return 0.0; }
        unsigned char b[8];
        for (int i = 0; i < n; i++) b[i] = (unsigned char)src[pos + (fmt == 1 ? i : n - 1 - i)];
        pos += (size_t)n;
        if (type == -4) { float f; std::memcpy(&f, b, 4); return f; }
        if (type == -8) { double d; std::memcpy(&d, b, 8); return d; }
        unsigned long long u = 0;
        for (int i = n - 1; i >= 0; i--) u = (u << 8) | b[i];
        if (sgn) { if (n == 1) return (double)(signed char)u; if (n == 2) return (double)(short)u; return (double)(int)u; }
        return (double)u;
    }
};
}  // namespace

int read_ply(const std::string& path, LoadedMesh* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    size_t pos = 0;
    auto next_line = [&](std::string* l) { size_t e = src.find('\n', pos); if (e == std::string::npos) return false; *l = src.substr(pos, e - pos); if (!l->empty() && l->back() == '\r') l->pop_back(); pos = e + 1; return true; };
    std::string line;
    if (!next_line(&line) || line != "ply") { *err = path + ": not a PLY file"; return RL_ERR_PARSE; }
    struct Elem { std::string name; size_t count; std::vector<PlyProp> props; std::vector<bool> sgn; };
    std::vector<Elem> elems;
    int fmt = -1;
    while (next_line(&line)) {
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "format") { std::string f; ls >> f; fmt = f == "ascii" ? 0 : (f == "binary_little_endian" ? 1 : (f == "binary_big_endian" ? 2 : -1)); }
        else if (key == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
        else if (key == "property" && !elems.empty()) {
            PlyProp p; std::string t; ls >> t;
            bool sg;
            if (t == "list") { std::string ct, it; ls >> ct >> it >> p.name; p.is_list = true; p.count_type = ply_type(ct); p.type = ply_type(it); sg = ply_signed(it); }
            else { ls >> p.name; p.type = ply_type(t); sg = ply_signed(t); }
            if (p.type == 0) { *err = path + ": unknown PLY property type"; return RL_ERR_PARSE; }
            elems.back().props.push_back(p); elems.back().sgn.push_back(sg);
        } else if (key == "end_header") break;
    }
    if (fmt < 0) { *err = path + ": unsupported PLY format"; return RL_ERR_PARSE; }
    PlyReader rd{src, pos, fmt};
    out->name = "";
    for (const Elem& e : elems) {
        int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
        for (size_t k = 0; k < e.props.size(); k++) {
            const std::string& n = e.props[k].name;
            if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
            else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
            else if (n == "u" || n == "s") iu = (int)k; else if (n == "v" || n == "t") iv = (int)k;
        }
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        if (is_vertex && (ix < 0 || iy < 0 || iz < 0)) { *err = path + ": vertex element without x y z"; return RL_ERR_PARSE; }
        std::vector<double> vals(e.props.size());
        for (size_t i = 0; i < e.count && rd.ok; i++) {
            for (size_t k = 0; k < e.props.size() && rd.ok; k++) {
                const PlyProp& p = e.props[k];
                if (!p.is_list) { vals[k] = rd.read(p.type, e.sgn[k]); continue; }
                const double cnt = rd.read(p.count_type, false);
                if (!(cnt >= 0.0 && cnt <= 65536.0)) { *err = path + ": PLY list length out of range"; return RL_ERR_PARSE; }
                size_t n = (size_t)cnt;
                std::vector<uint32_t> poly(n);
                for (size_t j = 0; j < n && rd.ok; j++) poly[j] = (uint32_t)rd.read(p.type, e.sgn[k]);
                if (is_face && (p.name == "vertex_indices" || p.name == "vertex_index"))
                    for (size_t j = 1; j + 1 < n; j++) out->idx.insert(out->idx.end(), {poly[0], poly[j], poly[j + 1]});
            }
            if (is_vertex) {
                out->pos.insert(out->pos.end(), {(float)vals[ix], (float)vals[iy], (float)vals[iz]});
                if (inx >= 0 && iny >= 0 && inz >= 0) out->nrm.insert(out->nrm.end(), {(float)vals[inx], (float)vals[iny], (float)vals[inz]});
                if (iu >= 0 && iv >= 0) out->uv.insert(out->uv.end(), {(float)vals[iu], (float)vals[iv]});
            }
        }
    }
    if (!rd.ok || out->pos.empty() || out->idx.empty()) { *err = path + ": truncated or empty PLY"; return RL_ERR_PARSE; }
    for (uint32_t i : out->idx) if (i >= out->pos.size() / 3) { *err = path + ": PLY face index out of range"; return RL_ERR_PARSE; }
    return RL_OK;
}

// ---------------------------------------------------------------------------------------------- zlib helper
namespace {
bool inflate_all(const unsigned char* data, size_t n, std::vector<unsigned char>* out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char*>(data);
    zs.avail_in = (uInt)n;
    unsigned char buf[1 << 16];
    int rc;
    do {
        zs.next_out = buf; zs.avail_out = sizeof(buf);
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return false; }
        out->insert(out->end(), buf, buf + (sizeof(buf) - zs.avail_out));
    } while (rc != Z_STREAM_END);
    inflateEnd(&zs);
    return true;
}
}  // namespace

// ---------------------------------------------------------------------------------------------- Mitsuba .serialized
int read_serialized(const std::string& path, int shape_index, LoadedMesh* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    auto u16 = [&](size_t o) { return (unsigned)(unsigned char)src[o] | ((unsigned)(unsigned char)src[o + 1] << 8); };
    auto rd = [&](size_t o, int n) { unsigned long long v = 0; for (int i = n - 1; i >= 0; i--) v = (v << 8) | (unsigned char)src[o + i]; return v; };
    if (src.size() < 8 || u16(0) != 0x041C) { *err = path + ": not a Mitsuba serialized file"; return RL_ERR_PARSE; }
    const unsigned version = u16(2);
    if (version != 3 && version != 4) { *err = path + ": unsupported serialized version"; return RL_ERR_PARSE; }
    const unsigned n_meshes = (unsigned)rd(src.size() - 4, 4);
    const int osz = version == 4 ? 8 : 4;
    if (shape_index < 0 || (unsigned)shape_index >= n_meshes || src.size() < 4 + (size_t)osz * n_meshes) { *err = path + ": shape index out of range"; return RL_ERR_PARSE; }
    const size_t table = src.size() - 4 - (size_t)osz * n_meshes;
    const size_t begin = (size_t)rd(table + (size_t)osz * shape_index, osz);
    const size_t end = (unsigned)shape_index + 1 < n_meshes ? (size_t)rd(table + (size_t)osz * (shape_index + 1), osz) : table;
    if (begin + 4 > end || end > src.size() || u16(begin) != 0x041C) { *err = path + ": corrupt offset table"; return RL_ERR_PARSE; }
    std::vector<unsigned char> raw;
    if (!inflate_all(reinterpret_cast<const unsigned char*>(src.data()) + begin + 4, end - begin - 4, &raw)) { *err = path + ": zlib stream error"; return RL_ERR_PARSE; }
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= raw.size(); };
    if (!need(4)) { *err = path + ": truncated mesh"; return RL_ERR_PARSE; }
    unsigned flags; std::memcpy(&flags, &raw[p], 4); p += 4;
    if (version == 4) { while (p < raw.size() && raw[p]) out->name.push_back((char)raw[p++]); p++; }
    if (!need(16)) { *err = path + ": truncated mesh"; return RL_ERR_PARSE; }
    unsigned long long nv, nt; std::memcpy(&nv, &raw[p], 8); std::memcpy(&nt, &raw[p + 8], 8); p += 16;
    if (nv > (1ull << 31) || nt > (1ull << 31)) { *err = path + ": implausible mesh size"; return RL_ERR_PARSE; }
    const bool dbl = (flags & 0x2000u) != 0;
    const size_t fs = dbl ? 8 : 4;
    auto read_floats = [&](size_t count, std::vector<float>* dst) {
        if (!need(count * fs)) return false;
        dst->resize(count);
        for (size_t i = 0; i < count; i++) { if (dbl) { double d; std::memcpy(&d, &raw[p + i * 8], 8); (*dst)[i] = (float)d; } else std::memcpy(&(*dst)[i], &raw[p + i * 4], 4); }
        p += count * fs;
        return true;
    };
    std::vector<float> skip;
    bool ok = read_floats((size_t)nv * 3, &out->pos);
    if (ok && (flags & 0x0001u)) ok = read_floats((size_t)nv * 3, &out->nrm);
    if (ok && (flags & 0x0002u)) ok = read_floats((size_t)nv * 2, &out->uv);
    if (ok && (flags & 0x0008u)) ok = read_floats((size_t)nv * 3, &skip);
    const size_t is = nv > 0xFFFFFFFFull ? 8 : 4;
    if (!ok || !need((size_t)nt * 3 * is)) { *err = path + ": truncated mesh"; return RL_ERR_PARSE; }
    out->idx.resize((size_t)nt * 3);
    for (size_t i = 0; i < (size_t)nt * 3; i++) { unsigned long long v = 0; std::memcpy(&v, &raw[p + i * is], is); if (v >= nv) { *err = path + ": index out of range"; return RL_ERR_PARSE; } out->idx[i] = (uint32_t)v; }
    return RL_OK;
}

// ---------------------------------------------------------------------------------------------- images
namespace {
int read_png(const std::string& path, HostBitmap* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (src.size() < 8 || std::memcmp(src.data(), sig, 8) != 0) { *err = path + ": not a PNG"; return RL_ERR_PARSE; }
    auto be32 = [&](size_t o) { return ((unsigned)(unsigned char)src[o] << 24) | ((unsigned)(unsigned char)src[o + 1] << 16) | ((unsigned)(unsigned char)src[o + 2] << 8) | (unsigned)(unsigned char)src[o + 3]; };
    unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, palette;
    for (size_t p = 8; p + 12 <= src.size();) {
        unsigned len = be32(p);
        std::string type = src.substr(p + 4, 4);
        if (p + 12 + len > src.size()) break;
        const unsigned char* d = reinterpret_cast<const unsigned char*>(src.data()) + p + 8;
        if (type == "IHDR" && len >= 13) { w = be32(p + 8); h = be32(p + 12); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (type == "PLTE") palette.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        p += 12 + len;
    }
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || (uint64_t)w * h > (1ull << 28) || !channels || interlace || (depth != 8 && depth != 16) || (ctype == 3 && depth != 8)) { *err = path + ": unsupported PNG variant (8/16-bit non-interlaced only)"; return RL_ERR_UNSUPPORTED; }
    std::vector<unsigned char> raw;
    if (!inflate_all(idat.data(), idat.size(), &raw)) { *err = path + ": zlib stream error"; return RL_ERR_PARSE; }
    const size_t bpp = (size_t)channels * depth / 8, stride = bpp * w;
    if (raw.size() < (stride + 1) * h) { *err = path + ": truncated PNG"; return RL_ERR_PARSE; }
    std::vector<unsigned char> img(stride * h);
    for (unsigned y = 0; y < h; y++) {
        const unsigned char* in = &raw[(stride + 1) * y];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = in[1 + x];
            switch (in[0]) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: { int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: break;
            }
            cur[x] = (unsigned char)v;
        }
    }
    out->w = w; out->h = h;
    out->rgb.assign((size_t)3 * w * h, 0.0f);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        unsigned char px[4] = {0, 0, 0, 255};
        for (int c = 0; c < channels; c++) px[c] = img[i * bpp + (size_t)c * depth / 8];   // 16-bit: the high byte, as image::to_rgb8 does
        unsigned char r, g, b;
        if (ctype == 3) { size_t k = (size_t)px[0] * 3; r = k + 2 < palette.size() ? palette[k] : 0; g = k + 2 < palette.size() ? palette[k + 1] : 0; b = k + 2 < palette.size() ? palette[k + 2] : 0; }
        else if (channels <= 2) r = g = b = px[0];
        else { r = px[0]; g = px[1]; b = px[2]; }
        out->rgb[3 * i] = (float)r / 255.0f; out->rgb[3 * i + 1] = (float)g / 255.0f; out->rgb[3 * i + 2] = (float)b / 255.0f;   // read_ldr_image
    }
    return RL_OK;
}

// OpenEXR (Bitmap::read_exr, structure.rs:607-640: the R, G, B channels as f32): single-part scanline files with NONE, RLE, ZIPS or
// ZIP compression and HALF / FLOAT / UINT channels.  Tiled, multi-part, deep and PIZ / PXR24 / B44 / DWA files are refused.
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; uint32_t m = man; do { e++; m <<= 1; } while (!(m & 0x400)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ff) << 13); }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}

int read_exr(const std::string& path, HostBitmap* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    const unsigned char* d = reinterpret_cast<const unsigned char*>(src.data());
    const size_t n = src.size();
    size_t p = 0;
    auto bad = [&](const char* what) { *err = path + ": " + what; return RL_ERR_PARSE; };
    auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, d + o, 4); return v; };
    if (n < 8 || u32(0) != 20000630u) return bad("not an OpenEXR file");
    const uint32_t version = u32(4);
    if ((version & 0xff) != 2 || (version & 0x1a00)) { *err = path + ": tiled / multi-part / deep OpenEXR files are not read"; return RL_ERR_UNSUPPORTED; }
    p = 8;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1; int32_t win[4] = {0, 0, -1, -1}; bool have_win = false;
    for (;;) {     // attributes: name\0 type\0 size value
        if (p >= n) return bad("truncated header");
        if (d[p] == 0) { p++; break; }
        std::string name, type;
        while (p < n && d[p]) name.push_back((char)d[p++]);
        p++;
        while (p < n && d[p]) type.push_back((char)d[p++]);
        p++;
        if (p + 4 > n) return bad("truncated header");
        const uint32_t size = u32(p); p += 4;
        if (size > n || p + size > n) return bad("truncated header");
        if (name == "channels") {
            size_t q = p;
            while (q < p + size && d[q]) {
                Chan c;
                while (q < p + size && d[q]) c.name.push_back((char)d[q++]);
                q++;
                if (q + 16 > p + size) return bad("bad channel list");
                c.type = (int)u32(q);
                if (u32(q + 8) != 1 || u32(q + 12) != 1) { *err = path + ": subsampled OpenEXR channels are not read"; return RL_ERR_UNSUPPORTED; }
                if (c.type < 0 || c.type > 2) return bad("bad channel type");
                q += 16;
                chans.push_back(c);
            }
        } else if (name == "compression" && size >= 1) compression = d[p];
        else if (name == "dataWindow" && size >= 16) { std::memcpy(win, d + p, 16); have_win = true; }
        p += size;
    }
    if (!have_win || chans.empty() || compression < 0) return bad("missing channels / compression / dataWindow");
    if (compression > 3) { *err = path + ": only NONE / RLE / ZIPS / ZIP compressed OpenEXR files are read (no PIZ / PXR24 / B44 / DWA decoder)"; return RL_ERR_UNSUPPORTED; }
    const int64_t w = (int64_t)win[2] - win[0] + 1, h = (int64_t)win[3] - win[1] + 1;
    if (w <= 0 || h <= 0 || w * h > (1ll << 28)) return bad("bad dataWindow");
    int idx[3] = {-1, -1, -1};
    size_t line_bytes = 0;
    std::vector<size_t> chan_off(chans.size());
    for (size_t c = 0; c < chans.size(); c++) {
        chan_off[c] = line_bytes;
        line_bytes += (size_t)w * (chans[c].type == 1 ? 2 : 4);
        if (chans[c].name == "R") idx[0] = (int)c; else if (chans[c].name == "G") idx[1] = (int)c; else if (chans[c].name == "B") idx[2] = (int)c;
    }
    if (idx[0] < 0 && idx[1] < 0 && idx[2] < 0) {      // luminance-only file: Y feeds all three
        for (size_t c = 0; c < chans.size(); c++) if (chans[c].name == "Y") idx[0] = idx[1] = idx[2] = (int)c;
        if (idx[0] < 0) return bad("no R / G / B channel");
    }
    const int lines_per_chunk = compression == 3 ? 16 : 1;
    const size_t n_chunks = (size_t)((h + lines_per_chunk - 1) / lines_per_chunk);
    if (p + 8 * n_chunks > n) return bad("truncated offset table");
    out->w = (uint32_t)w; out->h = (uint32_t)h;
    out->rgb.assign((size_t)3 * w * h, 0.0f);
    std::vector<unsigned char> buf, tmp;
    for (size_t k = 0; k < n_chunks; k++) {
        uint64_t off; std::memcpy(&off, d + p + 8 * k, 8);
        if (off + 8 > n) return bad("chunk offset out of range");
        int32_t y0; std::memcpy(&y0, d + off, 4);
        const uint32_t size = u32(off + 4);
        if (off + 8 + size > n) return bad("truncated chunk");
        const int64_t row0 = (int64_t)y0 - win[1];
        if (row0 < 0 || row0 >= h) return bad("chunk outside the dataWindow");
        const int64_t rows = std::min<int64_t>(lines_per_chunk, h - row0);
        const size_t want = line_bytes * (size_t)rows;
        const unsigned char* data = d + off + 8;
        if (compression != 0 && size < want) {
            tmp.clear();
            if (compression == 1) {     // RLE: signed run lengths
                for (size_t i = 0; i < size;) {
                    const int c = (signed char)data[i++];
                    if (c < 0) { const size_t m = (size_t)(-c); if (i + m > size) return bad("bad RLE run"); tmp.insert(tmp.end(), data + i, data + i + m); i += m; }
                    else { if (i >= size) return bad("bad RLE run"); tmp.insert(tmp.end(), (size_t)c + 1, data[i++]); }
                    if (tmp.size() > want) return bad("RLE chunk too long");
                }
            } else if (!inflate_all(data, size, &tmp)) return bad("zlib stream error");
            if (tmp.size() != want) return bad("chunk has the wrong size");
            for (size_t i = 1; i < want; i++) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);     // predictor
            buf.resize(want);
            const size_t half = (want + 1) / 2;
            for (size_t i = 0; i < want; i++) buf[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];     // de-interleave the two halves
            data = buf.data();
        } else if (size != want) return bad("chunk has the wrong size");
        for (int64_t r = 0; r < rows; r++)
            for (int c = 0; c < 3; c++) {
                if (idx[c] < 0) continue;
                const unsigned char* q = data + (size_t)r * line_bytes + chan_off[idx[c]];
                const int type = chans[idx[c]].type;
                for (int64_t x = 0; x < w; x++) {
                    float v;
                    if (type == 1) { uint16_t hv; std::memcpy(&hv, q + 2 * x, 2); v = half_to_float(hv); }
                    else if (type == 2) std::memcpy(&v, q + 4 * x, 4);
                    else { uint32_t uv; std::memcpy(&uv, q + 4 * x, 4); v = (float)uv; }
                    out->rgb[(size_t)3 * ((size_t)(row0 + r) * w + x) + c] = v;
                }
            }
    }
    return RL_OK;
}

// JPEG (read_ldr_image's `image::open` for .jpg / .jpeg textures): baseline / extended sequential and progressive Huffman, 8-bit, grey or YCbCr with
// any sampling factors, restart intervals; libjpeg's accurate integer IDCT ("islow") and triangle-filter ("fancy") chroma upsampling
// for 2:1 factors, which is also what the reference's decoder (jpeg-decoder) does — decoders are only required to agree to +-1 level,
// so textures read from a JPEG are the one input where bit-identity with the reference is not defined.  Arithmetic-coded, lossless,
// hierarchical and 12-bit files are refused.
struct JpegDecoder {
    const unsigned char* d; size_t n, p = 0;
    std::string err;
    struct Huff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool set = false; };
    struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; int pw = 0, ph = 0; };
    uint16_t qt[4][64]; bool qt_set[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    int width = 0, height = 0, restart = 0;
    bool progressive = false; int eobrun = 0;
    std::vector<std::vector<int16_t>> coefs;      // progressive: [component][block * 64 + natural index], MCU-padded block grid
    uint32_t bitbuf = 0; int bitcnt = 0; bool hit_marker = false;

    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    int u8() { return p < n ? d[p++] : -1; }
    int u16() { int a = u8(), b = u8(); return (a < 0 || b < 0) ? -1 : (a << 8) | b; }

    void build(Huff& h) {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            h.valptr[l] = k; h.mincode[l] = code;
            code += h.bits[l]; k += h.bits[l];
            h.maxcode[l] = h.bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        h.maxcode[17] = 0x7fffffff; h.set = true;
    }
    int bit() {
        if (bitcnt == 0) {
            int c = 0;
            if (!hit_marker) {
                c = u8();
                if (c < 0) c = 0;
                if (c == 0xff) {
                    int c2 = u8();
                    if (c2 != 0) { hit_marker = true; p -= 2; c = 0; }     // a marker: feed zeros until the caller handles it
                }
            }
            bitbuf = (uint32_t)c; bitcnt = 8;
        }
        bitcnt--;
        return (bitbuf >> bitcnt) & 1;
    }
    int receive(int s) { int v = 0; for (int i = 0; i < s; i++) v = (v << 1) | bit(); return v; }
    static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
    int decode(const Huff& h) {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | bit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return -1;
    }
    // libjpeg jidctint.c (accurate integer inverse DCT), CONST_BITS = 13, PASS1_BITS = 2
    static void idct(const int* in, uint8_t* out, int stride) {
        const int C0298 = 2446, C0390 = 3196, C0541 = 4433, C0765 = 6270, C0899 = 7373, C1175 = 9633, C1501 = 12299, C1847 = 15137, C1961 = 16069, C2053 = 16819, C2562 = 20995, C3072 = 25172;
        long ws[64];
        for (int c = 0; c < 8; c++) {
            const int* i = in + c;
            long z2 = i[16], z3 = i[48];
            long z1 = (z2 + z3) * C0541;
            long t2 = z1 + z3 * (-C1847), t3 = z1 + z2 * C0765;
            z2 = i[0]; z3 = i[32];
            long t0 = (z2 + z3) * 8192, t1 = (z2 - z3) * 8192;
            long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            t0 = i[56]; t1 = i[40]; t2 = i[24]; t3 = i[8];
            z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; long z4 = t1 + t3, z5 = (z3 + z4) * C1175;
            t0 *= C0298; t1 *= C2053; t2 *= C3072; t3 *= C1501;
            z1 *= -C0899; z2 *= -C2562; z3 *= -C1961; z4 *= -C0390;
            z3 += z5; z4 += z5;
            t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
            auto ds = [](long x) { return (x + (1L << 10)) >> 11; };
            ws[c] = ds(t10 + t3); ws[56 + c] = ds(t10 - t3); ws[8 + c] = ds(t11 + t2); ws[48 + c] = ds(t11 - t2);
            ws[16 + c] = ds(t12 + t1); ws[40 + c] = ds(t12 - t1); ws[24 + c] = ds(t13 + t0); ws[32 + c] = ds(t13 - t0);
        }
        for (int r = 0; r < 8; r++) {
            const long* w = ws + 8 * r;
            long z2 = w[2], z3 = w[6];
            long z1 = (z2 + z3) * C0541;
            long t2 = z1 + z3 * (-C1847), t3 = z1 + z2 * C0765;
            long t0 = (w[0] + w[4]) * 8192, t1 = (w[0] - w[4]) * 8192;
            long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
            z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; long z4 = t1 + t3, z5 = (z3 + z4) * C1175;
            t0 *= C0298; t1 *= C2053; t2 *= C3072; t3 *= C1501;
            z1 *= -C0899; z2 *= -C2562; z3 *= -C1961; z4 *= -C0390;
            z3 += z5; z4 += z5;
            t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
            auto px = [](long x) { long v = ((x + (1L << 17)) >> 18) + 128; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
            uint8_t* o = out + (size_t)r * stride;
            o[0] = px(t10 + t3); o[7] = px(t10 - t3); o[1] = px(t11 + t2); o[6] = px(t11 - t2);
            o[2] = px(t12 + t1); o[5] = px(t12 - t1); o[3] = px(t13 + t0); o[4] = px(t13 - t0);
        }
    }
    bool block(Comp& c, int bx, int by) {
        static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                       29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        int coef[64] = {0};
        int t = decode(dc[c.td]);
        if (t < 0 || t > 11) return fail("bad Huffman code (DC)");
        c.pred += extend(receive(t), t);
        coef[0] = c.pred * qt[c.tq][0];
        for (int k = 1; k < 64;) {
            int rs = decode(ac[c.ta]);
            if (rs < 0) return fail("bad Huffman code (AC)");
            int r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) return fail("AC run past the block");
            coef[zz[k]] = extend(receive(s), s) * qt[c.tq][k];
            k++;
        }
        idct(coef, &c.plane[(size_t)by * 8 * c.pw + (size_t)bx * 8], c.pw);
        return true;
    }
    bool scan() {
        int hmax = 1, vmax = 1;
        for (Comp& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (Comp& c : comps) {
            if (!qt_set[c.tq] || !dc[c.td].set || !ac[c.ta].set) return fail("missing quantisation / Huffman table");
            c.pw = mcux * c.h * 8; c.ph = mcuy * c.v * 8; c.plane.assign((size_t)c.pw * c.ph, 0); c.pred = 0;
        }
        int left = restart;
        bitcnt = 0; hit_marker = false;
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                if (restart && left == 0) {
                    bitcnt = 0; hit_marker = false;
                    if (p + 2 <= n && d[p] == 0xff && d[p + 1] >= 0xd0 && d[p + 1] <= 0xd7) p += 2; else return fail("missing restart marker");
                    for (Comp& c : comps) c.pred = 0;
                    left = restart;
                }
                for (Comp& c : comps)
                    for (int v = 0; v < c.v; v++)
                        for (int h = 0; h < c.h; h++)
                            if (!block(c, mx * c.h + h, my * c.v + v)) return false;
                left--;
            }
        return true;
    }
    static const uint8_t* zigzag() {
        static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                       29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        return zz;
    }
    void layout() {      // plane / coefficient storage on the MCU-padded grid
        int hmax = 1, vmax = 1;
        for (Comp& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        coefs.resize(comps.size());
        for (size_t i = 0; i < comps.size(); i++) {
            Comp& c = comps[i];
            c.pw = mcux * c.h * 8; c.ph = mcuy * c.v * 8;
            c.plane.assign((size_t)c.pw * c.ph, 0);
            if (progressive) coefs[i].assign((size_t)c.pw * c.ph, 0);
        }
    }
    // one block of a progressive scan (ITU T.81 G.1.2; libjpeg jdphuff.c): spectral band [ss, se], successive approximation ah -> al
    bool prog_block(Comp& c, int16_t* b, int ss, int se, int ah, int al) {
        const uint8_t* zz = zigzag();
        if (ss == 0) {
            if (ah == 0) {
                int t = decode(dc[c.td]);
                if (t < 0 || t > 11) return fail("bad Huffman code (DC)");
                c.pred += extend(receive(t), t);
                b[0] = (int16_t)(c.pred * (1 << al));
            } else if (bit()) b[0] |= (int16_t)(1 << al);
            return true;
        }
        if (ah == 0) {
            if (eobrun > 0) { eobrun--; return true; }
            for (int k = ss; k <= se;) {
                int rs = decode(ac[c.ta]);
                if (rs < 0) return fail("bad Huffman code (AC)");
                int r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += receive(r); break; }
                    k += 16; continue;
                }
                k += r;
                if (k > se) return fail("AC run past the band");
                b[zz[k]] = (int16_t)(extend(receive(sz), sz) * (1 << al));
                k++;
            }
            return true;
        }
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        auto refine = [&](int16_t& cf) { if (bit() && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1)); };
        if (eobrun == 0) {
            for (; k <= se; k++) {
                int rs = decode(ac[c.ta]);
                if (rs < 0) return fail("bad Huffman code (AC)");
                int r = rs >> 4, sz = rs & 15, value = 0;
                if (sz) { if (sz != 1) return fail("bad refinement code"); value = bit() ? p1 : m1; }
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += receive(r); break; }
                for (; k <= se; k++) {
                    int16_t& cf = b[zz[k]];
                    if (cf != 0) refine(cf);
                    else if (--r < 0) break;
                }
                if (sz && k <= se) b[zz[k]] = (int16_t)value;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) { int16_t& cf = b[zz[k]]; if (cf != 0) refine(cf); }
            eobrun--;
        }
        return true;
    }
    bool prog_scan(const std::vector<int>& which, int ss, int se, int ah, int al) {
        if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && which.size() != 1) || al > 13 || ah > 13) return fail("bad progressive scan parameters");
        int hmax = 1, vmax = 1;
        for (Comp& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        for (int i : which) { Comp& c = comps[i]; c.pred = 0; if ((ss == 0 && ah == 0 && !dc[c.td].set) || (ss > 0 && !ac[c.ta].set)) return fail("missing Huffman table"); }
        eobrun = 0; bitcnt = 0; hit_marker = false;
        int left = restart;
        auto maybe_restart = [&]() {
            if (!restart || left > 0) return true;
            bitcnt = 0; hit_marker = false;
            if (p + 2 <= n && d[p] == 0xff && d[p + 1] >= 0xd0 && d[p + 1] <= 0xd7) p += 2; else return fail("missing restart marker");
            for (int i : which) comps[i].pred = 0;
            eobrun = 0; left = restart;
            return true;
        };
        if (which.size() == 1) {       // non-interleaved: the component's own block grid
            Comp& c = comps[which[0]];
            const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
            const int bw = (cw + 7) / 8, bh = (ch + 7) / 8, stride = c.pw / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    if (!maybe_restart()) return false;
                    if (!prog_block(c, &coefs[which[0]][((size_t)by * stride + bx) * 64], ss, se, ah, al)) return false;
                    left--;
                }
        } else {
            const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (!maybe_restart()) return false;
                    for (int i : which) {
                        Comp& c = comps[i];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++)
                                if (!prog_block(c, &coefs[i][((size_t)(my * c.v + v) * (c.pw / 8) + (mx * c.h + h)) * 64], ss, se, ah, al)) return false;
                    }
                    left--;
                }
        }
        return true;
    }
    bool prog_finish() {
        const uint8_t* zz = zigzag();
        for (size_t i = 0; i < comps.size(); i++) {
            Comp& c = comps[i];
            if (!qt_set[c.tq]) return fail("missing quantisation table");
            int natq[64];
            for (int k = 0; k < 64; k++) natq[zz[k]] = qt[c.tq][k];
            const int bw = c.pw / 8, bh = c.ph / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    const int16_t* b = &coefs[i][((size_t)by * bw + bx) * 64];
                    int coef[64];
                    for (int k = 0; k < 64; k++) coef[k] = b[k] * natq[k];
                    idct(coef, &c.plane[(size_t)by * 8 * c.pw + (size_t)bx * 8], c.pw);
                }
        }
        return true;
    }
    void emit(HostBitmap* out) {
        int hmax = 1, vmax = 1;
        for (Comp& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
        out->w = (uint32_t)width; out->h = (uint32_t)height;
        out->rgb.resize((size_t)3 * width * height);
        std::vector<uint8_t> Y = full(comps[0], hmax, vmax), Cb, Cr;
        if (comps.size() == 3) { Cb = full(comps[1], hmax, vmax); Cr = full(comps[2], hmax, vmax); }
        for (size_t i = 0; i < (size_t)width * height; i++) {
            int r, g, b;
            if (comps.size() == 1) r = g = b = Y[i];
            else {       // JFIF YCbCr -> RGB with libjpeg's 16-bit fixed-point tables
                const int y = Y[i], cb = Cb[i] - 128, cr = Cr[i] - 128;
                r = y + ((91881 * cr + 32768) >> 16);
                g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                b = y + ((116130 * cb + 32768) >> 16);
                r = r < 0 ? 0 : r > 255 ? 255 : r; g = g < 0 ? 0 : g > 255 ? 255 : g; b = b < 0 ? 0 : b > 255 ? 255 : b;
            }
            out->rgb[3 * i] = (float)r / 255.0f; out->rgb[3 * i + 1] = (float)g / 255.0f; out->rgb[3 * i + 2] = (float)b / 255.0f;   // read_ldr_image
        }
    }
    // upsample one component to width x height (libjpeg: triangle filter for 2:1, replication otherwise)
    std::vector<uint8_t> full(const Comp& c, int hmax, int vmax) const {
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;     // the component's own size
        std::vector<uint8_t> out((size_t)width * height);
        const int fx = hmax / c.h, fy = vmax / c.v;
        auto at = [&](int x, int y) { x = x < 0 ? 0 : (x >= cw ? cw - 1 : x); y = y < 0 ? 0 : (y >= ch ? ch - 1 : y); return (int)c.plane[(size_t)y * c.pw + x]; };
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                int v;
                if (fx == 1 && fy == 1) v = at(x, y);
                else if (fx == 2 && fy == 1 && hmax == 2 * c.h) {               // h2v1 fancy: (3 * near + far + 1|2) >> 2
                    const int i = x >> 1;
                    v = (x & 1) ? (3 * at(i, y) + at(i + 1, y) + 2) >> 2 : (3 * at(i, y) + at(i - 1, y) + 1) >> 2;
                    if ((x & 1) == 0 && i == 0) v = at(0, y);
                    if ((x & 1) && i == cw - 1) v = at(cw - 1, y);
                } else if (fx == 2 && fy == 2 && hmax == 2 * c.h && vmax == 2 * c.v) {   // h2v2 fancy: 9/3/3/1 triangle
                    const int i = x >> 1, j = y >> 1;
                    const int jn = (y & 1) ? j + 1 : j - 1;
                    auto col = [&](int xi) { return 3 * at(xi, j) + at(xi, jn); };
                    const int cur = col(i);
                    if ((x & 1) == 0) v = i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + col(i - 1) + 8) >> 4;
                    else v = i == cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + col(i + 1) + 7) >> 4;
                } else v = at(x * c.h / hmax, y * c.v / vmax);
                out[(size_t)y * width + x] = (uint8_t)v;
            }
        return out;
    }
    bool run(HostBitmap* out) {
        if (u16() != 0xffd8) return fail("not a JPEG");
        bool have_frame = false, have_scan = false;
        for (;;) {
            int m = u8();
            if (m < 0) { if (progressive && have_scan) { if (!prog_finish()) return false; emit(out); return true; } return fail("truncated file"); }
            if (m != 0xff) continue;
            do { m = u8(); } while (m == 0xff);
            if (m < 0) return fail("truncated file");
            if (m == 0xd9) { if (progressive && have_scan) { if (!prog_finish()) return false; emit(out); return true; } return fail("no image data"); }
            if (m == 0x00 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
            const int len = u16();
            if (len < 2 || p + (size_t)len - 2 > n) return fail("bad segment length");
            const size_t end = p + len - 2;
            if (m == 0xdb) {
                while (p < end) {
                    int pq = u8(); int t = pq & 15;
                    if (t > 3 || (pq >> 4) > 1) return fail("bad quantisation table");
                    for (int i = 0; i < 64; i++) { int v = (pq >> 4) ? u16() : u8(); if (v < 0) return fail("truncated table"); qt[t][i] = (uint16_t)v; }
                    qt_set[t] = true;
                }
            } else if (m == 0xc4) {
                while (p < end) {
                    int tc = u8(); int t = tc & 15;
                    if (t > 3 || (tc >> 4) > 1) return fail("bad Huffman table");
                    Huff& h = (tc >> 4) ? ac[t] : dc[t];
                    int total = 0;
                    for (int l = 1; l <= 16; l++) { int b = u8(); if (b < 0) return fail("truncated table"); h.bits[l] = (uint8_t)b; total += b; }
                    if (total > 256 || p + total > end) return fail("bad Huffman table");
                    for (int i = 0; i < total; i++) h.vals[i] = d[p++];
                    build(h);
                }
            } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
                progressive = m == 0xc2;
                if (u8() != 8) return fail("only 8-bit JPEGs are read");
                height = u16(); width = u16();
                int nc = u8();
                if (width <= 0 || height <= 0 || (int64_t)width * height > (1 << 28) || (nc != 1 && nc != 3)) return fail("unsupported frame (size / component count)");
                comps.resize(nc);
                for (Comp& c : comps) { c.id = u8(); int hv = u8(); c.h = hv >> 4; c.v = hv & 15; c.tq = u8(); if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq < 0 || c.tq > 3) return fail("bad component"); }
                if (nc == 1) { comps[0].h = comps[0].v = 1; }
                if (progressive) layout();
                have_frame = true;
            } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
                return fail("lossless / hierarchical / arithmetic-coded JPEGs are not read");
            } else if (m == 0xdd) restart = u16();
            else if (m == 0xda) {
                if (!have_frame) return fail("scan before frame");
                int ns = u8();
                if (ns < 1 || ns > (int)comps.size() || (!progressive && ns != (int)comps.size())) return fail("non-interleaved baseline scans are not read");
                std::vector<int> which;
                for (int i = 0; i < ns; i++) {
                    int id = u8(), t = u8(); bool ok = false;
                    for (size_t ci = 0; ci < comps.size(); ci++) if (comps[ci].id == id) { comps[ci].td = t >> 4; comps[ci].ta = t & 15; ok = comps[ci].td < 4 && comps[ci].ta < 4; which.push_back((int)ci); }
                    if (!ok) return fail("bad scan header");
                }
                if (progressive) {
                    const int ss = u8(), se = u8(), a = u8();
                    p = end;
                    if (!prog_scan(which, ss, se, a >> 4, a & 15)) return false;
                    have_scan = true;
                    continue;
                }
                p = end;
                if (!scan()) return false;
                emit(out);
                return true;
            }
            p = end;
        }
    }
};

int read_jpeg(const std::string& path, HostBitmap* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    JpegDecoder dec{reinterpret_cast<const unsigned char*>(src.data()), src.size()};
    if (!dec.run(out)) { *err = path + ": " + dec.err; return dec.err.find("not read") != std::string::npos ? RL_ERR_UNSUPPORTED : RL_ERR_PARSE; }
    return RL_OK;
}

// Truevision TGA (read_ldr_image): true-colour (24 / 32 bit) and grey (8 bit) images, raw or run-length encoded, either row order.
int read_tga(const std::string& path, HostBitmap* out, std::string* err) {
    std::string src;
    if (!slurp(path, &src)) { *err = "cannot open " + path; return RL_ERR_IO; }
    const unsigned char* d = reinterpret_cast<const unsigned char*>(src.data());
    const size_t n = src.size();
    if (n < 18) { *err = path + ": truncated TGA header"; return RL_ERR_PARSE; }
    const int id_len = d[0], cmap_type = d[1], type = d[2], cmap_len = d[5] | (d[6] << 8), cmap_bits = d[7];
    const int w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), bpp = d[16], desc = d[17];
    const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
    if (!(type == 2 || type == 3 || type == 10 || type == 11) || (grey ? bpp != 8 : (bpp != 24 && bpp != 32)) || w <= 0 || h <= 0) {
        *err = path + ": unsupported TGA variant (true-colour 24/32-bit and 8-bit grey only)"; return RL_ERR_UNSUPPORTED;
    }
    size_t p = 18 + (size_t)id_len + (cmap_type ? (size_t)cmap_len * ((cmap_bits + 7) / 8) : 0);
    const size_t px = (size_t)bpp / 8, total = (size_t)w * h;
    std::vector<unsigned char> pix(total * px);
    if (!rle) {
        if (p + pix.size() > n) { *err = path + ": truncated TGA"; return RL_ERR_PARSE; }
        std::memcpy(pix.data(), d + p, pix.size());
    } else {
        size_t o = 0;
        while (o < total) {
            if (p >= n) { *err = path + ": truncated TGA"; return RL_ERR_PARSE; }
            const int c = d[p++], cnt = (c & 0x7f) + 1;
            if (o + cnt > total) { *err = path + ": TGA run past the image"; return RL_ERR_PARSE; }
            if (c & 0x80) {
                if (p + px > n) { *err = path + ": truncated TGA"; return RL_ERR_PARSE; }
                for (int i = 0; i < cnt; i++) std::memcpy(&pix[(o + i) * px], d + p, px);
                p += px;
            } else {
                if (p + px * cnt > n) { *err = path + ": truncated TGA"; return RL_ERR_PARSE; }
                std::memcpy(&pix[o * px], d + p, px * cnt);
                p += px * cnt;
            }
            o += cnt;
        }
    }
    out->w = (uint32_t)w; out->h = (uint32_t)h;
    out->rgb.resize(total * 3);
    const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const unsigned char* q = &pix[((size_t)(top_down ? y : h - 1 - y) * w + (right_left ? w - 1 - x : x)) * px];
            const unsigned char r = grey ? q[0] : q[2], g = grey ? q[0] : q[1], b = q[0];       // stored B, G, R [, A]
            float* o3 = &out->rgb[3 * ((size_t)y * w + x)];
            o3[0] = (float)r / 255.0f; o3[1] = (float)g / 255.0f; o3[2] = (float)b / 255.0f;   // read_ldr_image
        }
    return RL_OK;
}
}  // namespace

// Bitmap::read (structure.rs:670-683): by extension
int read_image(const std::string& path, HostBitmap* out, std::string* err) {
    const size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? std::string() : path.substr(dot + 1);
    for (char& c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == "pfm") {
        int rc = read_pfm(path.c_str(), &out->w, &out->h, &out->rgb);
        if (rc != RL_OK) *err = "cannot read " + path;
        return rc;
    }
    if (ext == "png") return read_png(path, out, err);
    if (ext == "exr") return read_exr(path, out, err);
    if (ext == "jpg" || ext == "jpeg") return read_jpeg(path, out, err);
    if (ext == "tga") return read_tga(path, out, err);
    *err = path + ": only .pfm, .exr, .png, .jpg and .tga images are read";
    return RL_ERR_UNSUPPORTED;
}

}  // namespace rl
