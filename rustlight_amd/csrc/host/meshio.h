// meshio.h — mesh / image file readers shared by the PBRT and Mitsuba front ends (see meshio.cpp).
#pragma once
#include <string>
#include <tuple>
#include <vector>

#include "scene.h"

namespace rl {

struct LoadedMesh {
    std::string name;
    std::vector<float> pos, nrm, uv;      // 3 / 3 / 2 per vertex; nrm / uv empty when the file has none
    std::vector<uint32_t> idx;            // 3 per triangle
    // OBJ only: the bound MTL entry (geometry.rs:66-92)
    bool has_material = false, has_kd = false;
    float kd[3] = {0, 0, 0};
    std::string kd_map;                   // resolved path of map_Kd
};

int read_obj(const std::string& path, std::vector<LoadedMesh>* out, std::string* err);
int read_ply(const std::string& path, LoadedMesh* out, std::string* err);
int read_serialized(const std::string& path, int shape_index, LoadedMesh* out, std::string* err);
int read_image(const std::string& path, HostBitmap* out, std::string* err);   // Bitmap::read: .pfm, .png

}  // namespace rl
