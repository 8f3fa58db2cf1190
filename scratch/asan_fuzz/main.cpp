#include <cstdio>
#include <string>
#include "../../rustlight_amd/csrc/host/meshio.h"
int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        std::string p = argv[i], err;
        std::string ext = p.substr(p.find_last_of('.') + 1);
        int rc;
        if (ext == "obj") { std::vector<rl::LoadedMesh> m; rc = rl::read_obj(p, &m, &err); }
        else if (ext == "ply") { rl::LoadedMesh m; rc = rl::read_ply(p, &m, &err); }
        else if (ext == "serialized") { rl::LoadedMesh m; rc = rl::read_serialized(p, 0, &m, &err); }
        else { rl::HostBitmap b; rc = rl::read_image(p, &b, &err); }
        std::printf("%d\n", rc);
    }
    return 0;
}
