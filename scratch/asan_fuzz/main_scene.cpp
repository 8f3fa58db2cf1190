#include <cstdio>
#include <string>
#include "../../include/rustlight_amd.h"
int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) { rl_scene* s = nullptr; int rc = rl_scene_load(argv[i], 1, &s); if (rc == 0 && s) { rl_scene_build_emitters(s); } if (s) rl_scene_destroy(s); std::printf("%d\n", rc); }
    return 0;
}
