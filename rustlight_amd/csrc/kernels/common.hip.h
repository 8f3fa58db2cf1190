// common.hip.h — what every kernel translation unit includes: device types, the f32 contract, shading, traversal, path state, the four
// stage functions and the cross-unit launchers.
#pragma once
#include <hip/hip_runtime.h>

#include "../../../include/rustlight_amd.h"
#include "../device_types.h"
#include "devmath.hip.h"
#include "shading.hip.h"
#include "trace.hip.h"
#include "pathstate.hip.h"
#include "stages.hip.h"
#include "launch.h"
