// World description -> flat device tables (host side, once per scene).
#pragma once

#include <string>
#include <vector>

#include "../../include/raytracer_rs_amd.h"
#include "rt_device_types.h"

namespace rt {

struct FlatScene {
    std::vector<RtNode> nodes;
    std::vector<RtSurface> surfaces;
    RtLight light{};
    uint32_t n_groups = 0;
    uint32_t n_leaves = 0;
    uint32_t n_skip_nodes = 0;  // range-skip nodes added to the table (not objects of the world)
    uint32_t max_nesting = 0;  // deepest nesting level of any node (top-level objects are level 0)
    bool reordered = false;    // a sibling list was emitted in spatial order instead of the reference's
};

// Returns RT_OK or an RT_ERR_* code with a message in err.
int flatten_world(const rt_world_desc &world, FlatScene &out, std::string &err);
int camera_to_device(const rt_camera_desc &camera, RtCameraDev &out, std::string &err);

}  // namespace rt
