// World description -> flat device tables (host side, once per scene).
#pragma once

#include <string>
#include <vector>

#include "../../include/raytracer_rs_amd.h"
#include "rt_device_types.h"

namespace rt {

struct FlatScene {
    // n_tables node tables of table_records records each, one after the other: the same nodes in the orders that suit
    // the eight octants of ray directions (oct_table[octant] = which table; bit a of the octant set = rays travelling
    // towards -a).  Every table ends with one inert record.  A world that keeps the reference's order has one table.
    std::vector<RtNode> nodes;
    uint32_t table_records = 0, n_tables = 0;
    uint8_t oct_table[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<RtSurface> surfaces;
    RtLight light{};
    uint32_t n_groups = 0;
    uint32_t n_leaves = 0;
    uint32_t n_skip_nodes = 0;  // range-skip nodes added to the table (not objects of the world)
    uint32_t max_nesting = 0;  // deepest nesting level of any node (top-level objects are level 0)
    bool reordered = false;    // some sibling list is emitted in spatial order instead of the reference's
};

// Returns RT_OK or an RT_ERR_* code with a message in err.
int flatten_world(const rt_world_desc &world, FlatScene &out, std::string &err);
int camera_to_device(const rt_camera_desc &camera, RtCameraDev &out, std::string &err);

}  // namespace rt
