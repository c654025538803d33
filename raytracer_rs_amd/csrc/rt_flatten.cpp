// Flattens a World (objects in depth-first pre-order) into the tables of rt_device_types.h.
//
// What the reference recomputes for every ray is computed here once, with the same algorithms, because it is a
// pure function of the scene:
//   - inverse(transformation) per object          (rendering/objects.rs:122-126, rendering/world.rs:53-58)
//   - inverse(world_transformation) per leaf      (rendering/objects.rs:87-102)
//   - inverse(pattern.transform) per pattern      (scene/patterns.rs:140-144)
//   - Group::bounds() per group                   (geometry/groups.rs:130-149)
// A matrix the reference would fail to invert at render time (`expect("... should be invertible")`) is reported
// here as RT_ERR_NOT_INVERTIBLE instead of aborting.
#include "rt_flatten.hpp"

#include <cmath>
#include <cstdio>

#include "rt_math.hpp"

namespace rt {
namespace {

void rows3(const Matrix4 &m, double out[12]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = m.d[r][c];
}

// Structure of a 3x4 affine matrix by EXACT comparison (-0.0 == 0.0): which products the device may skip.
int32_t classify(const double m[12]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (r != c && m[r * 4 + c] != 0.0) return RT_XFORM_GENERAL;
    if (m[0] == 1.0 && m[5] == 1.0 && m[10] == 1.0 && m[3] == 0.0 && m[7] == 0.0 && m[11] == 0.0) return RT_XFORM_IDENTITY;
    return RT_XFORM_DIAGONAL;
}

// Shape::bounds of the leaf kinds
Bounds leaf_bounds(const rt_object_desc &o) {
    switch (o.kind) {
        case RT_SPHERE:  // geometry/sphere.rs:83-85
        case RT_CUBE:    // geometry/cubes.rs:101-103
            return {{-1.0, -1.0, -1.0}, {1.0, 1.0, 1.0}};
        case RT_PLANE:  // geometry/planes.rs:63-68
            return {{-INFINITY, 0.0, -INFINITY}, {INFINITY, 0.0, INFINITY}};
        case RT_CYLINDER:  // geometry/cylinders.rs:115-120
            return {{-1.0, o.minimum, -1.0}, {1.0, o.maximum, 1.0}};
        case RT_CONE: {  // geometry/cones.rs:130-136
            double limit = std::fmax(std::fabs(o.minimum), std::fabs(o.maximum));
            return {{-limit, o.minimum, -limit}, {limit, o.maximum, limit}};
        }
        default:
            return Bounds::empty();
    }
}

struct Builder {
    const rt_world_desc &world;
    FlatScene &out;
    std::string &err;
    int code = RT_OK;

    bool fail(int c, const char *fmt, unsigned idx) {
        char buf[160];
        std::snprintf(buf, sizeof buf, fmt, idx);
        err = buf;
        code = c;
        return false;
    }

    // Appends object `idx` and its subtree; returns its Shape::bounds() (in its own space) through `bounds` and the
    // index of the next sibling through `next`.
    bool build(uint32_t idx, uint32_t depth, Bounds &bounds, uint32_t &next) {
        if (idx >= world.n_objects) return fail(RT_ERR_INVALID_ARGUMENT, "object list ends inside the group before object #%u", idx);
        const rt_object_desc &o = world.objects[idx];
        if (o.kind < RT_SPHERE || o.kind > RT_GROUP) return fail(RT_ERR_INVALID_ARGUMENT, "object #%u has an unknown kind", idx);
        if (depth > out.max_nesting) out.max_nesting = depth;

        Matrix4 T = Matrix4::from_array(o.transformation);
        auto Tinv = T.inverse();
        if (!Tinv) return fail(RT_ERR_NOT_INVERTIBLE, o.kind == RT_GROUP ? "group transformation should be invertible (object #%u)"
                                                                         : "shape transformation should be invertible (object #%u)", idx);
        size_t ni = out.nodes.size();
        out.nodes.emplace_back();
        {
            RtNode &n = out.nodes[ni];
            std::memset(&n, 0, sizeof n);
            rows3(*Tinv, n.inv);
            n.xform = classify(n.inv);
            n.tag = o.kind | (o.closed ? RT_NODE_CLOSED : 0);
            n.leaf = -1;
        }

        if (o.kind == RT_GROUP) {
            if (o.n_children < 0) return fail(RT_ERR_INVALID_ARGUMENT, "group #%u has a negative child count", idx);
            out.n_groups++;
            Bounds gb = Bounds::empty();
            uint32_t child = idx + 1;
            for (int32_t k = 0; k < o.n_children; ++k) {
                if (child >= world.n_objects) return fail(RT_ERR_INVALID_ARGUMENT, "object list ends inside group #%u", idx);
                Matrix4 child_T = Matrix4::from_array(world.objects[child].transformation);
                Bounds cb;
                uint32_t after = 0;
                if (!build(child, depth + 1, cb, after)) return false;
                gb = gb.merge(cb.transform(child_T));  // groups.rs:143-145
                child = after;
            }
            RtNode &n = out.nodes[ni];
            n.a[0] = gb.min.x; n.a[1] = gb.min.y; n.a[2] = gb.min.z;
            n.a[3] = gb.max.x; n.a[4] = gb.max.y; n.a[5] = gb.max.z;
            n.skip = (int32_t)out.nodes.size();
            bounds = gb;
            next = child;
            return true;
        }

        Matrix4 W = Matrix4::from_array(o.world_transformation);
        auto Winv = W.inverse();
        if (!Winv) return fail(RT_ERR_NOT_INVERTIBLE, "world transformation should be invertible (object #%u)", idx);
        RtSurface s;
        std::memset(&s, 0, sizeof s);
        rows3(*Winv, s.inv_world);
        const rt_material_desc &m = o.material;
        for (int i = 0; i < 3; ++i) s.color[i] = m.color[i];
        s.ambient = m.ambient; s.diffuse = m.diffuse; s.specular = m.specular; s.shininess = m.shininess;
        s.reflectivity = m.reflectivity; s.transparency = m.transparency; s.refractive_index = m.refractive_index;
        s.minimum = o.minimum; s.maximum = o.maximum;
        s.kind = o.kind;
        s.object = (int32_t)idx;
        s.pattern_kind = RT_PATTERN_NONE;
        if (m.pattern.kind != RT_PATTERN_NONE) {
            if (m.pattern.kind < RT_PATTERN_STRIPE || m.pattern.kind > RT_PATTERN_CHECKERS)
                return fail(RT_ERR_INVALID_ARGUMENT, "object #%u has an unknown pattern kind", idx);
            auto Pinv = Matrix4::from_array(m.pattern.transform).inverse();
            if (!Pinv) return fail(RT_ERR_NOT_INVERTIBLE, "pattern transform should be invertible (object #%u)", idx);
            rows3(*Pinv, s.pat_inv);
            for (int i = 0; i < 3; ++i) { s.pat_a[i] = m.pattern.color_a[i]; s.pat_b[i] = m.pattern.color_b[i]; }
            s.pattern_kind = m.pattern.kind;
        }
        RtNode &n = out.nodes[ni];
        n.a[0] = o.minimum;
        n.a[1] = o.maximum;
        n.leaf = (int32_t)out.surfaces.size();
        n.skip = (int32_t)ni + 1;
        out.surfaces.push_back(s);
        out.n_leaves++;
        bounds = leaf_bounds(o);
        next = idx + 1;
        return true;
    }
};

}  // namespace

int flatten_world(const rt_world_desc &world, FlatScene &out, std::string &err) {
    out = FlatScene{};
    if (world.n_objects && !world.objects) {
        err = "rt_world_desc.objects is NULL";
        return RT_ERR_INVALID_ARGUMENT;
    }
    out.nodes.reserve(world.n_objects);
    Builder b{world, out, err};
    uint32_t idx = 0;
    while (idx < world.n_objects) {
        Bounds ignored;
        uint32_t next = 0;
        if (!b.build(idx, 0, ignored, next)) return b.code;
        idx = next;
    }
    if (out.max_nesting > RT_MAX_GROUP_NESTING) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "groups are nested %u levels deep; at most %d are supported", out.max_nesting,
                      RT_MAX_GROUP_NESTING);
        err = buf;
        return RT_ERR_UNSUPPORTED;
    }
    out.light.has_light = world.has_light ? 1 : 0;
    for (int i = 0; i < 3; ++i) {
        out.light.position[i] = world.light_position[i];
        out.light.intensity[i] = world.light_intensity[i];
    }
    return RT_OK;
}

int camera_to_device(const rt_camera_desc &camera, RtCameraDev &out, std::string &err) {
    auto inv = Matrix4::from_array(camera.transform).inverse();  // rendering/camera.rs:46-49
    if (!inv) {
        err = "camera transform should be invertible";
        return RT_ERR_NOT_INVERTIBLE;
    }
    rows3(*inv, out.inv);
    Vec3 origin = mul_point(*inv, Vec3{0.0, 0.0, 0.0});  // rendering/camera.rs:51
    out.origin[0] = origin.x; out.origin[1] = origin.y; out.origin[2] = origin.z;
    out.half_width = camera.half_width;
    out.half_height = camera.half_height;
    out.pixel_size = camera.pixel_size;
    out.hsize = camera.hsize;
    out.vsize = camera.vsize;
    return RT_OK;
}

}  // namespace rt
