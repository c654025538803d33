// Plain-old-data tables the device kernels read; built on the host by rt_flatten.cpp.
//
// HBM layout (all read-only during a render):
//   RtNode[n_nodes + 1]   one per object, depth-first pre-order (= the reference's visiting order,
//                         rendering/world.rs:29-32,66-68), plus one inert trailing record so that the traversal may
//                         always prefetch record i+1.  A group's subtree is nodes (i, skip); its children are at
//                         nesting depth + 1.  Every node carries rows 0..2 of the inverse of its LOCAL
//                         transformation: rays are taken down the hierarchy one local inverse at a time, as
//                         World::collect_intersections / Object::intersect do (world.rs:53-58, objects.rs:121-126).
//                         The node index is wave-uniform during traversal, so a record is fetched with two or three
//                         wide scalar loads (128 bytes for a leaf, all 160 when the scene has groups).
//   RtSurface[n_leaves]   one per leaf: rows 0..2 of the inverse of the cached WORLD transformation (normals and
//                         patterns use that one, objects.rs:87-105) plus material and pattern.  Read per lane, only
//                         for the object a ray actually hit.
#pragma once

#include <stdint.h>

// RtNode.xform: structure of `inv`, found on the host by exact comparison.  The traversal skips the products with
// exact zeros (and with exact ones for the identity); that changes no value, only possibly the sign of a zero.
enum { RT_XFORM_GENERAL = 0, RT_XFORM_DIAGONAL = 1 /* scale + translation */, RT_XFORM_IDENTITY = 2,
       RT_XFORM_DIAGONAL_BALL = 3 /* a sphere with a scale-and-translate transformation: see below */ };
// RT_XFORM_DIAGONAL_BALL: of a scale-and-translate inverse the device reads inv[0], [3], [5], [7], [10], [11] only (the
// other entries of the 3x3 part are exact zeros).  For a sphere, four of the unread entries hold a ball, in the space
// of the sibling list the sphere stands in, that contains every point at which the reference can report a hit:
//   inv[1], inv[2], inv[4]   its centre (the image of the sphere's origin)
//   inv[6]                   its radius squared: the largest scale factor squared, padded, over 1 - 1e-9 (rt_flatten.cpp)
// A wave none of whose rays comes near the ball skips the ray transform and the quadratic altogether.

struct RtNode {
    double inv[12];   // rows 0..2 of inverse(transformation), row-major 3x4
    int32_t tag;      // RT_SPHERE .. RT_GROUP or RT_SKIPBOX in bits 0..7; bit 8: cylinder / cone is closed;
                      // bits 9..15: nesting level of the sibling list the node belongs to (0 = the world's own list)
    int32_t xform;    // RT_XFORM_*
    int32_t leaf;     // index into RtSurface[], -1 for groups
    int32_t skip;     // index of the first node after this node's subtree
    double a[6];      // group: bounds min.xyz, max.xyz (in the group's own space); leaf: a[0]=minimum, a[1]=maximum;
                      // range-skip node: a[0..2] hold its box as SIX FLOATS, (min, max) per axis x, y, z, each rounded
                      // away from the box's inside (the device tests it in f32); a[3..5] unused
};
#define RT_NODE_CLOSED 0x100
#define RT_NODE_LEVEL_SHIFT 9
#define RT_NODE_LEVEL_MASK 0x7f
#define RT_NODE_LOOKAHEAD 0x10000 /* range-skip node: also holds the box of the range-skip node at `skip` (below) */
#define RT_NODE_LIST_Q 0x20000    /* RT_SKIP_CONES node whose q (inv[0..5]) is, bit for bit, that of every other such node of its
                                     sibling list: d^T q d of a ray need be formed once per visit of the list */
// A node kind of this table only (the reference has no such object): a conservative box, in the space of the
// sibling list it stands in, around the nodes (i, skip).  When no ray of a wave can touch the box the traversal
// continues at `skip`; otherwise at i + 1, with the same ray.  Its matrix is the identity and it has no leaf.
#define RT_SKIPBOX 6
#define RT_SKIP_PAD 0.01  /* padding of such a box, as a fraction of its half-extent per axis */
// RtNode.xform of a range-skip node: RT_SKIP_CONES when cones lie below it; then inv[0..5] is the symmetric matrix q
// (xx, xy, xz, yy, yz, zz) and inv[6] the bound: a ray with |d^T q d| < inv[6] must not be culled (see rt_flatten.cpp).
enum { RT_SKIP_PLAIN = 1, RT_SKIP_CONES = 3 };
// A range-skip node X whose `skip` record R is a range-skip node of the same sibling list may carry R's test as well
// (tag bit RT_NODE_LOOKAHEAD): when X rejects the wave, R is what the traversal would fetch and test next, so it is
// tested at once and the traversal continues at R + 1 or at R's own `skip` without ever fetching R.  Then
//   inv[7..9]  R's box as R itself holds it: six floats, (min, max) per axis
//   inv[10]    R's bound on |d^T q d| (0 when no cones lie below R); R shares X's q
//   leaf       R's `skip`

struct RtSurface {
    double inv_world[12];  // rows 0..2 of inverse(world_transformation)
    double color[3];
    double ambient, diffuse, specular, shininess, reflectivity, transparency, refractive_index;
    double minimum, maximum;
    double pat_inv[12];    // rows 0..2 of inverse(pattern.transform)
    double pat_a[3], pat_b[3];
    int32_t kind;
    int32_t pattern_kind;  // RT_PATTERN_*
    int32_t object;
    int32_t xform;         // RT_XFORM_* of inv_world (never RT_XFORM_DIAGONAL_BALL)
    int32_t flags;         // RT_SURFACE_*
    int32_t _pad;
};
// A top-level sphere that the light is well outside of: in the sphere's own space the light lies at least two radii from
// the centre.  A shadow ray that starts strictly inside such a sphere leaves it before it reaches the light, so the
// point it starts from is in the sphere's own shadow whatever else the world holds (rt::shade_hit uses this).
#define RT_SURFACE_LIGHT_OUTSIDE 1

struct RtLight {
    double position[3];
    double intensity[3];
    int32_t has_light;
    int32_t _pad;
};

struct RtCameraDev {
    double inv[12];     // rows 0..2 of inverse(camera.transform)
    double origin[3];   // inverse * Point(0,0,0), the same for every pixel (rendering/camera.rs:51)
    double half_width, half_height, pixel_size;
    uint32_t hsize, vsize;
};

// Computations (rendering/intersections.rs:12-24) as the device kernels read and write it; the same layout as the
// public rt_computations.  `object` is an index into RtSurface[] on the device side of the call.
struct RtComps {
    double time;
    int32_t object;
    int32_t inside;
    double point[3], eye_vector[3], normal_vector[3], reflect_vector[3], over_point[3], under_point[3];
    double n1, n2;
};
enum { RT_PART_SHADE_HIT = 0, RT_PART_REFLECTED = 1, RT_PART_REFRACTED = 2 };

// Per-call counters; field order matches rt_stats (minus kernel_ms).
struct RtCounters {
    unsigned long long rays, primary, shadow, reflect, refract;
    unsigned long long prim_tests[5], prim_hits[5];
    unsigned long long aabb_tests, ray_transforms, shades, lit, specular, patterns, refract_setups, schlick;
};
#define RT_NUM_COUNTERS 23
