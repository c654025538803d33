// gfx950 (MI355X, CDNA4) kernels of the per-pixel render path: camera ray generation, scene traversal with
// group-AABB culling, ray/shape intersection, Phong lighting with shadow rays, patterns, and the depth-bounded
// reflection/refraction tree with Schlick blending.  IEEE f64 throughout, compiled with -ffp-contract=off: the
// reference (rustc) never fuses a multiply-add, and a fused one flips decisions such as `discriminant < 0.0`.
//
// Execution model
//   * one lane per pixel (or per query ray); a wave of 64 lanes covers an 8x8 pixel tile so that neighbouring lanes
//     see the same objects; one wave per workgroup, tiles handed out in order of estimated cost.
//   * scene traversal is WAVE-UNIFORM: all lanes of a wave walk the node table with one scalar node index, so node
//     records (inverse matrices, bounds, kind) arrive by scalar loads into SGPRs and the shape switch is a scalar
//     branch.  A group's subtree is entered when ANY lane's ray passes the group's bounds (wave ballot); lanes whose
//     own ray missed are masked for the subtree.  Each lane therefore tests exactly the primitives the reference
//     tests for its ray, in the reference's order -- minus whole runs of nodes that a range-skip node (a padded box
//     the host adds in front of long sibling lists and expensive members) proves out of reach of every lane.
//   * the recursion of World::shade_hit / color_at (rendering/world.rs:84-126) is an iterative per-lane loop with ONE
//     traversal call site: a lane alternates between tracing a radiance ray and tracing the shadow ray of the hit it
//     found.  A hit is shaded for both outcomes of the shadow test before the shadow ray is traced; its reflection
//     and refraction rays carry a scalar path weight and wait on a per-lane stack: top entry in registers, two
//     levels in LDS, the rest in scratch.
//   * no MFMA: nothing on this path is a dense contraction.  The node table is read through the scalar cache, the
//     per-hit surface records through the vector L1; LDS holds only the ray stacks.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>  // launch_render_levels scans its per-wave ray counts

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/raytracer_rs_amd.h"
#include "rt_device.hpp"

#ifdef RT_TIMING
// Developer build (-DRT_TIMING): per-wave shader-clock totals of the phases of radiance() (tools/timing.py reads them).
#define RT_TIMING_WAVES (1 << 20)
__device__ unsigned long long g_rt_timing[RT_TIMING_WAVES][8];  // per wave: trace | shade<<32, real-time begin, iterations | lanes<<32, lifetime | real-time lifetime<<40, skip | group<<32 node visits, leaf visits | records fetched twice<<32, the tile rendered, unused
extern "C" int rt_debug_timing(unsigned long long *out, unsigned n_waves) {
    if (n_waves > RT_TIMING_WAVES) n_waves = RT_TIMING_WAVES;
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rt_timing), (size_t)n_waves * 8 * sizeof(unsigned long long));
    return e == hipSuccess ? 0 : -4;
}
#define RT_TIC(var) const long long var = clock64()
__shared__ unsigned rt_wave_nodes[4];  // node records this wave visited: skip, group, leaf; wrong guesses of the next record (one wave per workgroup)
#define RT_NODE_VISIT(k) if ((threadIdx.x & 63) == 0) atomicAdd(&rt_wave_nodes[k], 1u)
#else
#define RT_TIC(var)
#define RT_NODE_VISIT(k)
#endif

namespace rt {
namespace {

#define RT_CONSTANT_AS __attribute__((address_space(4)))
typedef const double RT_CONSTANT_AS *cdp;
typedef const RtNode RT_CONSTANT_AS *NodeP;

constexpr double EPS = 1e-5;          // core/floats.rs:2
constexpr double AABB_EPS = 1e-10;    // geometry/bounds.rs:114
constexpr double CHECKER_EPS = 1e-6;  // scene/patterns.rs:106
constexpr int STACK_MAX = RT_MAX_DEPTH + 1;  // in memory: at most one parked ray per level, plus one while a hit pushes two
// ... of which the first ones are kept in LDS, 64 B per lane and level = 4 KB per wave and level: two in the kernels
// for flat scenes, one in those for scenes with groups, which also park a traversal ray in LDS (3 KB) and would
// otherwise be limited to 14 waves per CU by their 11 KB (one level: chapter_14 -5 %, generated scene at 8K -6 %)
#ifndef RT_NEST_LDS_LEVELS
#define RT_NEST_LDS_LEVELS 1
#endif
#ifndef RT_SHARE_FLAT
#define RT_SHARE_FLAT 0
#endif
template <int NEST>
constexpr int LDS_LEVELS = NEST > 0 ? RT_NEST_LDS_LEVELS : 2;
// Dynamic LDS of every kernel here, per wave of the workgroup: the LDS levels of the pending-ray stack (kernels that
// run radiance()), then the slot in which a traversal parks the level-1 ray while it walks a nested group (kernels
// built for scenes with groups: 6 x 64 doubles = 3 KB).  11 KB per wave for the render kernel of a scene with groups.
template <int NEST, bool RADIANCE>
constexpr unsigned LDS_WAVE_DOUBLES = ((RADIANCE ? LDS_LEVELS<NEST> * 8 : 0) + (NEST > 0 ? 6 : (RADIANCE && RT_SHARE_FLAT) ? 2 : 0)) * 64;
template <int NEST, bool RADIANCE>
constexpr size_t lds_bytes(unsigned block) { return (size_t)(block / 64) * LDS_WAVE_DOUBLES<NEST, RADIANCE> * sizeof(double); }
extern __shared__ double ray_stack_lds[];

enum {
    C_RAYS = 0, C_PRIMARY, C_SHADOW, C_REFLECT, C_REFRACT,
    C_PRIM_TESTS = 5, C_PRIM_HITS = 10,
    C_AABB = 15, C_XFORM, C_SHADES, C_LIT, C_SPECULAR, C_PATTERNS, C_REFRACT_SETUPS, C_SCHLICK
};
static_assert(C_SCHLICK + 1 == RT_NUM_COUNTERS, "counter table out of step with RtCounters");

template <bool STATS>
struct Ctr {
    unsigned c[RT_NUM_COUNTERS];
    __device__ __forceinline__ void clear() {
        for (int i = 0; i < RT_NUM_COUNTERS; ++i) c[i] = 0;
    }
    __device__ __forceinline__ void add(int i) { c[i]++; }
};
template <>
struct Ctr<false> {
    // How many range-skip boxes and group bounds this lane's ray passed: a cost estimate that only the tile
    // classification reads (everywhere else the increments are dead code).
    unsigned visits = 0;
    __device__ __forceinline__ void clear() {}
    __device__ __forceinline__ void add(int) {}
};

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
// core/tuples.rs:59-61
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// core/tuples.rs:35-37
__device__ __forceinline__ double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// core/tuples.rs:54-56: self - (normal * 2.0) * self.dot(normal)
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (n * 2.0) * dot(v, n); }

// v / m for a vector and its own magnitude (Tuple::normalize, core/tuples.rs:39-47; also the three divisions by
// `distance` in World::is_shadowed).  The compiler expands every f64 division into the same 11-instruction sequence:
// scale, reciprocal, two Newton steps, quotient, remainder, correction, fix-up.  The reciprocal and its refinement
// depend on the denominator only, so the three divisions share them here; quotient, remainder and correction are the
// hardware sequence's own steps and give its (correctly rounded) result bit for bit whenever that sequence would not
// rescale its operands -- which the exponent test on m guarantees for any component that is zero or within 2^-450 of
// m.  A -0.0 component comes out as +0.0.  Anything else takes the plain divisions.
__device__ __forceinline__ V3 div_by_magnitude(V3 v, double m) {
    const unsigned biased = ((unsigned)__double2hiint(m) >> 20) - 523u;  // sign bit set (never for a magnitude) fails too
    if (biased < 1000u) {  // 2^-500 <= m < 2^500
        double r = __builtin_amdgcn_rcp(m);
        double e = __builtin_fma(-m, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-m, r, 1.0);
        r = __builtin_fma(r, e, r);
        double qx = v.x * r, qy = v.y * r, qz = v.z * r;
        double rx = __builtin_fma(-m, qx, v.x), ry = __builtin_fma(-m, qy, v.y), rz = __builtin_fma(-m, qz, v.z);
        return mk(__builtin_fma(rx, r, qx), __builtin_fma(ry, r, qy), __builtin_fma(rz, r, qz));
    }
    return mk(v.x / m, v.y / m, v.z / m);
}

// Matrix4 * Point / Vector (core/matrices.rs:195-221) on rows 0..2 of an affine matrix.  The point form adds
// row[3]*1.0 == row[3]; the vector form would add row[3]*0.0, which can only change the sign of an exact zero and
// is left out.
template <typename P>
__device__ __forceinline__ V3 mul_point(P m, V3 p) {
    return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
template <typename P>
__device__ __forceinline__ V3 mul_vector(P m, V3 v) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// transpose(M) * Vector (rendering/objects.rs:102)
template <typename P>
__device__ __forceinline__ V3 mul_vector_transposed(P m, V3 v) {
    return mk(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
              m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

struct SceneK {
    NodeP nodes;
    const RtSurface *surfaces;
    int n_nodes;
    int max_depth;
    unsigned stack_base, park_base;  // this wave's parts of the dynamic LDS, in doubles from ray_stack_lds
    unsigned table_bytes;            // size of one node table
    unsigned long long oct_tables;   // byte k: the table for rays of direction octant k
    RtLight light;
};

// Closest hit so far: the smallest t >= 0, the earliest emitted winning ties -- the element
// Intersection::hit (rendering/intersections.rs:37-39) finds first in the stably sorted list (world.rs:34).
// `t` starts as NaN: `!(x >= NaN)` holds for every x, so the first non-negative candidate (even +infinity) is taken,
// and from then on t is an ordinary number and `!(x >= t)` is `x < t`.  Candidates arrive in the reference's emission
// order, so the strict comparison keeps the earliest of equal t's.  Written with selects: no branch per candidate.
//
// TIES: the node table of a scene with range-skip nodes lists long sibling lists in spatial order, not in the
// reference's (rt_flatten.cpp), so "earliest emitted" is no longer "first met".  Leaves are numbered in the
// reference's order whatever the table's, so among equal t's the smaller leaf number is the earlier emission.
template <bool TIES>
struct Closest {
    static constexpr bool every = false;  // the accumulator wants the closest hit, not every intersection
    double t;
    int leaf;
};
template <bool TIES>
__device__ __forceinline__ void cand(bool ok, double t, int leaf, Closest<TIES> &b) {
    bool nearer = !(t >= b.t);
    if constexpr (TIES) nearer |= (t == b.t) & (leaf < b.leaf);
    const bool better = ok & (t >= 0.0) & nearer;
    b.t = better ? t : b.t;
    b.leaf = better ? leaf : b.leaf;
}
// Two candidates of ONE object with ta <= tb (or a NaN among them), emitted ta first: the later one can only win
// when the earlier one is not eligible, so one update does for both.
template <bool TIES>
__device__ __forceinline__ void cand_ordered(bool oka, double ta, bool okb, double tb, int leaf, Closest<TIES> &b) {
    const bool first = oka & (ta >= 0.0);
    cand(first | okb, first ? ta : tb, leaf, b);
}
template <bool TIES>
__device__ __forceinline__ double hit_limit(const Closest<TIES> &b, double far) { return fmin(b.t, far); }
template <bool TIES>
__device__ __forceinline__ bool settled(const Closest<TIES> &) { return false; }
template <bool TIES>
__device__ __forceinline__ void pin(Closest<TIES> &b) { asm volatile("" : "+v"(b.t), "+v"(b.leaf)); }
// The kernels built for flat scenes without range-skip nodes (NEST == 0) always meet the objects in the reference's
// order; of the others, the render kernel exists in both forms (the host knows whether it reordered anything) and
// the rest assume it did.
template <int NEST>
constexpr bool ANY_ORDER = NEST > 0;

// What World::is_shadowed (rendering/world.rs:128-149) asks of the same traversal: is the closest hit with t >= 0
// nearer than `far`?  Which object it belongs to does not matter, and a lane that has its answer (any candidate
// with 0 <= t < far settles it) needs no further range of nodes opened on its behalf.
struct Occluded {
    static constexpr bool every = false;
    double t;  // closest so far, as in Best
    double far;
};
__device__ __forceinline__ void cand(bool ok, double t, int, Occluded &b) {
    const bool better = ok & (t >= 0.0) & !(t >= b.t);
    b.t = better ? t : b.t;
}
__device__ __forceinline__ void cand_ordered(bool oka, double ta, bool okb, double tb, int leaf, Occluded &b) {
    const bool first = oka & (ta >= 0.0);
    cand(first | okb, first ? ta : tb, leaf, b);
}
__device__ __forceinline__ double hit_limit(const Occluded &b, double) { return b.far; }
__device__ __forceinline__ bool settled(const Occluded &b) { return b.t < b.far; }
__device__ __forceinline__ void pin(Occluded &b) { asm volatile("" : "+v"(b.t)); }

// World::intersect itself (rendering/world.rs:27-36): every intersection the reference pushes, in the order it pushes
// them (negative and NaN t's included), into the lane's own slice of the output; `count` keeps counting past `cap` so
// that the caller learns the size it needs.
struct AllHits {
    static constexpr bool every = true;
    double *t;
    int *leaf;
    unsigned cap, count;
};
__device__ __forceinline__ void cand(bool ok, double t, int leaf, AllHits &b) {
    if (ok) {
        if (b.count < b.cap) {
            b.t[b.count] = t;
            b.leaf[b.count] = leaf;
        }
        b.count++;
    }
}
__device__ __forceinline__ void cand_ordered(bool oka, double ta, bool okb, double tb, int leaf, AllHits &b) {
    cand(oka, ta, leaf, b);
    cand(okb, tb, leaf, b);
}
__device__ __forceinline__ void pin(AllHits &) {}

// geometry/cubes.rs:41-59 (threshold EPSILON, extents -1..1) and geometry/bounds.rs:110-128 (threshold 1e-10)
__device__ __forceinline__ void check_axis(double origin, double direction, double lo, double hi, double threshold,
                                           double &tmin, double &tmax) {
    double n0 = lo - origin;
    double n1 = hi - origin;
    double a, b;
    if (fabs(direction) >= threshold) {
        a = n0 / direction;
        b = n1 / direction;
    } else {
        a = n0 * INFINITY;
        b = n1 * INFINITY;
    }
    const bool swap = a > b;
    tmin = swap ? b : a;
    tmax = swap ? a : b;
}

// geometry/bounds.rs:94-106; v_max_f64 / v_min_f64 return the non-NaN operand like f64::max / f64::min
__device__ __forceinline__ bool aabb_hit(const double (&bb)[6], V3 o, V3 d) {
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(o.x, d.x, bb[0], bb[3], AABB_EPS, xtmin, xtmax);
    check_axis(o.y, d.y, bb[1], bb[4], AABB_EPS, ytmin, ytmax);
    check_axis(o.z, d.z, bb[2], bb[5], AABB_EPS, ztmin, ztmax);
    double tmin = fmax(fmax(xtmin, ytmin), ztmin);
    double tmax = fmin(fmin(xtmax, ytmax), ztmax);
    return tmin <= tmax;
}

// geometry/sphere.rs:51-74.  a = d.d >= 0, so (-b - sq)/(2a) <= (-b + sq)/(2a) unless one is NaN.
template <class B>
__device__ __forceinline__ bool hit_sphere(V3 o, V3 d, bool on, int leaf, B &best) {
    double a = dot(d, d);
    double b = 2.0 * dot(d, o);
    double c = dot(o, o) - 1.0;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double den = 2.0 * a;
    // The far root can only be the closest hit when the near one is negative (or NaN): most rays never need it.
    double t = (-b - sq) / den;
    if constexpr (B::every) {
        cand_ordered(on, t, on, (-b + sq) / den, leaf, best);
    } else {
        if (on && !(t >= 0.0)) t = (-b + sq) / den;
        cand(on, t, leaf, best);
    }
    return true;
}

// geometry/planes.rs:46-56
template <class B>
__device__ __forceinline__ bool hit_plane(V3 o, V3 d, bool on, int leaf, B &best) {
    if (fabs(d.y) < EPS) return false;
    cand(on, -o.y / d.y, leaf, best);
    return true;
}

// Is Plane::local_intersect's t = -o.y / d.y certainly not a hit (negative and not zero)?  The sign of a correctly
// rounded quotient is the sign of the exact one unless it underflows to zero (-0.0 >= 0.0 holds, so a zero counts as a
// hit): o.y and d.y of the same sign, and |o.y / d.y| >= 2^-500.  (A ray with |d.y| < EPSILON is no hit either, but
// that the plane's own test finds out quickly enough.)
__device__ __forceinline__ bool plane_behind(double oy, double dy) {
    return (oy * dy > 0.0) & (fabs(oy) >= fabs(dy) * 0x1p-500);
}

// geometry/cubes.rs:63-86.  Both roots are emitted only when !(tmin > tmax).
template <class B>
__device__ __forceinline__ bool hit_cube(V3 o, V3 d, bool on, int leaf, B &best) {
    double tmin, tmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(o.x, d.x, -1.0, 1.0, EPS, tmin, tmax);
    check_axis(o.y, d.y, -1.0, 1.0, EPS, ytmin, ytmax);
    tmin = fmax(tmin, ytmin);
    tmax = fmin(tmax, ytmax);
    if (tmin > tmax) return false;
    check_axis(o.z, d.z, -1.0, 1.0, EPS, ztmin, ztmax);
    tmin = fmax(tmin, ztmin);
    tmax = fmin(tmax, ztmax);
    if (tmin > tmax) return false;
    cand_ordered(on, tmin, on, tmax, leaf, best);
    return true;
}

// geometry/cylinders.rs:124-144 (radius 1) and geometry/cones.rs:140-161 (radius |y|)
template <bool CONE, class B>
__device__ __forceinline__ bool hit_caps(V3 o, V3 d, double minimum, double maximum, bool closed, bool on, int leaf,
                                         B &best) {
    if (!closed || fabs(d.y - 0.0) < EPS) return false;
    double t0 = (minimum - o.y) / d.y;
    double x0 = o.x + t0 * d.x;
    double z0 = o.z + t0 * d.z;
    double r0 = CONE ? fabs(minimum) : 1.0;
    const bool in0 = (x0 * x0 + z0 * z0) <= (CONE ? r0 * r0 : 1.0);
    cand(in0 & on, t0, leaf, best);
    double t1 = (maximum - o.y) / d.y;
    double x1 = o.x + t1 * d.x;
    double z1 = o.z + t1 * d.z;
    double r1 = CONE ? fabs(maximum) : 1.0;
    const bool in1 = (x1 * x1 + z1 * z1) <= (CONE ? r1 * r1 : 1.0);
    cand(in1 & on, t1, leaf, best);
    return in0 | in1;
}

// The two wall roots of a cylinder or cone, each kept when its y lies strictly between the limits
// (cylinders.rs:78-96, cones.rs:85-103)
template <bool COUNT, class B>  // COUNT: the caller wants to know whether the reference's hit list is empty, not only the closest hit
__device__ __forceinline__ bool wall_roots(V3 o, V3 d, double b, double sq, double den, double minimum, double maximum,
                                           bool on, int leaf, B &best) {
    // The reference computes both roots and swaps them when t0 > t1.  sq >= 0, so -b - sq <= -b + sq, and a correctly
    // rounded quotient is monotonic in its numerator: t0 <= t1 when den > 0 and t0 >= t1 when den < 0 (a cone's `a` can
    // be negative).  The smaller root is therefore known before either division; with a NaN anywhere both roots are
    // NaN and neither is accepted, whichever way round they are named.
    const bool swap = !(den > 0.0);
    double ta = (swap ? (-b + sq) : (-b - sq)) / den;
    double y0 = o.y + ta * d.y;
    const bool in0 = minimum < y0 && y0 < maximum;
    bool in1 = false;
    if (COUNT || B::every || (on && !(in0 && ta >= 0.0))) {  // the larger root matters only when the smaller is no hit
        double tb = (swap ? (-b - sq) : (-b + sq)) / den;
        double y1 = o.y + tb * d.y;
        in1 = minimum < y1 && y1 < maximum;
        cand_ordered(in0 & on, ta, in1 & on, tb, leaf, best);
    } else {
        cand(in0 & on, ta, leaf, best);
    }
    return in0 | in1;
}

// geometry/cylinders.rs:57-99
template <bool COUNT, class B>
__device__ __forceinline__ bool hit_cylinder(V3 o, V3 d, double minimum, double maximum, bool closed, bool on,
                                             int leaf, B &best) {
    double a = d.x * d.x + d.z * d.z;
    bool any = false;
    if (!(fabs(a - 0.0) < EPS)) {
        double b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
        double c = o.x * o.x + o.z * o.z - 1.0;
        double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return false;  // no cap test on this branch (cylinders.rs:73-75)
        any = wall_roots<COUNT, B>(o, d, b, sqrt(disc), 2.0 * a, minimum, maximum, on, leaf, best);
    }
    const bool caps = hit_caps<false, B>(o, d, minimum, maximum, closed, on, leaf, best);
    return any | caps;
}

// geometry/cones.rs:57-107
template <bool COUNT, class B>
__device__ __forceinline__ bool hit_cone(V3 o, V3 d, double minimum, double maximum, bool closed, bool on, int leaf,
                                         B &best) {
    double a = d.x * d.x - d.y * d.y + d.z * d.z;
    double b = 2.0 * o.x * d.x - 2.0 * o.y * d.y + 2.0 * o.z * d.z;
    double c = o.x * o.x - o.y * o.y + o.z * o.z;
    bool any = false;
    if (fabs(a - 0.0) < EPS) {
        if (!(fabs(b - 0.0) < EPS)) {
            double t = -c / (2.0 * b);
            double y = o.y + t * d.y;
            any = minimum < y && y < maximum;
            cand(any & on, t, leaf, best);
        }
    } else {
        double disc = b * b - 4.0 * a * c;
        // a negative discriminant still falls through to the caps (cones.rs:80-83)
        if (!(disc < 0.0)) any = wall_roots<COUNT, B>(o, d, b, sqrt(disc), 2.0 * a, minimum, maximum, on, leaf, best);
    }
    const bool caps = hit_caps<true, B>(o, d, minimum, maximum, closed, on, leaf, best);
    return any | caps;
}

// A node record as it arrives in SGPRs (the node index is wave-uniform): two or three wide scalar loads.
//   a[0..15]  inv[0..7]
//   b[0..7]   inv[8..11];  b[8..11] tag, xform, leaf, skip;  b[12..15] a[0], a[1] (cylinder / cone limits)
//   c[0..7]   a[2..5] -- with b[12..15] a group's bounds; fetched only by kernels built for scenes with groups
//
// The loads are `volatile`: that pins them where they are written.  The traversal asks for the following record as
// soon as the matrix of the current one has been used, and without the qualifier the optimiser sinks that request
// down to its first use (the next iteration), which exposes the full scalar-cache latency once per node.  The
// compiler still tracks the outstanding loads itself and places the s_waitcnt before the first use.
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
struct NodeRaw {
    i32x16 a;
    i32x16 b;
    i32x8 c;
};
static_assert(sizeof(RtNode) == 160 && offsetof(RtNode, tag) == 96 && offsetof(RtNode, a) == 112, "RtNode layout moved");

template <bool WIDE>
__device__ __forceinline__ NodeRaw node_fetch(NodeP table, int index) {
    // a 32-bit byte offset (tables are far below 4 GB): one scalar multiply instead of a 64-bit one
    const char RT_CONSTANT_AS *p = (const char RT_CONSTANT_AS *)table + (unsigned)index * (unsigned)sizeof(RtNode);
    NodeRaw r;
    r.a = *(const volatile i32x16 RT_CONSTANT_AS *)(p);
    r.b = *(const volatile i32x16 RT_CONSTANT_AS *)(p + 64);
    if (WIDE) r.c = *(const volatile i32x8 RT_CONSTANT_AS *)(p + 128);
    return r;
}

__device__ __forceinline__ double dwords_to_double(int lo, int hi) { return __hiloint2double(hi, lo); }
// The loads are scalar (uniform address), but the optimiser treats the result of a volatile load as possibly
// divergent; the words that steer the traversal are marked wave-uniform explicitly.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Ray::transform (rendering/rays.rs:22-24) by a node's inverse.  xform is wave-uniform, so this is a scalar branch:
// a scale-and-translate inverse (every sphere of a typical scene) needs 9 operations instead of 33.
__device__ __forceinline__ void to_local(const NodeRaw &r, int xform, V3 o, V3 d, V3 &lo, V3 &ld) {
    double m[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = dwords_to_double(r.a[2 * k], r.a[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) m[8 + k] = dwords_to_double(r.b[2 * k], r.b[2 * k + 1]);
    if (xform != RT_XFORM_GENERAL) {
        // scale and translation only (the identity included: multiplying by its exact ones and adding its exact
        // zeros costs the same as copying the ray and needs no third code path)
        lo = mk(m[0] * o.x + m[3], m[5] * o.y + m[7], m[10] * o.z + m[11]);
        ld = mk(m[0] * d.x, m[5] * d.y, m[10] * d.z);
    } else {
        lo = mul_point(m, o);
        ld = mul_vector(m, d);
    }
}

// A sphere whose record carries a ball (RT_XFORM_DIAGONAL_BALL, rt_device_types.h): can the reference not report a hit
// with t >= 0 for the ray (o, d) of the record's sibling list?  The ball contains every point at which it can (the host
// pads it for the rounding of the reference's own arithmetic, rt_flatten.cpp; the part of that padding that grows
// with the distance of the ray's origin is taken off here: 1e-9 of |o - centre|^2, ten times what is needed, which
// also covers this function's own rounding).  With oc = centre - o, q = oc.d and m = |oc|^2 (1 - 1e-9) - r^2:
//   m > 0                    the origin is outside the ball, by more than the padding
//   q^2 < |d|^2 m            the line passes the ball: squared distance |oc|^2 - q^2 / |d|^2 > r^2 (+ padding)
//   q < 0                    the ball lies behind the origin: both roots of the reference's quadratic are negative,
//                            and with the origin outside by a relative 5e-10 its computed larger root -b + sqrt(b^2 -
//                            4ac) is too, because sqrt(b^2 - 4ac) <= b sqrt(1 - c / |lo|^2) is below b by far more
//                            than a rounding error
// This is the library's own test, not the reference's arithmetic: fused multiply-adds are fine here.  When no ray of a
// wave can hit, the wave skips the record: no ray transform, no quadratic.  On the 4K chapter_11 frame that is 85 % of
// all sphere records met (measured: neighbouring pixels' rays mostly pass the same few of its six spheres).
// (The record holds r^2 / (1 - 1e-9) and the caller passes dd = |d|^2 (1 - 1e-9): the same conditions with m scaled by a
// positive constant, and one operation fewer per record.  No short-circuit evaluation: one straight run of code.)
constexpr double BALL_SHRINK = 1.0 - 1e-9;
__device__ __forceinline__ bool misses_ball(const NodeRaw &r, V3 o, V3 d, double dd) {
    const double cx = dwords_to_double(r.a[2], r.a[3]), cy = dwords_to_double(r.a[4], r.a[5]), cz = dwords_to_double(r.a[8], r.a[9]);
    const double r2 = dwords_to_double(r.a[12], r.a[13]);
    const double x = cx - o.x, y = cy - o.y, z = cz - o.z;
    const double q = __builtin_fma(z, d.z, __builtin_fma(y, d.y, x * d.x));
    const double oc2 = __builtin_fma(z, z, __builtin_fma(y, y, x * x));
    const double m = oc2 - r2;
    const double apart = __builtin_fma(-dd, m, q * q);  // q^2 - |d|^2 (1 - 1e-9) m
    return (m > 0.0) & (fmin(q, apart) < 0.0);
}

// Range-skip test: may the reference find a hit with 0 <= t <= limit among objects inside the box?
//
// The box is padded on the host by RT_SKIP_PAD of its half-extent, which covers all rounding and the one place where
// the reference accepts a hit slightly outside a bounded shape: a cube axis with |d| < EPSILON is treated as parallel
// (cubes.rs:47-53), so the accepted point may sit up to EPSILON*t outside the slab.  That grows with t, hence a guard
// for boxes with a cube below them: `reach` bounds the distance from the ray origin to any point of the box,
// dmax = max|d_i|, so every hit in it has t <= reach / (0.99 dmax); culling is allowed only while EPSILON*t stays under
// the padding (reach <= 495 dmax: EPSILON * 500 / 0.99 < RT_SKIP_PAD / 1.9).  (Leaving the guard out for boxes with no
// cube below them, whose hits all lie ON their shapes, was measured: chapter_14 -0.7 %, the generated scene +2 %: not
// kept.)  Cones that take the reference's linear shortcut are dealt with by the caller: no box bounds those hits.  A
// NaN limit (no hit yet) rejects nothing.
//
// The test runs in f32: it is the library's own conservative test, not the reference's arithmetic, and in f32 it is
// two thirds of the instructions (packed subtractions and products, three-operand minima and maxima; chapter_14 at
// 4K 4.95 -> 4.75 ms, the generated scene at 8K 81 -> 76.5 ms, frames bit-identical).  What makes it safe:
//   * the record holds the padded box as floats rounded AWAY from its inside (rt_flatten.cpp), so the box only grows;
//   * the ray's origin is rounded to f32 (an error of 2^-24 |o| per coordinate, at most 2^-24 omax imax in t) and its
//     reciprocal direction is good to 2^-22 relative; every comparison below leaves room for sixteen times that:
//     e2 = 2^-20 omax imax absolutely (omax the largest |origin coordinate|, imax the largest finite |1/d|) plus 4e-6 of
//     the t's compared;
//   * on an axis with d = 0 the reciprocal is infinite and t is +-inf or NaN exactly as in the f64 form: NaN drops out
//     of v_min / v_max, equal infinities give inf - inf = NaN and no cull;
//   * anything not finite anywhere makes a comparison false, and a false comparison never culls.
// `limit` is the closest hit so far (or the light's distance), NaN for none.
struct SkipRay {
    float ox, oy, oz;  // origin
    float ix, iy, iz;  // 1 / direction
    float reach_max;   // 495 max|d|: boxes further away than this are not culled (see box_may_hit)
    float e2;          // absolute room in t
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ SkipRay skip_ray(V3 o, V3 d) {
    SkipRay s;
    s.ox = (float)o.x, s.oy = (float)o.y, s.oz = (float)o.z;
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    s.ix = __builtin_amdgcn_rcpf(dx), s.iy = __builtin_amdgcn_rcpf(dy), s.iz = __builtin_amdgcn_rcpf(dz);
    s.reach_max = 495.0f * fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
    const float inf = __builtin_inff();
    const float ax = fabsf(s.ix), ay = fabsf(s.iy), az = fabsf(s.iz);
    const float imax = fmaxf(fmaxf(ax < inf ? ax : 0.0f, ay < inf ? ay : 0.0f), az < inf ? az : 0.0f);
    const float omax = fmaxf(fmaxf(fabsf(s.ox), fabsf(s.oy)), fabsf(s.oz));
    s.e2 = 0x1p-20f * omax * imax;
    return s;
}
__device__ __forceinline__ bool box_may_hit32(f32x2 bx, f32x2 by, f32x2 bz, const SkipRay &r, double limit) {
    const f32x2 x = bx - r.ox, y = by - r.oy, z = bz - r.oz;  // (low face, high face) - origin, per axis
    const f32x2 tx = x * r.ix, ty = y * r.iy, tz = z * r.iz;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fminf(tz.x, tz.y));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fmaxf(tz.x, tz.y));
    const float reach = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(x.x), fabsf(x.y)), __builtin_fmaxf(fabsf(y.x), fabsf(y.y))),
                                        __builtin_fmaxf(fabsf(z.x), fabsf(z.y)));
    const bool near = reach <= r.reach_max;
    const float m = __builtin_fmaxf(fabsf(tmin), fabsf(tmax));
    const bool miss = tmin - tmax > __builtin_fmaf(m, 4e-6f, r.e2);
    const bool behind = tmax < -r.e2;
    const bool beyond = tmin > __builtin_fmaf((float)limit, 1.0f + 8e-6f, r.e2);  // (a limit beyond f32's range becomes +inf: no cull)
    return !((miss | behind | beyond) & near);
}

// Bounds::intersects (geometry/bounds.rs:94-128) decided in f32 whenever f32 can be SURE of the reference's answer.
// The reference's test is six f64 divisions (some 130 instructions as the compiler expands them); the same slab
// arithmetic in f32 -- packed, reciprocals instead of divisions -- is a third of that, and its result is the
// reference's whenever the two interval ends it compares are further apart than everything f32 may have lost:
//   * origin and box rounded to f32 (2^-24 of their magnitudes each), differences, reciprocal directions (2^-22
//     relative) and products: |t_f32 - t_reference| <= 2^-22 pmax imax + 2^-21 |t| with pmax the largest |coordinate|
//     of origin and box and imax the largest |1/d|; the room demanded is 2^-19 pmax imax + 8e-6 max(|tmin|, |tmax|),
//     eight and more times that;
//   * every |d| must exceed 1e-6: the reference then divides (its threshold for the `* INFINITY` branch is 1e-10) and
//     nothing here is near a special case; infinite bounds, overflow and NaN make a comparison false, and a false
//     comparison means "not sure".
// Returns true when sure, with the reference's answer in `hit`.  When some ray of a wave is not sure, the caller
// runs the reference's arithmetic (aabb_hit) for the wave.
__device__ __forceinline__ bool aabb_sure(const double (&bb)[6], V3 o, V3 d, bool &hit) {
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const bool regular = __builtin_fminf(__builtin_fminf(fabsf(dx), fabsf(dy)), fabsf(dz)) > 1e-6f;
    const float ix = __builtin_amdgcn_rcpf(dx), iy = __builtin_amdgcn_rcpf(dy), iz = __builtin_amdgcn_rcpf(dz);
    f32x2 bx, by, bz;
    bx.x = (float)bb[0], bx.y = (float)bb[3];
    by.x = (float)bb[1], by.y = (float)bb[4];
    bz.x = (float)bb[2], bz.y = (float)bb[5];
    const f32x2 tx = (bx - ox) * ix, ty = (by - oy) * iy, tz = (bz - oz) * iz;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fminf(tz.x, tz.y));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fmaxf(tz.x, tz.y));
    const float imax = __builtin_fmaxf(__builtin_fmaxf(fabsf(ix), fabsf(iy)), fabsf(iz));
    const float bmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(bx.x), fabsf(bx.y)), __builtin_fmaxf(fabsf(by.x), fabsf(by.y))),
                                       __builtin_fmaxf(fabsf(bz.x), fabsf(bz.y)));
    const float pmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)), bmax);
    const float room = __builtin_fmaf(__builtin_fmaxf(fabsf(tmin), fabsf(tmax)), 8e-6f, 0x1p-19f * pmax * imax);
    const float gap = tmax - tmin;
    hit = gap > room;            // tmin <= tmax beyond doubt
    const bool miss = -gap > room;  // tmin > tmax beyond doubt
    return regular & (hit | miss);
}

// The control words of a node record, moved to registers of their own: nothing of the record is needed once the ray
// has been transformed, so the next record can be loaded where this one lies.
// (readfirstlane on the operands: the asm wants scalar registers whatever the optimiser has come to believe about the
// record; on values that already are in scalar registers it costs nothing.  The optimiser takes what an asm statement
// returns for possibly divergent; these are wave-uniform.)
struct NodeWords {
    int tag, leaf, skip;
    double a0, a1;
    f32x2 f0, f1;  // the same 16 bytes read as four floats (a range-skip node's box: min.x, min.y, min.z, max.x)
};
__device__ __forceinline__ NodeWords node_words(const NodeRaw &raw) {
    auto pair = [](int lo, int hi) { return (long long)(((unsigned long long)(unsigned)uniform(hi) << 32) | (unsigned)uniform(lo)); };
    long long w0, w1, w2, w3;
    asm volatile("s_mov_b64 %0, %4\n\ts_mov_b64 %1, %5\n\ts_mov_b64 %2, %6\n\ts_mov_b64 %3, %7"
                 : "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3)
                 : "s"(pair(raw.b[8], raw.b[9])), "s"(pair(raw.b[10], raw.b[11])), "s"(pair(raw.b[12], raw.b[13])),
                   "s"(pair(raw.b[14], raw.b[15])));
    NodeWords w;
    w.tag = uniform((int)w0);
    w.leaf = uniform((int)w1);
    w.skip = uniform((int)(w1 >> 32));
    w.a0 = __builtin_bit_cast(double, w2);
    w.a1 = __builtin_bit_cast(double, w3);
    w.f0 = __builtin_bit_cast(f32x2, w2);
    w.f1 = __builtin_bit_cast(f32x2, w3);
    return w;
}

// Shape::local_intersect of the leaf kinds on the ray (lo, ld) in the leaf's own space.  Same dispatch twice: as a
// chain that reaches the sphere test after one comparison (the flat scenes of the book are spheres and planes: -2 %
// there, +30 % on a deep scene of cubes and cones), and as a switch, which the compiler turns into a balanced tree.
template <bool SPHERE_FIRST, bool STATS, class B>
__device__ __forceinline__ bool hit_leaf(int kind, int tag, V3 lo, V3 ld, double a0, double a1, bool on, int leaf, B &best) {
    const bool closed = (tag & RT_NODE_CLOSED) != 0;
    if constexpr (SPHERE_FIRST) {
        if (kind == RT_SPHERE) return hit_sphere(lo, ld, on, leaf, best);
        if (kind == RT_PLANE) return hit_plane(lo, ld, on, leaf, best);
        if (kind == RT_CUBE) return hit_cube(lo, ld, on, leaf, best);
        if (kind == RT_CYLINDER) return hit_cylinder<STATS, B>(lo, ld, a0, a1, closed, on, leaf, best);
        return hit_cone<STATS, B>(lo, ld, a0, a1, closed, on, leaf, best);
    } else {
        switch (kind) {
            case RT_SPHERE: return hit_sphere(lo, ld, on, leaf, best);
            case RT_PLANE: return hit_plane(lo, ld, on, leaf, best);
            case RT_CUBE: return hit_cube(lo, ld, on, leaf, best);
            case RT_CYLINDER: return hit_cylinder<STATS, B>(lo, ld, a0, a1, closed, on, leaf, best);
            default: return hit_cone<STATS, B>(lo, ld, a0, a1, closed, on, leaf, best);
        }
    }
}

// World::intersect (rendering/world.rs:27-36) over a node table WITHOUT groups or range-skip nodes: every record is a
// leaf of the world's top-level list.  The running index is wave-uniform.  Per node: take the ray into the node's
// space (this uses up the matrix), request the following record into the registers the matrix occupied, then run the
// shape test while that request is in flight.  Lanes that are not `active` run the tests too (there is no cheaper
// way to idle them); nobody reads their result.
template <bool STATS, class B>
__device__ __forceinline__ void trace_flat(const SceneK &S, V3 o, V3 d, bool active, B &best, Ctr<STATS> &ctr) {
    const int end = S.n_nodes;
    if (end <= 0) return;
    NodeRaw raw = node_fetch<false>(S.nodes, 0);
    // Consume one word of the first record here, so that its arrival is awaited before the loop and not at the top
    // of every iteration (where the wait would also cover the request for the following record).
    asm volatile("" ::"s"(raw.b[8]));
    const double dd = __builtin_fma(d.z, d.z, __builtin_fma(d.y, d.y, d.x * d.x)) * BALL_SHRINK;  // for misses_ball
    int i = 0;
    while (i < end) {
        const NodeWords w = node_words(raw);
        const int xform = uniform(raw.b[9]);  // only read before the next record is requested
        const int kind = w.tag & 0xff;
        bool visit = true;  // wave-uniform
        if constexpr (!STATS && !B::every) {  // (the counting kernel tests what the reference tests)
            if (xform == RT_XFORM_DIAGONAL_BALL) {
                visit = __ballot(active & !misses_ball(raw, o, d, dd)) != 0ull;
            } else if (kind == RT_PLANE && xform != RT_XFORM_GENERAL) {
                // the y components of the ray in the plane's space, as to_local computes them
                const double ky = dwords_to_double(raw.a[10], raw.a[11]);
                visit = __ballot(active & !plane_behind(ky * o.y + dwords_to_double(raw.a[14], raw.a[15]), ky * d.y)) != 0ull;
            }
        }
        NodeRaw nraw;
        RT_NODE_VISIT(2);
        if (visit) {
            V3 lo, ld;
            to_local(raw, xform, o, d, lo, ld);
            asm volatile("" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(ld.x), "+v"(ld.y), "+v"(ld.z));  // the transform is complete
            nraw = node_fetch<false>(S.nodes, i + 1);  // the table has one spare record at its end
            const bool any = hit_leaf<true, STATS, B>(kind, w.tag, lo, ld, w.a0, w.a1, true, w.leaf, best);
            if (STATS && active) {
                ctr.add(C_XFORM);
                ctr.add(C_PRIM_TESTS + kind);
                if (any) ctr.add(C_PRIM_HITS + kind);
            }
        } else {
            nraw = node_fetch<false>(S.nodes, i + 1);
        }
        // One definition of the closest hit per iteration: without it the optimiser carries two copies of (t, leaf)
        // round the loop -- the one the tests read and the one the caller gets -- and moves both at every join.
        pin(best);
        raw = nraw;
        i = uniform(i + 1);
    }
}

// trace_flat for the occlusion queries of the plain render kernel: the walk starts at record `start` and ends as soon as
// every ray of the wave has its occluder -- the answer does not depend on the order.  The caller names the record of
// the object its rays start on: a ray that starts INSIDE a closed object is occluded by that very object, so the shadow
// rays of a wave inside a glass ball (most rays of a scene with glass in it) are settled by one record instead of all.
// No wrap-round: a table of leaves lies in memory twice, one copy after the other (rt_scene_create).
template <bool STATS, class B>
__device__ __forceinline__ void trace_flat_from(const SceneK &S, V3 o, V3 d, bool active, B &best, Ctr<STATS> &ctr, int start) {
    if (S.n_nodes <= 0) return;
    int i = uniform(start >= 0 && start < S.n_nodes ? start : 0);
    int end = uniform(i + S.n_nodes);
    NodeRaw raw = node_fetch<false>(S.nodes, i);
    asm volatile("" ::"s"(raw.b[8]));  // as in trace_flat
    const double dd = __builtin_fma(d.z, d.z, __builtin_fma(d.y, d.y, d.x * d.x)) * BALL_SHRINK;  // for misses_ball
    while (i < end) {
        const NodeWords w = node_words(raw);
        const int xform = uniform(raw.b[9]);  // only read before the next record is requested
        const int kind = w.tag & 0xff;
        bool visit = true;  // wave-uniform
        // (a ray that has its answer asks for no record)
        if (xform == RT_XFORM_DIAGONAL_BALL) {
            visit = __ballot(active && !settled(best) && !misses_ball(raw, o, d, dd)) != 0ull;
        } else if (kind == RT_PLANE && xform != RT_XFORM_GENERAL) {
            const double ky = dwords_to_double(raw.a[10], raw.a[11]);
            visit = __ballot(active && !settled(best) && !plane_behind(ky * o.y + dwords_to_double(raw.a[14], raw.a[15]), ky * d.y)) != 0ull;
        }
        NodeRaw nraw;
        RT_NODE_VISIT(2);
        if (visit) {
            V3 lo, ld;
            to_local(raw, xform, o, d, lo, ld);
            asm volatile("" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(ld.x), "+v"(ld.y), "+v"(ld.z));  // the transform is complete
            nraw = node_fetch<false>(S.nodes, i + 1);
            hit_leaf<true, STATS, B>(kind, w.tag, lo, ld, w.a0, w.a1, true, w.leaf, best);
            if (__ballot(active && !settled(best)) == 0ull) end = i;  // every ray of the wave has its answer
        } else {
            nraw = node_fetch<false>(S.nodes, i + 1);
        }
        pin(best);  // as in trace_flat
        raw = nraw;
        i = uniform(i + 1);
    }
}

// Where a traversal parks the ray of an outer sibling list while it walks a group's children: the ray of nesting
// level 1 in the wave's LDS (field-major, one 8-byte column per lane: conflict-free), deeper levels in a per-lane
// array in scratch memory (indexed by the wave-uniform level, so a whole wave's access is one contiguous run).  The
// world-space ray (level 0) needs no parking: the caller's copy is alive throughout.
constexpr int MAX_NEST = RT_MAX_GROUP_NESTING;
struct ParkedRays {
    double *lds;                   // this lane's column of the wave's parking slot (6 x 64 doubles)
    double deep[MAX_NEST > 2 ? MAX_NEST - 2 : 1][6];
    __device__ __forceinline__ void park(int level, V3 o, V3 d) {  // level >= 1
        if (level == 1) {
            lds[0 * 64] = o.x; lds[1 * 64] = o.y; lds[2 * 64] = o.z;
            lds[3 * 64] = d.x; lds[4 * 64] = d.y; lds[5 * 64] = d.z;
        } else {
            double *e = deep[level - 2];
            e[0] = o.x; e[1] = o.y; e[2] = o.z; e[3] = d.x; e[4] = d.y; e[5] = d.z;
        }
    }
    __device__ __forceinline__ void fetch(int level, V3 &o, V3 &d) const {  // level >= 1
        if (level == 1) {
            o = mk(lds[0 * 64], lds[1 * 64], lds[2 * 64]);
            d = mk(lds[3 * 64], lds[4 * 64], lds[5 * 64]);
        } else {
            const double *e = deep[level - 2];
            o = mk(e[0], e[1], e[2]);
            d = mk(e[3], e[4], e[5]);
        }
    }
};

// World::intersect / collect_intersections (rendering/world.rs:27-78) over a node table with groups and range-skip
// nodes, as ONE loop over the table for every nesting level (the reference recurses, world.rs:66-68).  The table is
// in depth-first order and every record carries its nesting level, so the walk needs no stack of list ends: a record
// at a shallower level than the one being walked means the lists in between have ended.  State:
//   i, level          wave-uniform: the record being visited and the nesting level of the sibling list it belongs to
//   (co, cd)          the ray in the space of that list: the world ray taken through the local inverse of every
//                     enclosing group, one after the other (world.rs:53-58)
//   act               per lane: the deepest level at which the lane's ray has passed every enclosing group's bounds;
//                     the lane takes part in the tests of the current list when act == level
// Entering a group (ANY lane's ray passes its bounds) parks (co, cd) and continues with the group-space ray; leaving
// fetches the parked ray back.  `far`: hits beyond it do not matter to the caller (range-skip nodes use it; the t
// parameter is the same in every space because directions are never renormalised).  Per node as in trace_flat: ray
// into the node's space, next record requested over the current one, test while the request is in flight.
// CONES false: no range-skip node of the table has cones below it (a kernel built without the cone-direction code
// keeps fewer values across the record loop).
// EXIT (accumulators that can be settled: occlusion queries): the walk ends once every ray of the wave has its answer.
template <bool STATS, class B, bool CONES = true, bool EXIT = false>
__device__ __forceinline__ void trace_nested(const SceneK &S, V3 o, V3 d, bool active0, double far, B &best, Ctr<STATS> &ctr) {
    const int end = S.n_nodes;
    if (end <= 0) return;
    // The node table whose sibling lists are in front-to-back order for most rays of the wave: per axis, the sign
    // that the majority of the participating lanes' directions have.  Any table gives the same result (the closest
    // hit does not depend on the order objects are met in); this one finds it earliest.
    NodeP nodes = S.nodes;
    if constexpr (!STATS && !B::every) {
        const unsigned half = (unsigned)__popcll(__ballot(active0)) >> 1;
        const unsigned oct = ((unsigned)__popcll(__ballot(active0 && d.x < 0.0)) > half ? 1u : 0u) | ((unsigned)__popcll(__ballot(active0 && d.y < 0.0)) > half ? 2u : 0u) |
                             ((unsigned)__popcll(__ballot(active0 && d.z < 0.0)) > half ? 4u : 0u);
        const unsigned table = (unsigned)(S.oct_tables >> (8u * oct)) & 0xffu;
        nodes = (NodeP)((const char RT_CONSTANT_AS *)S.nodes + (size_t)table * S.table_bytes);
    }
    NodeRaw raw = node_fetch<true>(nodes, 0);
    asm volatile("" ::"s"(raw.b[8]));  // as in trace_flat
    ParkedRays parked;
    parked.lds = ray_stack_lds + S.park_base + (threadIdx.x & 63u);
    V3 co = o, cd = d;
    int level = 0;
    int act = active0 ? 0 : -1;
    SkipRay sray = {};  // for range-skip nodes: (co, cd) as floats, made at the first such node met after a change of level
    bool have_inv = false;
    double list_qdd = 0.0;  // d^T q d for the list being walked, once a node that shares the list's q has formed it
    bool have_q = false;
    int i = 0;
    while (i < end) {
        if constexpr (EXIT) {
            if (__ballot(active0 && !settled(best)) == 0ull) break;  // every ray of the wave has its answer
        }
        const NodeWords w = node_words(raw);
        const int xform = uniform(raw.b[9]);  // only read before the next record is requested
        const int kind = w.tag & 0xff;
        const int lvl = (w.tag >> RT_NODE_LEVEL_SHIFT) & RT_NODE_LEVEL_MASK;
        if (lvl < level) {  // wave-uniform: the list(s) being walked ended before this record
            act = act > lvl ? lvl : act;
            if (lvl == 0) {
                co = o;
                cd = d;
            } else {
                parked.fetch(lvl, co, cd);
            }
            level = lvl;
            have_inv = false, have_q = false;
        }
        const bool active = act == level;
        // Everything that reads the first 96 bytes of the record happens before the next record is requested.
        V3 lo = co, ld = cd;
        bool shortcut = false;  // a cone below this range-skip node takes the reference's linear shortcut for this ray
        double qdd = 0.0;
        if (kind == RT_SKIPBOX) {
            if constexpr (!STATS && !B::every) {
                if (CONES && xform == RT_SKIP_CONES) {
                    double m[7];
#pragma unroll
                    for (int k = 0; k < 7; ++k) m[k] = dwords_to_double(raw.a[2 * k], raw.a[2 * k + 1]);
                    const bool list_q = (w.tag & RT_NODE_LIST_Q) != 0;  // every such node of this list has this very q
                    if (list_q && have_q) {
                        qdd = list_qdd;
                    } else {
                        qdd = (m[0] * cd.x + m[1] * cd.y + m[2] * cd.z) * cd.x + (m[1] * cd.x + m[3] * cd.y + m[4] * cd.z) * cd.y +
                              (m[2] * cd.x + m[4] * cd.y + m[5] * cd.z) * cd.z;
                        if (list_q) {
                            list_qdd = qdd;
                            have_q = true;
                        }
                    }
                    shortcut = !(fabs(qdd) >= m[6]);
                }
            }
        } else {
            to_local(raw, xform, co, cd, lo, ld);
        }
        double bb[6];
        bb[0] = w.a0;
        bb[1] = w.a1;
#pragma unroll
        for (int k = 0; k < 4; ++k) bb[2 + k] = dwords_to_double(raw.c[2 * k], raw.c[2 * k + 1]);
        // The record to have ready next.  A range-skip node sends the traversal on to `skip` about every other time,
        // so there is nothing to guess from: its test (which needs nothing but the record and the ray) runs first and
        // the request goes out for the record that is certain to be next.  After every other kind of node the
        // following record is asked for before the test (the table has one spare record at its end); only a group
        // that no ray enters, one in twenty, makes that the wrong one.
        int guess = i + 1;
        if (kind == RT_SKIPBOX) {
            if constexpr (!STATS && !B::every) {  // the counting kernel visits everything the reference visits
                if (!have_inv) {
                    sray = skip_ray(co, cd);
                    have_inv = true;
                }
                // the box as floats: (min, max) per axis
                f32x2 bz;
                bz.x = __int_as_float(raw.c[0]);
                bz.y = __int_as_float(raw.c[1]);
                const bool may = active && !settled(best) && (shortcut || box_may_hit32(w.f0, w.f1, bz, sray, hit_limit(best, far)));
                ctr.visits += may ? 1u : 0u;
                if (__ballot(may) == 0ull) {
                    guess = w.skip;
                    if (w.tag & RT_NODE_LOOKAHEAD) {
                        // The record at `skip` is a range-skip node of this list whose test this record carries
                        // along: decide it now and go on behind it, or inside it, without fetching it.
                        f32x2 rx, ry, rz;
                        rx.x = __int_as_float(raw.a[14]), rx.y = __int_as_float(raw.a[15]);
                        ry.x = __int_as_float(raw.b[0]), ry.y = __int_as_float(raw.b[1]);
                        rz.x = __int_as_float(raw.b[2]), rz.y = __int_as_float(raw.b[3]);
                        const double rbound = dwords_to_double(raw.b[4], raw.b[5]);
                        const bool rshort = CONES && xform == RT_SKIP_CONES && !(fabs(qdd) >= rbound);
                        const bool rmay = active && !settled(best) && (rshort || box_may_hit32(rx, ry, rz, sray, hit_limit(best, far)));
                        ctr.visits += rmay ? 1u : 0u;
                        guess = __ballot(rmay) != 0ull ? w.skip + 1 : w.leaf;
                    }
                }
            }
        }
        asm volatile("" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(ld.x), "+v"(ld.y), "+v"(ld.z));  // the transform is complete
        NodeRaw nraw = node_fetch<true>(nodes, guess);
        int ni = guess;
        RT_NODE_VISIT(kind == RT_SKIPBOX ? 0 : kind == RT_GROUP ? 1 : 2);
        if (kind == RT_GROUP) {
            // Group: bounds in the group's own space against the ray in that space (world.rs:44-63)
            bool inside = false;
            if (active) {
                ctr.add(C_XFORM);
                ctr.add(C_AABB);
            }
            // in f32 when every ray's answer is beyond doubt (all but rays that graze the bounds), else as the reference does
            bool sure = true;
            if (active) sure = aabb_sure(bb, lo, ld, inside);
            if (__ballot(!sure) != 0ull) inside = active && aabb_hit(bb, lo, ld);
            if constexpr (!STATS) ctr.visits += inside ? 4u : 0u;
            ni = w.skip;
            if (__ballot(inside) != 0ull && w.skip != i + 1) {  // (a group without children has nothing to walk)
                if (level > 0) parked.park(level, co, cd);
                co = lo;
                cd = ld;
                level = level + 1;
                act = inside ? level : act;
                have_inv = false, have_q = false;
                ni = i + 1;
            }
        } else if (kind != RT_SKIPBOX) {
            const bool any = hit_leaf<false, STATS, B>(kind, w.tag, lo, ld, w.a0, w.a1, active, w.leaf, best);
            if (STATS && active) {
                ctr.add(C_XFORM);
                ctr.add(C_PRIM_TESTS + kind);
                if (any) ctr.add(C_PRIM_HITS + kind);
            }
        }
        if (ni != guess) {
            nraw = node_fetch<true>(nodes, ni);  // guessed wrong: fetch the record wanted
            RT_NODE_VISIT(3);
        }
        pin(best);  // as in trace_flat
        raw = nraw;
        i = uniform(ni);  // every operand above is wave-uniform; keep the index in an SGPR
    }
}

// NEST, the template parameter of the kernels: 0 = a table of leaves only; 1 = anything; 2 = a table with groups or
// range-skip nodes none of which has cones below it (render_kernel only: the kernel then carries no cone-direction
// code, and spills a quarter fewer registers in the record loop -- chapter_14 at 4K 5.34 -> 5.09 ms).
constexpr bool nest_cones(int nest) { return nest != 2; }
template <int NEST, bool STATS, bool TIES = ANY_ORDER<NEST>>
__device__ __forceinline__ Closest<TIES> trace(const SceneK &S, V3 o, V3 d, bool active, double far, Ctr<STATS> &ctr) {
    Closest<TIES> best;
    best.t = __builtin_nan("");
    best.leaf = -1;
    if (STATS && active) ctr.add(C_RAYS);
    if constexpr (NEST == 0) trace_flat<STATS>(S, o, d, active, best, ctr);
    else trace_nested<STATS, Closest<TIES>, nest_cones(NEST)>(S, o, d, active, far, best, ctr);
    return best;
}

// World::is_shadowed's question about the ray (o, d): a hit with 0 <= t < distance?
#ifndef RT_OWN_SHADOW
#define RT_OWN_SHADOW true
#endif
#ifndef RT_FLAT_SHADOW_ROTATE
#define RT_FLAT_SHADOW_ROTATE true
#endif
// `start` (flat tables): the record of the object the wave's rays start on, or -1.
template <int NEST, bool STATS, bool EXIT = false>
__device__ __forceinline__ bool occluded(const SceneK &S, V3 o, V3 d, bool active, double distance, Ctr<STATS> &ctr, int start = -1) {
    Occluded oc;
    oc.far = distance;
    oc.t = __builtin_nan("");
    if (STATS && active) ctr.add(C_RAYS);
    if constexpr (NEST == 0) {
        if constexpr (RT_FLAT_SHADOW_ROTATE && EXIT && !STATS) trace_flat_from<STATS>(S, o, d, active, oc, ctr, start);
        else trace_flat<STATS>(S, o, d, active, oc, ctr);
    }
    else trace_nested<STATS, Occluded, nest_cones(NEST), EXIT>(S, o, d, active, distance, oc, ctr);
    return oc.t < distance;
}

// Shape::local_normal_at of the leaf kinds
__device__ __forceinline__ V3 local_normal_at(int kind, double minimum, double maximum, V3 p) {
    switch (kind) {
        case RT_SPHERE:  // geometry/sphere.rs:77-80
            return p;
        case RT_PLANE:  // geometry/planes.rs:58-60
            return mk(0.0, 1.0, 0.0);
        case RT_CUBE: {  // geometry/cubes.rs:88-98
            double maxc = fmax(fmax(fabs(p.x), fabs(p.y)), fabs(p.z));
            if (fabs(maxc - fabs(p.x)) < EPS) return mk(p.x, 0.0, 0.0);
            if (fabs(maxc - fabs(p.y)) < EPS) return mk(0.0, p.y, 0.0);
            return mk(0.0, 0.0, p.z);
        }
        case RT_CYLINDER: {  // geometry/cylinders.rs:101-111
            double dist = p.x * p.x + p.z * p.z;
            if (dist < 1.0 && p.y >= maximum - EPS) return mk(0.0, 1.0, 0.0);
            if (dist < 1.0 && p.y <= minimum + EPS) return mk(0.0, -1.0, 0.0);
            return mk(p.x, 0.0, p.z);
        }
        default: {  // geometry/cones.rs:109-126
            double dist = p.x * p.x + p.z * p.z;
            if (dist < 1.0 && p.y >= maximum - EPS) return mk(0.0, 1.0, 0.0);
            if (dist < 1.0 && p.y <= minimum + EPS) return mk(0.0, -1.0, 0.0);
            double y = sqrt(p.x * p.x + p.z * p.z);
            if (p.y > 0.0) y = -y;
            return mk(p.x, y, p.z);
        }
    }
}

// Object::normal_at (rendering/objects.rs:171-184)
// `object_point` = world_to_object(world_point) (objects.rs:87-91), which Pattern::color_at_object needs again.
// DIAGONAL: leave out the products with the exact zeros of a scale-and-translate inverse, as the ray transform does
// (at most the sign of a zero differs).  The choice is per lane; it pays where a wave's hits are mostly on such
// objects -- the flat scenes of the book -- and costs where rotated and unrotated leaves mix.
template <bool DIAGONAL = false>
__device__ __forceinline__ V3 normal_at(const RtSurface *s, V3 world_point, V3 &object_point) {
    const double *m = s->inv_world;
    V3 lp, wn;
    if (DIAGONAL && s->xform != RT_XFORM_GENERAL) {
        lp = mk(m[0] * world_point.x + m[3], m[5] * world_point.y + m[7], m[10] * world_point.z + m[11]);
        object_point = lp;
        V3 ln = local_normal_at(s->kind, s->minimum, s->maximum, lp);
        wn = mk(m[0] * ln.x, m[5] * ln.y, m[10] * ln.z);
    } else {
        lp = mul_point(m, world_point);
        object_point = lp;
        V3 ln = local_normal_at(s->kind, s->minimum, s->maximum, lp);
        wn = mul_vector_transposed(m, ln);
    }
    return div_by_magnitude(wn, magnitude(wn));
}

// Pattern::color_at (scene/patterns.rs:81-135); `%` on f64 is fmod, `as i32` saturates (v_cvt_i32_f64 does too)
__device__ __forceinline__ V3 pattern_color(const RtSurface *s, V3 p) {
    V3 a = mk(s->pat_a[0], s->pat_a[1], s->pat_a[2]);
    V3 b = mk(s->pat_b[0], s->pat_b[1], s->pat_b[2]);
    switch (s->pattern_kind) {
        case RT_PATTERN_STRIPE:
            return (fmod(floor(p.x), 2.0) == 0.0) ? a : b;
        case RT_PATTERN_GRADIENT: {
            V3 distance = b - a;
            double f = p.x;
            if (f < 0.0) f = 0.0; else if (f > 1.0) f = 1.0;
            return a + distance * f;
        }
        case RT_PATTERN_RING:
            return (fmod(floor(sqrt(p.x * p.x + p.z * p.z)), 2.0) == 0.0) ? a : b;
        default: {
            unsigned x = (unsigned)__double2int_rz(floor(p.x + CHECKER_EPS));
            unsigned y = (unsigned)__double2int_rz(floor(p.y + CHECKER_EPS));
            unsigned z = (unsigned)__double2int_rz(floor(p.z + CHECKER_EPS));
            return (((x + y + z) & 1u) == 0u) ? a : b;
        }
    }
}

// x^n for a small non-negative integer n by binary powering: at most 2*12 multiplications, relative error below
// n * 2^-53.  Phong exponents are such integers in every scene file; the general pow is kept for the rest.
__device__ __forceinline__ double powi_small(double x, unsigned n) {
    double r = 1.0;
    while (n != 0u) {
        if (n & 1u) r = r * x;
        x = x * x;
        n >>= 1;
    }
    return r;
}
constexpr double POWI_MAX = 4096.0;
// The general case, out of line: inlined, its two dozen polynomial constants are hoisted into registers that the
// whole render loop then has to carry (or spill).
__device__ __attribute__((noinline)) double pow_general(double x, double y) { return pow(x, y); }

// Material::lighting (scene/materials.rs:122-172) with Pattern::color_at_object (scene/patterns.rs:137-146), evaluated
// BEFORE the shadow ray is traced: returns the result for in_shadow = true in `shadowed` (ambient only: the reference
// adds black diffuse and specular terms, which changes nothing) and the result for in_shadow = false in `lit`.
// `flags` bit 0: the diffuse branch is taken when not in shadow; bit 1: powf is evaluated then.
// need_lit false: the caller knows the point to be in shadow; `lit` is then not computed (it comes back as `shadowed`).
template <bool STATS>
__device__ __forceinline__ void lighting(const RtSurface *s, const RtLight &light, V3 point, V3 object_point, V3 eye,
                                         V3 normal, V3 &shadowed, V3 &lit, int &flags, Ctr<STATS> &ctr, bool need_lit = true) {
    V3 color;
    if (s->pattern_kind != RT_PATTERN_NONE) {
        V3 pattern_point = mul_point(s->pat_inv, object_point);
        color = pattern_color(s, pattern_point);
        ctr.add(C_PATTERNS);
    } else {
        color = mk(s->color[0], s->color[1], s->color[2]);
    }
    V3 li = mk(light.intensity[0], light.intensity[1], light.intensity[2]);
    V3 lp = mk(light.position[0], light.position[1], light.position[2]);
    V3 effective = mk(color.x * li.x, color.y * li.y, color.z * li.z);
    V3 ambient = effective * s->ambient;
    shadowed = ambient;
    lit = ambient;
    flags = 0;
    if (!need_lit) return;
    V3 lv = lp - point;
    lv = div_by_magnitude(lv, magnitude(lv));
    double ldn = dot(lv, normal);
    if (ldn >= 0.0) {
        flags = 1;
        V3 diffuse = (effective * s->diffuse) * ldn;
        V3 specular = mk(0.0, 0.0, 0.0);
        V3 rv = reflect(-lv, normal);
        double rde = dot(rv, eye);
        if (rde > 0.0) {
            flags = 3;
            const double shininess = s->shininess;
            double factor;
            const bool small_int = shininess >= 0.0 && shininess <= POWI_MAX && floor(shininess) == shininess;
            if (small_int) factor = powi_small(rde, (unsigned)shininess);
            else factor = pow_general(rde, shininess);
            specular = (li * s->specular) * factor;
        }
        lit = (ambient + diffuse) + specular;
    }
}

// Camera::ray_for_pixel (rendering/camera.rs:39-54) with the inverse of the view transform precomputed
__device__ __forceinline__ void ray_for_pixel(const RtCameraDev &cam, unsigned px, unsigned py, V3 &origin, V3 &dir) {
    double xo = ((double)px + 0.5) * cam.pixel_size;
    double yo = ((double)py + 0.5) * cam.pixel_size;
    double wx = cam.half_width - xo;
    double wy = cam.half_height - yo;
    V3 pixel = mul_point(cam.inv, mk(wx, wy, -1.0));
    origin = mk(cam.origin[0], cam.origin[1], cam.origin[2]);
    V3 v = pixel - origin;
    dir = div_by_magnitude(v, magnitude(v));
}

// World::color_at (rendering/world.rs:113-126) for one ray per lane.  Returns the colour in `acc`.
//
// The reference's recursion  color = surface + reflected*k_r + refracted*k_t  (world.rs:101-110, with
// reflected = child*reflectivity, world.rs:167, and the Schlick factors when both are positive) is evaluated as a
// sum over the nodes of the ray tree of  weight(node) * surface(node), weight being the product of the factors on
// the path from the root.  That differs from the nested form only by floating-point reassociation (a few ulp).
//
// One round of the loop takes every lane's current radiance ray through: traversal; for the lanes that hit, shading
// for both outcomes of the shadow test and the reflection and refraction rays (pushed on a per-lane stack: the one with
// the shorter ray tree to expect last, so that it is walked first); traversal of the shadow rays; choice between the two sums;
// next ray from the stack.  All lanes of a wave are thus always in the same phase: the two traversals never mix
// scattered radiance rays with the coherent shadow rays, and the shading code runs once per round.  A lane whose
// ray missed simply sits out the shadow traversal.
// What a hit contributes and what it spawns: Intersection::prepare_computations (rendering/intersections.rs:45-63),
// Material::lighting for both outcomes of the shadow test (materials.rs:122-172), the reflection and refraction rays
// of World::reflected_color / refracted_color with their path weights and Schlick's blend (world.rs:105-108, 155-198,
// intersections.rs:123-136), and the shadow ray of World::is_shadowed (world.rs:128-141).  `w` is the path weight of
// the ray that hit, `remaining` its recursion budget.
struct HitShade {
    V3 lit, shadowed;          // surface colour if the light reaches the point / if it does not (not yet weighted)
    int flags;                 // (counting kernel) bit 0: the lit colour has a diffuse term, bit 1: a specular one
    bool reflect, refract;     // which children the hit spawns
    bool from_inside;
    bool own_shadow;           // the hit is known to lie in its own object's shadow: `lit` is not computed and no shadow ray need be walked
    V3 over, under;            // over_point, under_point (intersections.rs:60-63): where the children start
    V3 normal;                 // at the hit, turned towards the eye: the reflection's direction is reflect(rd, normal) (intersections.rs:62)
    V3 refract_dir;
    double w_reflect, w_refract;
    V3 shadow_dir;             // from `over` towards the light, normalised
    double light_distance;
};
template <bool DIAGONAL, bool STATS, bool OWN_SHADOW = false>
__device__ __forceinline__ void shade_hit(const RtLight &light, V3 light_pos, const RtSurface *s, V3 ro, V3 rd, double t, double w, int remaining,
                                          HitShade &h, Ctr<STATS> &ctr) {
    V3 point = ro + rd * t;
    V3 object_point;
    V3 normal = normal_at<DIAGONAL>(s, point, object_point);
    V3 eye = -rd;
    h.from_inside = dot(normal, eye) < 0.0;
    if (h.from_inside) normal = -normal;
    h.over = point + normal * EPS;
    ctr.add(C_SHADES);
    h.flags = 0;
    h.own_shadow = false;
    if constexpr (OWN_SHADOW) {
        // A hit from inside a sphere that the light is well outside of (RT_SURFACE_LIGHT_OUTSIDE): the shadow ray starts
        // strictly inside the sphere -- c = |o'|^2 - 1 of the reference's quadratic (sphere.rs:51-74) is below -1e-6 here,
        // and the reference's own o' (the ray taken through the local inverses of the enclosing groups, world.rs:53-58)
        // differs from this one by less than 1e-8 (rt_flatten.cpp), so its c is negative too, by far more than the
        // rounding of b^2 - 4ac: the quadratic has a root t2 > 0 -- and the light lies at least as far beyond the
        // sphere's surface as half its longest chord, so t2 <= 2/3 of the light's distance: World::is_shadowed
        // (world.rs:128-149) answers true whatever else the ray would meet.  Neither the lit colour nor the shadow
        // walk is needed for such a hit (Material::lighting returns the ambient term alone for a point in shadow).
        if (h.from_inside && (s->flags & RT_SURFACE_LIGHT_OUTSIDE) != 0) {
            const double *m = s->inv_world;
            const V3 p = s->xform != RT_XFORM_GENERAL ? mk(m[0] * h.over.x + m[3], m[5] * h.over.y + m[7], m[10] * h.over.z + m[11]) : mul_point(m, h.over);
            h.own_shadow = dot(p, p) - 1.0 < -1e-6;
        }
    }
    lighting<STATS>(s, light, point, object_point, eye, normal, h.shadowed, h.lit, h.flags, ctr, !h.own_shadow);

    const double reflectivity = s->reflectivity;
    const double transparency = s->transparency;
    h.reflect = remaining != 0 && reflectivity != 0.0;   // world.rs:156-162
    h.refract = remaining != 0 && transparency != 0.0;   // world.rs:175-181
    double cos_i = dot(eye, normal);
    // n1/n2 as Intersection::prepare_computations leaves them when handed the one-element list [self]
    // (intersections.rs:41-43,66-103): n1 = 1.0, n2 = the hit object's refractive index; a non-finite t
    // fails float_equal(t, t) and leaves both at 1.0.
    const double n1 = 1.0;
    const double n2 = (fabs(t - t) < EPS) ? s->refractive_index : 1.0;
    h.refract_dir = mk(0.0, 0.0, 0.0);
    if (h.refract) {  // world.rs:183-194
        ctr.add(C_REFRACT_SETUPS);
        double n_ratio = n1 / n2;
        double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (sin2_t > 1.0) {
            h.refract = false;
        } else {
            double cos_t = sqrt(1.0 - sin2_t);
            h.refract_dir = normal * (n_ratio * cos_i - cos_t) - eye * n_ratio;
        }
    }
    h.w_reflect = w * reflectivity;
    h.w_refract = w * transparency;
    if (reflectivity > 0.0 && transparency > 0.0) {  // world.rs:105-108, intersections.rs:123-136
        ctr.add(C_SCHLICK);
        double c = cos_i;
        double reflectance;
        bool total = false;
        if (n1 > n2) {
            double n = n1 / n2;
            double sin2_t = (n * n) * (1.0 - c * c);
            if (sin2_t > 1.0) total = true;
            c = sqrt(1.0 - sin2_t);
        }
        if (total) {
            reflectance = 1.0;
        } else {
            double q = (n1 - n2) / (n1 + n2);
            double r0 = q * q;
            double m = 1.0 - c;
            double m2 = m * m;
            double m5 = m * (m2 * m2);  // powi(5)
            reflectance = r0 + (1.0 - r0) * m5;
        }
        h.w_reflect = h.w_reflect * reflectance;
        h.w_refract = h.w_refract * (1.0 - reflectance);
    }
    h.normal = normal;
    h.under = point - normal * EPS;
    h.light_distance = 0.0;
    h.shadow_dir = rd;
    if (!h.own_shadow) {  // (no shadow ray is walked for a hit that has its answer)
        V3 v = light_pos - h.over;  // World::is_shadowed (rendering/world.rs:128-141)
        h.light_distance = magnitude(v);
        h.shadow_dir = div_by_magnitude(v, h.light_distance);
    }
    ctr.add(C_SHADOW);
}

__device__ __forceinline__ double readlane_double(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
#ifndef RT_SHARE_MIN_IDLE
#define RT_SHARE_MIN_IDLE 24
#endif
constexpr int SHARE_MIN_IDLE = RT_SHARE_MIN_IDLE;

template <int NEST, bool STATS, bool TIES = ANY_ORDER<NEST>>
__device__ __forceinline__ void radiance(const SceneK &S, bool valid, V3 ro, V3 rd, int remaining0, V3 &acc,
                                         Ctr<STATS> &ctr) {
    bool have = valid;  // (ro, rd, w, remaining) is a radiance ray to trace
    double w = 1.0;
    int remaining = remaining0;
    acc = mk(0.0, 0.0, 0.0);
    // The kernels for scenes with groups share rays between lanes: a lane whose own pixel has no ray left takes a
    // pending ray off another lane's stack (below), so a ray carries the lane its colour belongs to.
    constexpr bool SHARE = NEST > 0 || RT_SHARE_FLAT;
    constexpr int SHARE_MAX = NEST > 0 ? 48 : 16;  // rays handed over per round: 8 values x 48 fill the 6 x 64 doubles that park traversal rays
    const int lane = (int)(threadIdx.x & 63u);
    int owner = lane;

    // Pending rays (origin, direction, weight, remaining budget), last in first out.  The newest one lives in
    // registers (nx_*, valid while nx_rem >= 0); below it, the first LDS_LEVELS entries live in the workgroup's LDS
    // (field-major, one 8-byte bank column per lane: conflict-free) and only deeper ones in a per-lane array in
    // scratch memory -- scratch is interleaved across the wave's lanes, so the few lanes that push at any time
    // dirty whole cache lines, and those lines travel to HBM.  A pop hands out the register copy and at once asks
    // for the entry below it, so that load completes during the next traversal instead of being waited for.
    V3 nx_o = mk(0, 0, 0), nx_d = mk(0, 0, 0);
    double nx_w = 0.0;
    int nx_rem = -1;
    double *const lds = ray_stack_lds + S.stack_base + (threadIdx.x & 63u);
    constexpr int LDS_LEVELS = rt::LDS_LEVELS<NEST>;
    double stk[STACK_MAX - LDS_LEVELS][7];
    int stk_rem[STACK_MAX - LDS_LEVELS];
    int sp = 0;
    // The kernels for scenes with groups keep no register copy of the newest entry: their traversal needs the fifteen
    // registers more than a pop needs to be free of a memory round trip (with the copy: 21 spilled VGPRs, 1-2 % slower).
    constexpr bool REG_TOP = NEST == 0;
    auto push = [&](V3 po, V3 pd, double pw, int prem) {
        if constexpr (SHARE) prem |= owner << 8;
        if constexpr (!REG_TOP) {
            nx_o = po;
            nx_d = pd;
            nx_w = pw;
            nx_rem = prem;
        }
        if (!REG_TOP || nx_rem >= 0) {
            if (sp < LDS_LEVELS) {
                double *e = lds + sp * (8 * 64);
                e[0 * 64] = nx_o.x; e[1 * 64] = nx_o.y; e[2 * 64] = nx_o.z;
                e[3 * 64] = nx_d.x; e[4 * 64] = nx_d.y; e[5 * 64] = nx_d.z;
                e[6 * 64] = nx_w;
                e[7 * 64] = __hiloint2double(0, nx_rem);
            } else {
                const int d = sp - LDS_LEVELS;
                stk[d][0] = nx_o.x; stk[d][1] = nx_o.y; stk[d][2] = nx_o.z;
                stk[d][3] = nx_d.x; stk[d][4] = nx_d.y; stk[d][5] = nx_d.z;
                stk[d][6] = nx_w;
                stk_rem[d] = nx_rem;
            }
            sp++;
        }
        if constexpr (REG_TOP) {
            nx_o = po;
            nx_d = pd;
            nx_w = pw;
            nx_rem = prem;
        }
    };

    const V3 light_pos = mk(S.light.position[0], S.light.position[1], S.light.position[2]);

#ifdef RT_TIMING
    long long tm_trace = 0, tm_shade = 0, tm_iters = 0, tm_lanes = 0, tm_hist = 0;
    const long long tm_begin = clock64();
    const long long tm_rbegin = wall_clock64();
    if (threadIdx.x < 4) rt_wave_nodes[threadIdx.x] = 0;
#endif
    while (__ballot(have) != 0ull) {
        RT_TIC(tm0);
        const Closest<TIES> best = trace<NEST, STATS, TIES>(S, ro, rd, have, (double)INFINITY, ctr);
        RT_TIC(tm1);
        const bool shade = have && best.leaf >= 0 && S.light.has_light;  // else black: world.rs:124, world.rs:86-89
        V3 acc_lit = acc;         // the sum if the hit is not in shadow (acc: if it is)
        V3 term = mk(0.0, 0.0, 0.0);  // (SHARE) this ray's term of its owner's sum
        V3 sro = ro, srd = rd;    // the shadow ray
        double h_dist = 0.0;
        int h_flags = 0;
        bool own_shadow = false;
        if (shade) {
            HitShade h;
            shade_hit<NEST == 0, STATS, RT_OWN_SHADOW && !STATS>(S.light, light_pos, S.surfaces + best.leaf, ro, rd, best.t, w, remaining, h, ctr);
            h_flags = h.flags;
            own_shadow = h.own_shadow;
            if constexpr (SHARE) {  // the two candidate terms; the sum is formed in the owner's lane, below
                acc_lit = h.lit * w;
                term = h.shadowed * w;
            } else {
                acc_lit = acc + h.lit * w;
                acc = acc + h.shadowed * w;
            }
            const int child_remaining = remaining - 1;
            // The child pushed last is walked first.  For a hit from outside that is the reflection; for a hit from
            // INSIDE an object the refraction: it leaves the object and its ray tree is short, while the reflection
            // stays inside and goes on spawning exit rays, one pending per bounce.  Walking the short tree first
            // keeps the stack of pending rays shallow: on the 4K chapter_11 frame the bytes that pending rays below
            // the LDS levels write to scratch memory fall from 0.61 GB to 0.21 GB per frame, and the frame is 1 %
            // faster.  The order in which a pixel's terms are summed changes with it, the terms do not.
            const bool reflect_last = h.reflect && !h.from_inside;
            if (h.reflect && !reflect_last) {
                ctr.add(C_REFLECT);
                push(h.over, reflect(rd, h.normal), h.w_reflect, child_remaining);
            }
            if (h.refract) {
                ctr.add(C_REFRACT);
                push(h.under, h.refract_dir, h.w_refract, child_remaining);
            }
            if (reflect_last) {
                ctr.add(C_REFLECT);
                push(h.over, reflect(rd, h.normal), h.w_reflect, child_remaining);
            }
            h_dist = h.light_distance;
            sro = h.over;
            srd = h.shadow_dir;
        }
        RT_TIC(tm2);
        if (__ballot(shade) != 0ull) {
            const bool walk = shade && !own_shadow;  // the shadow rays that need a traversal
            bool in_shadow = own_shadow;
            if (__ballot(walk) != 0ull) {
                // (flat tables hold the objects in the reference's order, one record per leaf: record = leaf number)
                int start = -1;
                if constexpr (NEST == 0 && !STATS) start = __builtin_amdgcn_readlane(best.leaf, __builtin_ctzll(__ballot(walk)));
                in_shadow = occluded<NEST, STATS, NEST == 0>(S, sro, srd, walk, h_dist, ctr, start) || own_shadow;  // world.rs:145-148
            }
            if (shade && !in_shadow) {
                if constexpr (SHARE) term = acc_lit;
                else acc = acc_lit;
                if (STATS) {
                    if (h_flags & 1) ctr.add(C_LIT);
                    if (h_flags & 2) ctr.add(C_SPECULAR);
                }
            }
            if constexpr (SHARE) {
                // Every term goes to the sum of the lane that owns its ray: a lane's own first, then the others' in
                // lane order, one at a time -- a fixed order, so a frame comes out the same every time.
                if (shade && owner == lane) acc = acc + term;
                unsigned long long foreign = __ballot(shade && owner != lane);
                while (foreign != 0ull) {
                    const int j = __builtin_ctzll(foreign);
                    foreign &= foreign - 1ull;
                    const int to = __builtin_amdgcn_readlane(owner, j);
                    const V3 t = mk(readlane_double(term.x, j), readlane_double(term.y, j), readlane_double(term.z, j));
                    if (lane == to) acc = acc + t;
                }
            }
        }
        RT_TIC(tm3);
        // the lane's next ray
        if constexpr (!REG_TOP) nx_rem = -1;
        have = REG_TOP ? nx_rem >= 0 : sp > 0;
        if (have) {
            if constexpr (REG_TOP) {
                ro = nx_o;
                rd = nx_d;
                w = nx_w;
                remaining = nx_rem;
                nx_rem = -1;
            }
            if (sp > 0) {
                sp--;
                if (sp < LDS_LEVELS) {
                    const double *e = lds + sp * (8 * 64);
                    nx_o = mk(e[0 * 64], e[1 * 64], e[2 * 64]);
                    nx_d = mk(e[3 * 64], e[4 * 64], e[5 * 64]);
                    nx_w = e[6 * 64];
                    nx_rem = __double2loint(e[7 * 64]);
                } else {
                    const int d = sp - LDS_LEVELS;
                    nx_o = mk(stk[d][0], stk[d][1], stk[d][2]);
                    nx_d = mk(stk[d][3], stk[d][4], stk[d][5]);
                    nx_w = stk[d][6];
                    nx_rem = stk_rem[d];
                }
            }
            if constexpr (!REG_TOP) {
                ro = nx_o;
                rd = nx_d;
                w = nx_w;
                remaining = nx_rem;
                nx_rem = -1;
            }
            if constexpr (SHARE) {
                owner = remaining >> 8;
                remaining &= 0xff;
            }
        }
        if constexpr (SHARE) {
            // Lanes left without a ray take one from lanes that still hold pending ones: the k-th idle lane gets
            // the top of the k-th such lane's stack, handed over through the LDS that parks traversal rays (idle
            // between traversals).  An idle lane's stack is empty, so the ray it receives starts a stack of at most
            // STACK_MAX entries there like any other ray.  Which lane gets which ray depends only on the shape of
            // the ray trees, so the counting kernel and the plain one make the same moves.
            const unsigned long long idle = __ballot(!have);
            const unsigned long long rich = __ballot(sp > 0);
            if (__popcll(idle) >= SHARE_MIN_IDLE && rich != 0ull) {
                int n = __popcll(idle) < __popcll(rich) ? __popcll(idle) : __popcll(rich);
                n = n < SHARE_MAX ? n : SHARE_MAX;
                const int give = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(rich >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rich, 0u));
                const int take = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                double *mail = ray_stack_lds + S.park_base;  // [8][SHARE_MAX]
                if (sp > 0 && give < n) {
                    sp--;
                    V3 go, gd;
                    double gw;
                    int grem;
                    if (sp < LDS_LEVELS) {
                        const double *e = lds + sp * (8 * 64);
                        go = mk(e[0 * 64], e[1 * 64], e[2 * 64]);
                        gd = mk(e[3 * 64], e[4 * 64], e[5 * 64]);
                        gw = e[6 * 64];
                        grem = __double2loint(e[7 * 64]);
                    } else {
                        const int d = sp - LDS_LEVELS;
                        go = mk(stk[d][0], stk[d][1], stk[d][2]);
                        gd = mk(stk[d][3], stk[d][4], stk[d][5]);
                        gw = stk[d][6];
                        grem = stk_rem[d];
                    }
                    double *m = mail + give;
                    m[0 * SHARE_MAX] = go.x; m[1 * SHARE_MAX] = go.y; m[2 * SHARE_MAX] = go.z;
                    m[3 * SHARE_MAX] = gd.x; m[4 * SHARE_MAX] = gd.y; m[5 * SHARE_MAX] = gd.z;
                    m[6 * SHARE_MAX] = gw;
                    m[7 * SHARE_MAX] = __hiloint2double(0, grem);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (!have && take < n) {
                    const double *m = mail + take;
                    ro = mk(m[0 * SHARE_MAX], m[1 * SHARE_MAX], m[2 * SHARE_MAX]);
                    rd = mk(m[3 * SHARE_MAX], m[4 * SHARE_MAX], m[5 * SHARE_MAX]);
                    w = m[6 * SHARE_MAX];
                    const int meta = __double2loint(m[7 * SHARE_MAX]);
                    owner = meta >> 8;
                    remaining = meta & 0xff;
                    have = true;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the parked-ray area is free again before the next traversal writes it
            }
        }
#ifdef RT_TIMING
        {
            const long long tm4 = clock64();
            tm_trace += (tm1 - tm0) + (tm3 - tm2);
            tm_shade += (tm2 - tm1) + (tm4 - tm3);
            tm_iters += 1;
            tm_lanes += __popcll(__ballot(have));
            { const int n_ = __popcll(__ballot(have)); tm_hist += (n_ <= 8 ? 1ll : 0ll) + (n_ <= 16 ? (1ll << 16) : 0ll) + (n_ <= 32 ? (1ll << 32) : 0ll) + (n_ <= 48 ? (1ll << 48) : 0ll); }
        }
#endif
    }
#ifdef RT_TIMING
    if ((threadIdx.x & 63) == 0) {
        const unsigned wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wv < RT_TIMING_WAVES) {
            g_rt_timing[wv][0] = (unsigned long long)tm_trace | ((unsigned long long)tm_shade << 32);
            g_rt_timing[wv][1] = (unsigned long long)tm_rbegin;
            g_rt_timing[wv][2] = (unsigned long long)tm_iters | ((unsigned long long)tm_lanes << 32);
            g_rt_timing[wv][3] = (unsigned long long)(clock64() - tm_begin) | ((unsigned long long)(wall_clock64() - tm_rbegin) << 40);
            g_rt_timing[wv][4] = (unsigned long long)rt_wave_nodes[0] | ((unsigned long long)rt_wave_nodes[1] << 32);
            g_rt_timing[wv][5] = (unsigned long long)rt_wave_nodes[2] | ((unsigned long long)rt_wave_nodes[3] << 32);
            g_rt_timing[wv][7] = (unsigned long long)tm_hist;
        }
    }
#endif
}

template <bool STATS>
__device__ __forceinline__ void flush_counters(Ctr<STATS> &, unsigned long long *) {}
template <>
__device__ __forceinline__ void flush_counters<true>(Ctr<true> &ctr, unsigned long long *out) {
    for (int i = 0; i < RT_NUM_COUNTERS; ++i) {
        unsigned long long v = ctr.c[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(out + i, v);
    }
}

template <int NEST, bool RADIANCE>
__device__ __forceinline__ SceneK scene_k(const DeviceScene &d, int max_depth) {
    SceneK S;
    S.nodes = (NodeP)d.nodes;
    S.surfaces = d.surfaces;
    S.n_nodes = d.n_nodes;
    S.max_depth = max_depth;
    S.stack_base = (unsigned)uniform((int)(threadIdx.x >> 6)) * LDS_WAVE_DOUBLES<NEST, RADIANCE>;
    S.park_base = S.stack_base + (RADIANCE ? LDS_LEVELS<NEST> * 8 * 64 : 0);
    S.table_bytes = d.table_records * (unsigned)sizeof(RtNode);
    S.oct_tables = d.oct_tables;
    S.light = d.light;
    return S;
}

// convert_canvas_color_value_to_decimal_rgb_value (rendering/canvas.rs:123-130): (value * 255.0).round() as u32, then
// min(255).  `round` is half-away-from-zero; the Rust cast saturates and maps NaN to 0.
__device__ __forceinline__ unsigned quantize_value(double v) {
    const double r = round(v * 255.0);
    return !(r > 0.0) ? 0u : r >= 255.0 ? 255u : (unsigned)r;
}

// The frame is cut into 8x8 pixel tiles, one per wave, numbered row-major over the rows being rendered.
//
// A launch hands its tiles out in order of expected cost, most expensive first, so that its last waves are cheap
// ones.  The estimate comes from the ray through the tile's centre (classify_tiles_kernel); tiles are then counting-
// sorted by the quarter-octave of the estimate (order_count_kernel, order_scatter_kernel).  Scratch layout, in words:
//   [0, 128)                 tiles per class           (cleared by classify_tiles_kernel)
//   [128, 256)               tiles placed so far, per class (cleared likewise)
//   [256 + 0*n, 256 + 1*n)   the tiles in launch order
//   [256 + 1*n, 256 + 2*n)   cost estimate per tile
//   [256 + 2*n, 256 + 3*n)   class per tile
enum { TILE_CLASSES = 128, TILE_HEADER = 2 * TILE_CLASSES };
enum { RENDER_BLOCK = 64 };  // threads per workgroup of render_kernel
__device__ __forceinline__ unsigned image_row(const RowMap &rows, unsigned r) {
    return rows.row_begin + (r / rows.band_rows) * rows.band_pitch + (r % rows.band_rows);
}
__device__ __forceinline__ size_t tile_cost_offset(unsigned n_tiles) { return TILE_HEADER + (size_t)n_tiles; }
__device__ __forceinline__ size_t tile_class_offset(unsigned n_tiles) { return TILE_HEADER + 2 * (size_t)n_tiles; }

// For every distinct key among the wave's `valid` lanes: one atomicAdd of the number of lanes holding it to
// counters[key]; each lane gets the counter's previous value plus its rank among the lanes with the same key.
__device__ __forceinline__ unsigned wave_aggregated_add(unsigned *counters, unsigned key, bool valid) {
    const unsigned lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    unsigned slot = 0;
    while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned k = (unsigned)__shfl((int)key, leader, 64);
        const unsigned long long same = __ballot(valid && key == k) & todo;
        unsigned base = 0;
        if ((int)lane == leader) base = atomicAdd(counters + k, (unsigned)__popcll(same));
        base = (unsigned)__shfl((int)base, leader, 64);
        if (valid && key == k) slot = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

// How deep the ray tree below a hit tends to be, by what was hit: rays refracted into a bounded shape bounce about
// inside it, a transparent plane lets them go.
template <bool TIES>
__device__ __forceinline__ unsigned surface_weight(const SceneK &S, bool valid, const Closest<TIES> &hit) {
    if (!valid || hit.leaf < 0) return 1u;
    const RtSurface *s = S.surfaces + hit.leaf;
    const bool traps = s->kind != RT_PLANE && s->transparency > 0.0;
    if (traps) return s->reflectivity > 0.0 ? 24u : 12u;
    return (s->transparency > 0.0 || s->reflectivity > 0.0) ? 6u : 2u;
}

#ifndef RT_CLASSIFY_BOUNCES
#define RT_CLASSIFY_BOUNCES 1
#endif
// One lane per tile: the cost estimate of the tile from its centre ray.
template <int NEST>
__global__ __launch_bounds__(256) void classify_tiles_kernel(DeviceScene scene, RtCameraDev cam, RowMap rows,
                                                             unsigned *__restrict__ scratch) {
    static_assert(TILE_HEADER == 256, "the first workgroup clears the census and the cursors, one word per thread");
    if (blockIdx.x == 0) scratch[threadIdx.x] = 0u;  // read by the kernels that follow this one
    // A wave takes an 8x8 block of tiles, not 64 tiles of one row: its 64 centre rays then cross one region of the
    // scene instead of a strip as wide as the frame, and walk far fewer records between them.
    const unsigned tiles_x = (cam.hsize + 7) >> 3;
    const unsigned tiles_y = (rows.n_tiles + tiles_x - 1) / tiles_x;
    const unsigned blocks_x = (tiles_x + 7) >> 3;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const unsigned tx = (wave % blocks_x) * 8 + (lane & 7u), ty = (wave / blocks_x) * 8 + (lane >> 3);
    const bool valid = tx < tiles_x && ty < tiles_y;
    const unsigned tile = ty * tiles_x + tx;
    unsigned x = tx * 8 + 4, r = ty * 8 + 4;
    if (x >= cam.hsize) x = cam.hsize - 1;
    if (r >= rows.n_rows) r = rows.n_rows - 1;
    SceneK S = scene_k<NEST, false>(scene, 0);
    Ctr<false> ctr;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    if (valid) ray_for_pixel(cam, x, image_row(rows, r), ro, rd);
    // What the centre ray meets, and what the reflection and the refraction ray of that hit meet: one level of the
    // ray tree, no shading.  A floor tile that mirrors a glass ball is as deep as the ball.
    const Closest<false> best = trace<NEST, false, false>(S, ro, rd, valid, (double)INFINITY, ctr);
    unsigned weight = surface_weight(S, valid, best);
    const bool hit = valid && best.leaf >= 0;
    const RtSurface *s = S.surfaces + (hit ? best.leaf : 0);
    V3 over = ro, under = ro, reflected = rd, refracted = rd;
    bool reflects = false, refracts = false;
    if (hit) {
        const V3 point = ro + rd * best.t;
        V3 object_point;
        V3 normal = normal_at(s, point, object_point);
        const V3 eye = -rd;
        if (dot(normal, eye) < 0.0) normal = -normal;
        over = point + normal * EPS;
        under = point - normal * EPS;
        reflects = s->reflectivity != 0.0;
        reflected = reflect(rd, normal);
        if (s->transparency != 0.0) {  // world.rs:183-194 with n1 = 1, n2 = the object's index
            const double n_ratio = 1.0 / s->refractive_index;
            const double cos_i = dot(eye, normal);
            const double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
            refracts = !(sin2_t > 1.0);
            if (refracts) refracted = normal * (n_ratio * cos_i - sqrt(1.0 - sin2_t)) - eye * n_ratio;
        }
    }
    // The reflection is followed for RT_CLASSIFY_BOUNCES more hits: a floor tile whose mirror image is a cluster of
    // mirror balls has rays that go on bouncing among them, and is as slow as the cluster's own tiles.
    V3 from = over, along = reflected;
    bool bounces = reflects;
    for (int bounce = 0; bounce <= RT_CLASSIFY_BOUNCES && __ballot(bounces) != 0ull; ++bounce) {
        const Closest<false> child = trace<NEST, false, false>(S, from, along, bounces, (double)INFINITY, ctr);
        if (bounces) weight += surface_weight(S, true, child);
        bounces = bounces && child.leaf >= 0;
        if (bounces) {
            const RtSurface *c = S.surfaces + child.leaf;
            const V3 point = from + along * child.t;
            V3 object_point;
            V3 normal = normal_at(c, point, object_point);
            if (dot(normal, -along) < 0.0) normal = -normal;
            from = point + normal * EPS;
            along = reflect(along, normal);
            bounces = c->reflectivity != 0.0;
        }
    }
    if (__ballot(refracts) != 0ull) {
        const Closest<false> child = trace<NEST, false, false>(S, under, refracted, refracts, (double)INFINITY, ctr);
        if (refracts) weight += surface_weight(S, true, child);
    }
    if (valid) scratch[tile_cost_offset(rows.n_tiles) + tile] = weight * (8u + ctr.visits);  // >= 8
}

// One lane per tile: the tile's class, from the largest estimate among the tile and its eight neighbours -- a tile
// on the silhouette of a glass object is as slow as the slowest of its pixels (the grazing ones are the slowest of
// all), whatever its centre ray met -- and the census of the classes.  Counts are gathered per workgroup in LDS
// first: a frame has a handful of populated classes, and thousands of waves adding to the same few words of global
// memory would queue up behind each other.
enum { ORDER_BLOCK = 1024 };
__global__ __launch_bounds__(ORDER_BLOCK) void order_count_kernel(RowMap rows, unsigned tiles_x, unsigned *__restrict__ scratch) {
    __shared__ unsigned census[TILE_CLASSES];
    if (threadIdx.x < TILE_CLASSES) census[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned tile = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = tile < rows.n_tiles;
    const unsigned *cost_of = scratch + tile_cost_offset(rows.n_tiles);
    unsigned cost = 8u;
    if (valid) {
        const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
        const int tiles_y = (int)((rows.n_tiles + tiles_x - 1) / tiles_x);
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = tx + dx, ny = ty + dy;
                if (nx < 0 || ny < 0 || nx >= (int)tiles_x || ny >= tiles_y) continue;
                const unsigned c = cost_of[(unsigned)ny * tiles_x + (unsigned)nx];
                cost = c > cost ? c : cost;
            }
    }
    const int lg = 31 - __clz((int)(cost | 8u));                        // 3 .. 31
    const unsigned cls = 4u * (unsigned)(lg - 3) + ((cost >> (lg - 2)) & 3u);  // quarter octaves: 0 .. 115
    if (valid) scratch[tile_class_offset(rows.n_tiles) + tile] = cls;
    (void)wave_aggregated_add(census, cls, valid);
    __syncthreads();
    if (threadIdx.x < TILE_CLASSES && census[threadIdx.x] != 0u) atomicAdd(scratch + threadIdx.x, census[threadIdx.x]);
}

// One lane per tile: its place in the launch order -- after every tile of a more expensive class.  A workgroup
// reserves one run of places per class for all its tiles.
__global__ __launch_bounds__(ORDER_BLOCK) void order_scatter_kernel(RowMap rows, unsigned *__restrict__ scratch) {
    __shared__ unsigned before[TILE_CLASSES];  // tiles in classes more expensive than c, then + this workgroup's run
    __shared__ unsigned mine[TILE_CLASSES];    // this workgroup's tiles of class c
    if (threadIdx.x < TILE_CLASSES) {
        unsigned sum = 0;
        for (unsigned c = threadIdx.x + 1; c < TILE_CLASSES; ++c) sum += scratch[c];
        before[threadIdx.x] = sum;
        mine[threadIdx.x] = 0u;
    }
    __syncthreads();
    const unsigned tile = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = tile < rows.n_tiles;
    const unsigned cls = valid ? scratch[tile_class_offset(rows.n_tiles) + tile] : 0u;
    const unsigned slot = wave_aggregated_add(mine, cls, valid);
    __syncthreads();
    if (threadIdx.x < TILE_CLASSES && mine[threadIdx.x] != 0u)
        before[threadIdx.x] += atomicAdd(scratch + TILE_CLASSES + threadIdx.x, mine[threadIdx.x]);
    __syncthreads();
    if (valid) scratch[TILE_HEADER + before[cls] + slot] = tile;
}

// Canvas::write_pixel (rendering/canvas.rs:26-32) for one 8x8 tile held by one wave: lane (lane & 7, lane >> 3) has
// the colour of pixel x of packed row r (frame row y).  `staging` is LDS of the wave's own, 1 536 bytes at least.
__device__ __forceinline__ void store_tile(const RtCameraDev &cam, const RowMap &rows, unsigned tile, unsigned lane, bool valid, unsigned x,
                                           unsigned r, unsigned y, V3 acc, double *staging, double *__restrict__ rgb) {
    const unsigned tiles_x = (cam.hsize + 7) >> 3;
    // Canvas::write_pixel (rendering/canvas.rs:26-32) for the tile.  A lane's pixel is 3 values in the middle of a
    // row, so lanes storing their own pixels write 24-byte (or 3-byte) pieces 24 (3) bytes apart.  A whole tile is
    // instead laid out in LDS as its 8 rows of 24 values (the pending-ray stack is empty by now and lends the space)
    // and stored row by row with consecutive lanes on consecutive words: 8-byte stores covering 192 contiguous bytes
    // per tile row for the f64 canvas, one dword per lane for the 8-bit frame.
    // Where the rows land: packed one after the other from the start of the buffer, or (rows.in_place) at their own
    // place in a whole frame.  A tile's eight rows are consecutive frame rows when bands are whole tiles high.
    const unsigned r_out = rows.in_place ? y : r;
    const unsigned tile_row_out = rows.in_place ? image_row(rows, (tile / tiles_x) * 8) : (tile / tiles_x) * 8;
    const size_t idx = ((size_t)r_out * cam.hsize + x) * 3;  // Canvas index (canvas.rs:40-42)
    const bool whole_tile = __ballot(valid) == ~0ull && (!rows.in_place || (rows.band_rows & 7u) == 0u);
    if (rows.quantized) {
        const unsigned q0 = quantize_value(acc.x), q1 = quantize_value(acc.y), q2 = quantize_value(acc.z);
        const size_t row_bytes = (size_t)cam.hsize * 3;
        uint8_t *tile0 = rows.quantized + (size_t)tile_row_out * row_bytes + (size_t)((tile % tiles_x) * 8) * 3;
        if (whole_tile && ((row_bytes | reinterpret_cast<size_t>(rows.quantized)) & 3) == 0) {
            uint8_t *bytes = reinterpret_cast<uint8_t *>(staging);  // 8 rows x 24 bytes
            const unsigned at = (lane >> 3) * 24 + (lane & 7) * 3;
            bytes[at] = (uint8_t)q0;
            bytes[at + 1] = (uint8_t)q1;
            bytes[at + 2] = (uint8_t)q2;
            __builtin_amdgcn_wave_barrier();
            if (lane < 48) {  // 6 dwords per tile row
                const unsigned row = lane / 6, word = lane % 6;
                const unsigned v = reinterpret_cast<const unsigned *>(bytes)[row * 6 + word];
                *reinterpret_cast<unsigned *>(tile0 + row * row_bytes + word * 4) = v;
            }
        } else if (valid) {
            rows.quantized[idx] = (uint8_t)q0;
            rows.quantized[idx + 1] = (uint8_t)q1;
            rows.quantized[idx + 2] = (uint8_t)q2;
        }
    } else if (whole_tile) {
        double *vals = staging;  // 8 rows x 24 doubles
        const unsigned at = (lane >> 3) * 24 + (lane & 7) * 3;
        vals[at] = acc.x;
        vals[at + 1] = acc.y;
        vals[at + 2] = acc.z;
        __builtin_amdgcn_wave_barrier();
        double *tile0 = rgb + ((size_t)tile_row_out * cam.hsize + (size_t)((tile % tiles_x) * 8)) * 3;
#pragma unroll
        for (unsigned k = 0; k < 3; ++k) {
            const unsigned j = lane + 64 * k;  // 192 values
            tile0[(size_t)(j / 24) * cam.hsize * 3 + j % 24] = vals[j];
        }
    } else if (valid) {
        rgb[idx] = acc.x;
        rgb[idx + 1] = acc.y;
        rgb[idx + 2] = acc.z;
    }
}

// Camera::render (rendering/camera.rs:56-67) for the rows of `rows`: one wave per 8x8 pixel tile.
// Launched with one wave per workgroup (RENDER_BLOCK).  Flat scenes: four waves per SIMD, a budget of 128 VGPRs per
// lane, which that kernel fits without spilling.  The general build for scenes with groups (NEST 1): five waves per
// SIMD (96 VGPRs, over a hundred of the traversal's values spilled): its node tables run to megabytes, the record
// fetches miss the scalar cache, and a fifth wave to switch to is worth more than the spills cost (the generated scene
// at 8K: 70.4 ms against 74.3 at four waves; six waves 72.1).  The build without cone directions (NEST 2: chapter_14,
// whose table is 0.1 MB): four waves, no spilled vector registers -- 4.50 ms against 4.60 at five.
template <bool STATS, int NEST, bool TIES>
#ifndef RT_NEST_WAVES
#define RT_NEST_WAVES 5
#endif
#ifndef RT_NEST2_WAVES
#define RT_NEST2_WAVES 4
#endif
__global__ __launch_bounds__(RENDER_BLOCK, NEST == 2 ? RT_NEST2_WAVES : NEST > 0 ? RT_NEST_WAVES : 4) void render_kernel(DeviceScene scene, RtCameraDev cam, RowMap rows, int max_depth,
                                                     double *__restrict__ rgb, unsigned long long *counters) {
    const unsigned lane = threadIdx.x & 63;
    unsigned tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rows.tile_list) {  // the tile this wave renders: the launch order lists the expensive ones first
        if (tile >= rows.n_tiles) return;
        tile = ((const unsigned RT_CONSTANT_AS *)rows.tile_list)[TILE_HEADER + tile];
    }
    const unsigned tiles_x = (cam.hsize + 7) >> 3;
    const unsigned x = (tile % tiles_x) * 8 + (lane & 7);
    const unsigned r = (tile / tiles_x) * 8 + (lane >> 3);  // row in the packed output
    const unsigned y = image_row(rows, r);
    const bool valid = x < cam.hsize && r < rows.n_rows && y < rows.row_end;
    SceneK S = scene_k<NEST, true>(scene, max_depth);
    Ctr<STATS> ctr;
    ctr.clear();
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    if (valid) {
        ray_for_pixel(cam, x, y, ro, rd);
        ctr.add(C_PRIMARY);
    }
    V3 acc;
    radiance<NEST, STATS, TIES>(S, valid, ro, rd, max_depth, acc, ctr);
#ifdef RT_TIMING
    if (lane == 0 && blockIdx.x < RT_TIMING_WAVES) g_rt_timing[blockIdx.x][6] = tile;
#endif
    store_tile(cam, rows, tile, lane, valid, x, r, y, acc, ray_stack_lds + S.stack_base, rgb);
    flush_counters<STATS>(ctr, counters);
}

template <bool STATS, int NEST>
__global__ __launch_bounds__(256) void color_at_kernel(DeviceScene scene, const double *__restrict__ rays, unsigned n,
                                                       int remaining, double *__restrict__ colors,
                                                       unsigned long long *counters) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    SceneK S = scene_k<NEST, true>(scene, remaining);
    Ctr<STATS> ctr;
    ctr.clear();
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    if (valid) {
        ro = mk(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
        rd = mk(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    }
    V3 acc;
    radiance<NEST, STATS>(S, valid, ro, rd, remaining, acc, ctr);
    if (valid) {
        colors[i * 3 + 0] = acc.x;
        colors[i * 3 + 1] = acc.y;
        colors[i * 3 + 2] = acc.z;
    }
    flush_counters<STATS>(ctr, counters);
}

__global__ __launch_bounds__(256) void intersect_kernel(DeviceScene scene, const double *__restrict__ rays, unsigned n,
                                                        double *__restrict__ t, int *__restrict__ object) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    SceneK S = scene_k<1, false>(scene, 0);
    Ctr<false> ctr;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    if (valid) {
        ro = mk(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
        rd = mk(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    }
    const Closest<true> best = trace<1, false>(S, ro, rd, valid, (double)INFINITY, ctr);
    if (valid) {
        t[i] = best.leaf >= 0 ? best.t : __builtin_nan("");
        object[i] = best.leaf >= 0 ? S.surfaces[best.leaf].object : -1;
    }
}

// World::intersect (rendering/world.rs:27-36) for one ray per lane: every intersection, stably sorted by t.  Lane i
// owns t[i*cap .. (i+1)*cap) and object[...]; count[i] is the number the reference's list would hold (the lists are
// only complete when it is <= cap).  The sort is an insertion sort by (t, object): among equal t's the reference's
// stable sort (world.rs:34) keeps its emission order, which is object order (the node table may list siblings in
// another order), and within one object the order of emission, which the traversal keeps.
__global__ __launch_bounds__(256) void intersect_all_kernel(DeviceScene scene, const double *__restrict__ rays, unsigned n,
                                                            unsigned cap, unsigned *__restrict__ count,
                                                            double *__restrict__ t, int *__restrict__ object) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    SceneK S = scene_k<1, false>(scene, 0);
    Ctr<false> ctr;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    AllHits all;
    all.t = t + (size_t)(valid ? i : 0) * cap;
    all.leaf = object + (size_t)(valid ? i : 0) * cap;
    all.cap = valid ? cap : 0u;
    all.count = 0u;
    if (valid) {
        ro = mk(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
        rd = mk(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    }
    trace_nested<false>(S, ro, rd, valid, (double)INFINITY, all, ctr);
    if (!valid) return;
    count[i] = all.count;
    const unsigned m = all.count < cap ? all.count : cap;
    for (unsigned k = 0; k < m; ++k) all.leaf[k] = S.surfaces[all.leaf[k]].object;
    for (unsigned k = 1; k < m; ++k) {
        const double tk = all.t[k];
        const int ok = all.leaf[k];
        unsigned j = k;
        while (j > 0 && (tk < all.t[j - 1] || (tk == all.t[j - 1] && ok < all.leaf[j - 1]))) {
            all.t[j] = all.t[j - 1];
            all.leaf[j] = all.leaf[j - 1];
            --j;
        }
        all.t[j] = tk;
        all.leaf[j] = ok;
    }
}

__device__ __forceinline__ V3 load3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ void store3(double *p, V3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

// The geometric half of Intersection::prepare_computations_for_intersections (rendering/intersections.rs:51-63) for
// hit i = (comps[i].time, surface comps[i].object) on rays[i]; time, object, n1 and n2 are the caller's.
__global__ __launch_bounds__(256) void prepare_computations_kernel(DeviceScene scene, const double *__restrict__ rays,
                                                                   unsigned n, RtComps *__restrict__ comps) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RtComps &c = comps[i];
    const V3 ro = load3(rays + i * 6), rd = load3(rays + i * 6 + 3);
    const V3 point = ro + rd * c.time;  // Ray::position (rendering/rays.rs:18-20)
    V3 object_point;
    V3 normal = normal_at(scene.surfaces + c.object, point, object_point);
    const V3 eye = -rd;
    const bool inside = dot(normal, eye) < 0.0;
    if (inside) normal = -normal;
    c.inside = inside ? 1 : 0;
    store3(c.point, point);
    store3(c.eye_vector, eye);
    store3(c.normal_vector, normal);
    store3(c.reflect_vector, reflect(rd, normal));
    store3(c.over_point, point + normal * EPS);
    store3(c.under_point, point - normal * EPS);
}

// Computations::schlick (rendering/intersections.rs:123-136)
__device__ __forceinline__ double schlick(V3 eye, V3 normal, double n1, double n2) {
    double c = dot(eye, normal);
    if (n1 > n2) {
        const double n = n1 / n2;
        const double sin2_t = (n * n) * (1.0 - c * c);
        if (sin2_t > 1.0) return 1.0;
        c = sqrt(1.0 - sin2_t);
    }
    const double q = (n1 - n2) / (n1 + n2);
    const double r0 = q * q;
    const double m = 1.0 - c;
    const double m2 = m * m;
    return r0 + (1.0 - r0) * (m * (m2 * m2));
}

// World::shade_hit_internal, reflected_color_internal and refracted_color_internal (rendering/world.rs:84-111,
// 155-168, 174-198) on caller-made Computations, which is how the reference's own tests reach them: the first level
// uses the record's n1 / n2 and the nested sum of world.rs:107-110 as written; the rays it spawns go through
// color_at like any other ray.
template <int PART, int NEST>
__global__ __launch_bounds__(256) void shade_part_kernel(DeviceScene scene, const RtComps *__restrict__ comps, unsigned n,
                                                         int remaining, double *__restrict__ colors) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    SceneK S = scene_k<NEST, true>(scene, remaining);
    Ctr<false> ctr;
    const RtComps &c = comps[valid ? i : 0];
    const RtSurface *s = S.surfaces + c.object;
    const V3 eye = load3(c.eye_vector), normal = load3(c.normal_vector);
    const bool lit_world = S.light.has_light != 0;

    V3 surface = mk(0.0, 0.0, 0.0);
    if (PART == RT_PART_SHADE_HIT) {
        const V3 over = load3(c.over_point), point = load3(c.point);
        const V3 v = mk(S.light.position[0], S.light.position[1], S.light.position[2]) - over;  // world.rs:136-138
        const double distance = magnitude(v);
        const bool in_shadow = occluded<NEST, false>(S, over, div_by_magnitude(v, distance), valid && lit_world, distance, ctr);
        if (valid && lit_world) {
            V3 shadowed, lit;
            int flags;
            lighting<false>(s, S.light, point, mul_point(s->inv_world, point), eye, normal, shadowed, lit, flags, ctr);
            surface = in_shadow ? shadowed : lit;
        }
    }

    V3 reflected = mk(0.0, 0.0, 0.0);
    if (PART != RT_PART_REFRACTED) {
        const bool go = valid && remaining != 0 && s->reflectivity != 0.0 && (PART != RT_PART_SHADE_HIT || lit_world);
        V3 child;
        radiance<NEST, false>(S, go, load3(c.over_point), load3(c.reflect_vector), remaining - 1, child, ctr);
        if (go) reflected = child * s->reflectivity;
    }

    V3 refracted = mk(0.0, 0.0, 0.0);
    if (PART != RT_PART_REFLECTED) {
        bool go = valid && remaining != 0 && s->transparency != 0.0 && (PART != RT_PART_SHADE_HIT || lit_world);
        const double n_ratio = c.n1 / c.n2;
        const double cos_i = dot(eye, normal);
        const double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
        if (sin2_t > 1.0) go = false;
        const double cos_t = sqrt(1.0 - sin2_t);
        const V3 direction = normal * (n_ratio * cos_i - cos_t) - eye * n_ratio;
        V3 child;
        radiance<NEST, false>(S, go, load3(c.under_point), direction, remaining - 1, child, ctr);
        if (go) refracted = child * s->transparency;
    }

    if (!valid) return;
    V3 out;
    if (PART == RT_PART_REFLECTED) {
        out = reflected;
    } else if (PART == RT_PART_REFRACTED) {
        out = refracted;
    } else if (!lit_world) {
        out = mk(0.0, 0.0, 0.0);  // world.rs:86-89
    } else if (s->reflectivity > 0.0 && s->transparency > 0.0) {
        const double reflectance = schlick(eye, normal, c.n1, c.n2);
        out = (surface + reflected * reflectance) + refracted * (1.0 - reflectance);
    } else {
        out = (surface + reflected) + refracted;
    }
    store3(colors + (size_t)i * 3, out);
}

__global__ __launch_bounds__(256) void is_shadowed_kernel(DeviceScene scene, const double *__restrict__ points,
                                                          unsigned n, uint8_t *__restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    SceneK S = scene_k<1, false>(scene, 0);
    Ctr<false> ctr;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    double distance = 0.0;
    const bool lit_world = S.light.has_light != 0;
    if (valid) {
        ro = mk(points[i * 3 + 0], points[i * 3 + 1], points[i * 3 + 2]);
        V3 v = mk(S.light.position[0], S.light.position[1], S.light.position[2]) - ro;  // world.rs:136-138
        distance = magnitude(v);
        rd = div_by_magnitude(v, distance);
    }
    const bool hit = occluded<1, false>(S, ro, rd, valid && lit_world, distance, ctr);
    if (valid) out[i] = (lit_world && hit) ? 1 : 0;
}

__global__ __launch_bounds__(256) void normal_at_kernel(DeviceScene scene, const int *__restrict__ leaf,
                                                        const double *__restrict__ points, unsigned n,
                                                        double *__restrict__ normals) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = leaf[i];
    V3 r = mk(__builtin_nan(""), __builtin_nan(""), __builtin_nan(""));
    V3 object_point;
    if (l >= 0 && l < scene.n_leaves)
        r = normal_at(scene.surfaces + l, mk(points[i * 3 + 0], points[i * 3 + 1], points[i * 3 + 2]), object_point);
    normals[i * 3 + 0] = r.x;
    normals[i * 3 + 1] = r.y;
    normals[i * 3 + 2] = r.z;
}

__global__ __launch_bounds__(256) void ray_for_pixel_kernel(RtCameraDev cam, const unsigned *__restrict__ px,
                                                            const unsigned *__restrict__ py, unsigned n,
                                                            double *__restrict__ rays) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d;
    ray_for_pixel(cam, px[i], py[i], o, d);
    rays[i * 6 + 0] = o.x; rays[i * 6 + 1] = o.y; rays[i * 6 + 2] = o.z;
    rays[i * 6 + 3] = d.x; rays[i * 6 + 4] = d.y; rays[i * 6 + 5] = d.z;
}

// convert_canvas_color_value_to_decimal_rgb_value (rendering/canvas.rs:123-130) over a frame that is still in device
// memory: (value * 255.0).round() as u32, then min(255).  `round` is half-away-from-zero; the Rust cast saturates and
// maps NaN to 0.  Eight values per lane: 64 B loaded, 8 B stored, both contiguous across the wave.
__global__ __launch_bounds__(256) void quantize_kernel(const double *__restrict__ rgb, size_t n, uint8_t *__restrict__ out) {
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (base >= n) return;
    unsigned q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        q[k] = quantize_value(base + k < n ? rgb[base + k] : 0.0);
    }
    if (base + 8 <= n && (reinterpret_cast<size_t>(out + base) & 7) == 0) {
        uint2 w;
        w.x = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
        w.y = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
        *reinterpret_cast<uint2 *>(out + base) = w;
    } else {
        for (int k = 0; k < 8 && base + k < n; ++k) out[base + k] = (uint8_t)q[k];
    }
}

// Packed stripes -> their places in a frame (the collection step of a frame rendered on several GPUs): word i of the
// packed source belongs to stripe i / stripe_words.  T is the widest word that divides every size and address involved.
template <typename T>
__global__ __launch_bounds__(256) void place_stripes_kernel(const T *__restrict__ src, T *__restrict__ dst, size_t stripe_words,
                                                           size_t dst_pitch_words, size_t total_words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_words) return;
    const size_t stripe = i / stripe_words;
    dst[stripe * dst_pitch_words + (i - stripe * stripe_words)] = src[i];
}

inline unsigned blocks_for(unsigned n) { return (n + 255u) / 256u; }

// Camera::render for a world without groups in which nothing reflects or transmits (World::reflected_color and
// refracted_color return black without tracing, world.rs:156-162, 175-181), or with a recursion budget of 0: a pixel
// is its primary ray, one shading and one shadow ray.  No loop, no stack of pending rays: half the registers of
// render_kernel, so twice the waves per SIMD -- and such frames are made of short waves whose time is latency (a
// record fetch after the other), which is what more waves hide.  The same functions in the same order as one round
// of radiance(): the same bits.
#ifndef RT_PRIMARY_WAVES
#define RT_PRIMARY_WAVES 6
#endif
#ifndef RT_PRIMARY_KERNEL
#define RT_PRIMARY_KERNEL 1
#endif
__global__ __launch_bounds__(RENDER_BLOCK, RT_PRIMARY_WAVES) void primary_kernel(DeviceScene scene, RtCameraDev cam, RowMap rows, double *__restrict__ rgb) {
    __shared__ double staging[192];
    const unsigned lane = threadIdx.x & 63;
    unsigned tile = blockIdx.x;
    if (rows.tile_list) {
        if (tile >= rows.n_tiles) return;
        tile = ((const unsigned RT_CONSTANT_AS *)rows.tile_list)[TILE_HEADER + tile];
    }
    const unsigned tiles_x = (cam.hsize + 7) >> 3;
    const unsigned x = (tile % tiles_x) * 8 + (lane & 7);
    const unsigned r = (tile / tiles_x) * 8 + (lane >> 3);  // row in the packed output
    const unsigned y = image_row(rows, r);
    const bool valid = x < cam.hsize && r < rows.n_rows && y < rows.row_end;
    SceneK S = scene_k<0, false>(scene, 0);
    Ctr<false> ctr;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    if (valid) ray_for_pixel(cam, x, y, ro, rd);
    V3 acc = mk(0.0, 0.0, 0.0);
    const Closest<false> best = trace<0, false, false>(S, ro, rd, valid, (double)INFINITY, ctr);
    const bool shade = valid && best.leaf >= 0 && S.light.has_light;  // else black: world.rs:124, world.rs:86-89
    V3 acc_lit = acc, sro = ro, srd = rd;
    double h_dist = 0.0;
    bool own_shadow = false;
    if (shade) {
        HitShade h;
        const V3 light_pos = mk(S.light.position[0], S.light.position[1], S.light.position[2]);
        shade_hit<true, false, RT_OWN_SHADOW>(S.light, light_pos, S.surfaces + best.leaf, ro, rd, best.t, 1.0, 0, h, ctr);
        acc_lit = acc + h.lit * 1.0;
        acc = acc + h.shadowed * 1.0;
        h_dist = h.light_distance;
        sro = h.over;
        srd = h.shadow_dir;
        own_shadow = h.own_shadow;
    }
    if (__ballot(shade) != 0ull) {
        const bool walk = shade && !own_shadow;
        bool in_shadow = own_shadow;
        if (__ballot(walk) != 0ull) {
            const int start = __builtin_amdgcn_readlane(best.leaf, __builtin_ctzll(__ballot(walk)));
            in_shadow = occluded<0, false, true>(S, sro, srd, walk, h_dist, ctr, start) || own_shadow;  // world.rs:145-148
        }
        if (shade && !in_shadow) acc = acc_lit;
    }
    store_tile(cam, rows, tile, lane, valid, x, r, y, acc, staging, rgb);
}

}  // namespace

// The kernels exist in two builds: for scenes without groups (NEST = 0: every record a leaf, no group or range-skip
// code, the lean 128-byte record fetch) and for everything else (NEST = 1: one loop over the table whatever the depth
// of nesting).
template <bool STATS>
static void launch_render_nest(int nest, dim3 grid, unsigned block, hipStream_t stream, const DeviceScene &scene, const RtCameraDev &camera,
                               const RowMap &rows, int max_depth, double *rgb, unsigned long long *counters) {
#define RT_LAUNCH(N, T) render_kernel<STATS, N, T><<<grid, block, lds_bytes<N, true>(block), stream>>>(scene, camera, rows, max_depth, rgb, counters)
    if (nest == 0) {
        RT_LAUNCH(0, false);
        return;
    }
    // The counting variant always breaks ties by leaf number (right for any table order, and its speed does not
    // matter); the plain one only when the host reordered a sibling list, because the comparison is paid per candidate.
    if (!STATS && scene.any_order != 0 && scene.cone_nodes == 0) RT_LAUNCH(2, true);
    else if (STATS || scene.any_order != 0) RT_LAUNCH(1, true);
    else RT_LAUNCH(1, false);
#undef RT_LAUNCH
}

#include "rt_levels.inl"

size_t render_scratch_words(uint32_t hsize, uint32_t n_rows) {
    return TILE_HEADER + 3 * (size_t)((hsize + 7) / 8) * ((n_rows + 7) / 8);
}

hipError_t launch_render(const DeviceScene &scene, const RtCameraDev &camera, const RowMap &rows_in, int max_depth,
                         double *rgb, unsigned long long *counters, uint32_t *tile_scratch, hipStream_t stream,
                         hipEvent_t *marks, bool order_cached) {
    if (rows_in.n_rows == 0 || camera.hsize == 0 || rows_in.band_rows == 0) return hipSuccess;
    const unsigned long long tiles = (unsigned long long)((camera.hsize + 7) / 8) * ((rows_in.n_rows + 7) / 8);
    if (tiles > 0x7fffffffull) return hipErrorInvalidValue;
    RowMap rows = rows_in;
    rows.n_tiles = (unsigned)tiles;
    if (marks) {
        hipError_t e = hipEventRecord(marks[0], stream);
        if (e != hipSuccess) return e;
    }
    if (tile_scratch && order_cached) {
        rows.tile_list = tile_scratch;  // the order an earlier launch worked out for these very rows and this very camera
    } else if (tile_scratch) {
        const unsigned tiles_x = (camera.hsize + 7) / 8, tiles_y = (rows.n_tiles + tiles_x - 1) / tiles_x;
        const dim3 cgrid(blocks_for(((tiles_x + 7) / 8) * ((tiles_y + 7) / 8) * 64u));  // one wave per 8x8 block of tiles
        if (scene.max_nesting == 0) classify_tiles_kernel<0><<<cgrid, 256, lds_bytes<0, false>(256), stream>>>(scene, camera, rows, tile_scratch);
        else classify_tiles_kernel<1><<<cgrid, 256, lds_bytes<1, false>(256), stream>>>(scene, camera, rows, tile_scratch);
        const dim3 ogrid((rows.n_tiles + ORDER_BLOCK - 1) / ORDER_BLOCK);
        order_count_kernel<<<ogrid, ORDER_BLOCK, 0, stream>>>(rows, (camera.hsize + 7) / 8, tile_scratch);
        order_scatter_kernel<<<ogrid, ORDER_BLOCK, 0, stream>>>(rows, tile_scratch);
        rows.tile_list = tile_scratch;
    }
    if (marks) {
        hipError_t e = hipEventRecord(marks[1], stream);
        if (e != hipSuccess) return e;
    }
    // One wave per workgroup: a finished wave's slot is refilled at once, without waiting for three more.
    const dim3 grid(rows.n_tiles);
    const unsigned block = RENDER_BLOCK;
    if (counters)
        launch_render_nest<true>(scene.max_nesting, grid, block, stream, scene, camera, rows, max_depth, rgb, counters);
    else if (RT_PRIMARY_KERNEL && scene.max_nesting == 0 && (scene.no_secondary != 0 || max_depth == 0))
        primary_kernel<<<grid, block, 0, stream>>>(scene, camera, rows, rgb);
    else
        launch_render_nest<false>(scene.max_nesting, grid, block, stream, scene, camera, rows, max_depth, rgb, nullptr);
    hipError_t launched = hipGetLastError();
    if (launched == hipSuccess && marks) launched = hipEventRecord(marks[2], stream);
    return launched;
}

// Tile-ordering scratch for callers that must not allocate (rt_render_device_noalloc): one array per GPU, held by
// every GPU the library's code object is loaded on until the process ends.  In host code the name `scratch` is only a
// handle for hipGetSymbolAddress -- it is NOT a device address; host code uses the pointer that call returned
// (one per device: every GPU has its own copy of the array at its own address).
static __device__ __align__(256) uint32_t scratch[TILE_HEADER + 3u * RT_NOALLOC_MAX_TILES];
static_assert(sizeof(scratch) == RT_NOALLOC_SCRATCH_BYTES, "header and array disagree about the size of the static scratch");

namespace {
constexpr int MAX_DEVICES = 64;
uint32_t *g_static_scratch[MAX_DEVICES] = {};  // per device; written once, under g_static_scratch_mutex
std::mutex g_static_scratch_mutex;
}  // namespace

hipError_t static_render_scratch_resolve(int device) {
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (g_static_scratch[device]) return hipSuccess;
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) return hipErrorInvalidDevice;  // the address is that of the CURRENT device's copy
    uint32_t *ptr = nullptr;
    e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_static_scratch[device] = ptr;
    return hipSuccess;
}

uint32_t *static_render_scratch(int device) {
    if (device < 0 || device >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    return g_static_scratch[device];
}

hipError_t launch_color_at(const DeviceScene &scene, const double *rays, uint32_t n, int remaining, double *colors,
                           unsigned long long *counters, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const bool flat = scene.max_nesting == 0;
    if (counters) {
        if (flat) color_at_kernel<true, 0><<<blocks_for(n), 256, lds_bytes<0, true>(256), stream>>>(scene, rays, n, remaining, colors, counters);
        else color_at_kernel<true, 1><<<blocks_for(n), 256, lds_bytes<1, true>(256), stream>>>(scene, rays, n, remaining, colors, counters);
    } else {
        if (flat) color_at_kernel<false, 0><<<blocks_for(n), 256, lds_bytes<0, true>(256), stream>>>(scene, rays, n, remaining, colors, nullptr);
        else color_at_kernel<false, 1><<<blocks_for(n), 256, lds_bytes<1, true>(256), stream>>>(scene, rays, n, remaining, colors, nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_intersect(const DeviceScene &scene, const double *rays, uint32_t n, double *t, int32_t *object,
                            hipStream_t stream) {
    if (n == 0) return hipSuccess;
    intersect_kernel<<<blocks_for(n), 256, lds_bytes<1, false>(256), stream>>>(scene, rays, n, t, object);
    return hipGetLastError();
}

hipError_t launch_intersect_all(const DeviceScene &scene, const double *rays, uint32_t n, uint32_t cap, uint32_t *count,
                                double *t, int32_t *object, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    intersect_all_kernel<<<blocks_for(n), 256, lds_bytes<1, false>(256), stream>>>(scene, rays, n, cap, count, t, object);
    return hipGetLastError();
}

hipError_t launch_prepare_computations(const DeviceScene &scene, const double *rays, uint32_t n, RtComps *comps,
                                       hipStream_t stream) {
    if (n == 0) return hipSuccess;
    prepare_computations_kernel<<<blocks_for(n), 256, 0, stream>>>(scene, rays, n, comps);
    return hipGetLastError();
}

template <int PART>
static void launch_part(const DeviceScene &scene, const RtComps *comps, uint32_t n, int remaining, double *colors, hipStream_t stream) {
    if (scene.max_nesting == 0) shade_part_kernel<PART, 0><<<blocks_for(n), 256, lds_bytes<0, true>(256), stream>>>(scene, comps, n, remaining, colors);
    else shade_part_kernel<PART, 1><<<blocks_for(n), 256, lds_bytes<1, true>(256), stream>>>(scene, comps, n, remaining, colors);
}

hipError_t launch_shade_part(const DeviceScene &scene, int part, const RtComps *comps, uint32_t n, int remaining,
                             double *colors, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    switch (part) {
        case RT_PART_SHADE_HIT: launch_part<RT_PART_SHADE_HIT>(scene, comps, n, remaining, colors, stream); break;
        case RT_PART_REFLECTED: launch_part<RT_PART_REFLECTED>(scene, comps, n, remaining, colors, stream); break;
        case RT_PART_REFRACTED: launch_part<RT_PART_REFRACTED>(scene, comps, n, remaining, colors, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_is_shadowed(const DeviceScene &scene, const double *points, uint32_t n, uint8_t *out,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    is_shadowed_kernel<<<blocks_for(n), 256, lds_bytes<1, false>(256), stream>>>(scene, points, n, out);
    return hipGetLastError();
}

hipError_t launch_normal_at(const DeviceScene &scene, const int32_t *leaf, const double *points, uint32_t n,
                            double *normals, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    normal_at_kernel<<<blocks_for(n), 256, 0, stream>>>(scene, leaf, points, n, normals);
    return hipGetLastError();
}

hipError_t launch_quantize(const double *rgb, size_t n, uint8_t *out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t per_block = 256 * 8;
    quantize_kernel<<<(unsigned)((n + per_block - 1) / per_block), 256, 0, stream>>>(rgb, n, out);
    return hipGetLastError();
}

hipError_t launch_place_stripes(const void *src, void *dst, size_t stripe_bytes, size_t dst_pitch_bytes, size_t total_bytes,
                                hipStream_t stream) {
    if (total_bytes == 0) return hipSuccess;
    if (stripe_bytes == 0) return hipErrorInvalidValue;
    const size_t all = stripe_bytes | dst_pitch_bytes | total_bytes | reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst);
    auto launch = [&](auto word) {
        using T = decltype(word);
        const size_t n = total_bytes / sizeof(T);
        if ((n + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
        place_stripes_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(static_cast<const T *>(src), static_cast<T *>(dst),
                                                                                stripe_bytes / sizeof(T), dst_pitch_bytes / sizeof(T), n);
        return hipGetLastError();
    };
    if ((all & 15) == 0) return launch(uint4{});
    if ((all & 3) == 0) return launch(uint32_t{});
    return launch(uint8_t{});
}

hipError_t launch_ray_for_pixel(const RtCameraDev &camera, const uint32_t *px, const uint32_t *py, uint32_t n,
                                double *rays, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    ray_for_pixel_kernel<<<blocks_for(n), 256, 0, stream>>>(camera, px, py, n, rays);
    return hipGetLastError();
}

}  // namespace rt
