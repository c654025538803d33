// Flattens a World (objects in depth-first pre-order) into the tables of rt_device_types.h.
//
// What the reference recomputes for every ray is computed here once, with the same algorithms, because it is a
// pure function of the scene:
//   - inverse(transformation) per object          (rendering/objects.rs:122-126, rendering/world.rs:53-58)
//   - inverse(world_transformation) per leaf      (rendering/objects.rs:87-102)
//   - inverse(pattern.transform) per pattern      (scene/patterns.rs:140-144)
//   - Group::bounds() per group                   (geometry/groups.rs:130-149)
// Added on top (no counterpart in the reference): range-skip nodes, conservative boxes around runs of consecutive
// siblings that let the traversal step over a whole run no ray of the wave can touch.
// A matrix the reference would fail to invert at render time (`expect("... should be invertible")`) is reported
// here as RT_ERR_NOT_INVERTIBLE instead of aborting.
#include "rt_flatten.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

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

// The ball of an RT_XFORM_DIAGONAL_BALL record (rt_device_types.h), from the inverse k_i x + t_i of a scale-and-translate
// transformation: centre -t_i / k_i, radius max 1/|k_i|.  Returns false when no ball should be used.
//
// What the ball has to contain.  The reference reports a hit of the sphere when the discriminant it computes in the
// sphere's own space from the transformed ray is not negative (geometry/sphere.rs:51-74).  Both the transformed ray
// and the discriminant carry rounding errors, so a ray that truly passes the unit sphere at distance sqrt(1 + e) may
// still be reported as hitting it.  In the sphere's space, with L the distance of the transformed origin from the
// centre: the transformed origin k_i o_i + t_i is off by at most 2^-52 |k| (|o| + |centre|), which moves the squared
// distance of the line by at most 2 L times that; and b^2 - 4ac, in units of 4a, is off by at most some thirty
// 2^-53 L^2.  Back in the list's space (distances scale by at most the largest scale factor, L <= |o - centre| over
// the smallest, |k| <= 1 over the smallest) and with the scale factors at most a hundredfold apart -- anything else
// gets no ball -- that is an excess of the squared distance of at most
//     4.2e-11 |o - centre|^2 + 4.4e-12 |centre|^2.
// The device takes 1e-9 |o - centre|^2 off the squared distance it measures and the radius here carries 1e-10
// |centre|^2: more than twenty times the bound each.  The 1e-9 on the radius itself covers the rounding of the
// centre and of the device's own arithmetic.
bool ball_of(const double inv[12], double centre[3], double &radius2) {
    const double k[3] = {inv[0], inv[5], inv[10]}, t[3] = {inv[3], inv[7], inv[11]};
    double smax = 0.0, smin = INFINITY, c2 = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double s = 1.0 / std::fabs(k[i]);
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        smax = std::max(smax, s);
        smin = std::min(smin, s);
        centre[i] = -t[i] / k[i];
        if (!std::isfinite(centre[i])) return false;
        c2 += centre[i] * centre[i];
    }
    if (smax > 100.0 * smin) return false;
    radius2 = (smax * smax * (1.0 + 1e-9) + 1e-10 * c2) / (1.0 - 1e-9);  // the device compares with |o - centre|^2 unshrunk
    return std::isfinite(radius2);
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

// Range-skip nodes (see RT_SKIPBOX in rt_device_types.h).  A world without groups whose list has at most
// SKIP_MIN_LIST members is left as the reference has it: a test per member is cheaper there than wider records and
// extra branches.  Everywhere else every sibling list gets an order tree, and the tree nodes that become range-skip
// nodes are chosen by expected cost (Builder::plan).
static size_t env_size(const char *name, size_t fallback) {
    const char *e = std::getenv(name);
    return e && *e ? (size_t)std::strtoul(e, nullptr, 10) : fallback;
}
static const size_t SKIP_MIN_LIST = env_size("RT_SKIP_MIN_LIST", 12);  // (developer override for tuning runs)
// What a wave spends on one record, in vector instructions, the traversal loop's own share included: the figures
// the choice of range-skip nodes is made with.  (Developer override: RT_COST_BOX.)
static const double COST_BOX = (double)env_size("RT_COST_BOX", 40);  // a range-skip node: slab test, reach guard, its record
static const double COST_RECORD = 20.0;                              // the loop's own work per record
static const double COST_RAY_GENERAL = 33.0, COST_RAY_DIAGONAL = 9.0;  // the ray into a node's space
static const double COST_BOUNDS = 75.0;                              // a group's bounds test with its six divisions
static const double COST_LEAF[5] = {25.0, 8.0, 60.0, 70.0, 80.0};    // intersection routine: sphere, plane, cube, cylinder, cone

// The directions for which some cone below a node takes the shortcut of cones.rs:66-78: with a = dx^2 - dy^2 + dz^2
// of the ray in the cone's space below EPSILON, the reference accepts t = -c / (2 b) -- a point that need not lie on
// the cone or anywhere near it -- whenever that point's y is within the cone's limits.  No bounding box is
// conservative for such a ray, so a range-skip node must let it through.  In the space of a sibling list,
// a = lambda * d^T q d with q the unit-norm symmetric matrix A^T diag(1,-1,1) A (A = linear part of the chain of
// inverses down to the cone) and lambda > 0.  A node can carry one q: cones whose q's agree (same axis and
// proportions, any size -- every cone of a typical scene) share it; otherwise `ok` is false and the run gets no node.
// When may two cones' q's count as one?  Replacing q_k by q_0 moves d^T q d by at most 3 tol |d|^2, and the bound a
// node tests against (2 EPSILON / lambda_min) leaves EPSILON / lambda_min of room above the values at which the
// reference takes its shortcut.  q's that are meant to be equal -- same axis and proportions -- differ by rounding
// only (they have unit norm: 1e-16 or so per entry), so the tolerance can be tiny: 1e-12, and never more than a
// thousandth of that room (small cones have little: lambda grows with 1 / size^2).
inline double cone_q_tolerance(double lambda_min) { return std::fmin(1e-12, 1e-3 * 1e-5 / lambda_min); }

struct ConeDirs {
    bool any = false, ok = true;
    double q[6] = {0, 0, 0, 0, 0, 0};  // xx, xy, xz, yy, yz, zz
    double lambda_min = 0.0;

    static ConeDirs of_cone(const Matrix4 *inv) {
        ConeDirs c;
        c.any = true;
        if (!inv) {
            c.ok = false;
            return c;
        }
        double Q[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Q[i][j] = inv->d[0][i] * inv->d[0][j] - inv->d[1][i] * inv->d[1][j] + inv->d[2][i] * inv->d[2][j];
        c.set(Q, 1.0);
        return c;
    }
    void set(const double Q[3][3], double lambda) {
        double n = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) n += Q[i][j] * Q[i][j];
        n = std::sqrt(n);
        if (!(n > 0.0) || !std::isfinite(n)) {
            ok = false;
            return;
        }
        q[0] = Q[0][0] / n; q[1] = Q[0][1] / n; q[2] = Q[0][2] / n; q[3] = Q[1][1] / n; q[4] = Q[1][2] / n; q[5] = Q[2][2] / n;
        lambda_min = lambda * n;
    }
    // the same set of directions seen from the parent of a group whose inverse transformation is `inv`
    ConeDirs through(const Matrix4 *inv) const {
        ConeDirs c = *this;
        if (!any || !ok) return c;
        if (!inv) {
            c.ok = false;
            return c;
        }
        const double S[3][3] = {{q[0], q[1], q[2]}, {q[1], q[3], q[4]}, {q[2], q[4], q[5]}};
        double Q[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double v = 0.0;
                for (int k = 0; k < 3; ++k)
                    for (int l = 0; l < 3; ++l) v += inv->d[k][i] * S[k][l] * inv->d[l][j];
                Q[i][j] = v;
            }
        c.set(Q, lambda_min);
        return c;
    }
    void merge(const ConeDirs &o) {
        if (!o.any) return;
        if (!any) {
            *this = o;
            return;
        }
        if (!o.ok) ok = false;
        if (!ok) return;
        lambda_min = std::fmin(lambda_min, o.lambda_min);
        for (int i = 0; i < 6; ++i)
            if (!(std::fabs(q[i] - o.q[i]) <= cone_q_tolerance(lambda_min))) ok = false;
    }
};

struct Builder {
    const rt_world_desc &world;
    FlatScene &out;
    std::string &err;
    int code = RT_OK;
    Builder(const rt_world_desc &w, FlatScene &o, std::string &e) : world(w), out(o), err(e) {}
    // per object, from measure(): Shape::bounds() in its own space, the same in its parent's space, its children
    std::vector<Bounds> own, in_parent;
    std::vector<ConeDirs> cones;  // per object, in its parent's space
    std::vector<double> ray_cost;  // per object: cost of taking a ray into its space
    std::vector<std::vector<uint32_t>> kids;
    std::vector<int32_t> leaf_id;  // per object: its number among the leaves in the reference's (depth-first) order
    std::vector<RtNode> *nodes = nullptr;  // the table being emitted
    bool counting = true;   // emitting the first table: the one whose groups and range-skip nodes are counted
    bool ordered = false;   // lists are emitted in spatial order with range-skip nodes (false: a world without groups
                            // and with a short list keeps the reference's order and gets the lean kernels)

    bool fail(int c, const char *fmt, unsigned idx) {
        char buf[160];
        std::snprintf(buf, sizeof buf, fmt, idx);
        err = buf;
        code = c;
        return false;
    }

    // Pass 1: structure and bounds of object `idx` and its subtree; `next` = index of its next sibling.
    bool measure(uint32_t idx, uint32_t depth, uint32_t &next) {
        if (idx >= world.n_objects) return fail(RT_ERR_INVALID_ARGUMENT, "object list ends inside the group before object #%u", idx);
        const rt_object_desc &o = world.objects[idx];
        if (o.kind < RT_SPHERE || o.kind > RT_GROUP) return fail(RT_ERR_INVALID_ARGUMENT, "object #%u has an unknown kind", idx);
        if (depth > out.max_nesting) out.max_nesting = depth;
        if (depth > RT_MAX_GROUP_NESTING) return fail(RT_ERR_UNSUPPORTED, "groups are nested too deep at object #%u", idx);
        if (o.kind == RT_GROUP) {
            if (o.n_children < 0) return fail(RT_ERR_INVALID_ARGUMENT, "group #%u has a negative child count", idx);
            Bounds gb = Bounds::empty();
            uint32_t child = idx + 1;
            for (int32_t k = 0; k < o.n_children; ++k) {
                if (child >= world.n_objects) return fail(RT_ERR_INVALID_ARGUMENT, "object list ends inside group #%u", idx);
                uint32_t after = 0;
                if (!measure(child, depth + 1, after)) return false;
                kids[idx].push_back(child);
                gb = gb.merge(in_parent[child]);  // groups.rs:143-145
                child = after;
            }
            own[idx] = gb;
            next = child;
        } else {
            own[idx] = leaf_bounds(o);
            leaf_id[idx] = (int32_t)out.n_leaves++;
            next = idx + 1;
        }
        const Matrix4 T = Matrix4::from_array(o.transformation);
        {
            // (scale and translation only, by the forward matrix: its inverse then has the same structure)
            bool diagonal = true;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    if (r != c && T.d[r][c] != 0.0) diagonal = false;
            ray_cost[idx] = diagonal ? COST_RAY_DIAGONAL : COST_RAY_GENERAL;
        }
        if (o.kind == RT_CONE || (o.kind == RT_GROUP && !kids[idx].empty())) {
            const auto Tinv = T.inverse();
            if (o.kind == RT_CONE) {
                cones[idx] = ConeDirs::of_cone(Tinv ? &*Tinv : nullptr);
            } else {
                ConeDirs below;
                for (uint32_t k : kids[idx]) below.merge(cones[k]);
                cones[idx] = below.through(Tinv ? &*Tinv : nullptr);
            }
        }
        in_parent[idx] = own[idx].transform(T);
        return true;
    }

    // A conservative box around the siblings `members`, or false when there is none worth testing.
    bool skip_box(const std::vector<uint32_t> &members, double box[6], ConeDirs &dirs) const {
        Bounds u = Bounds::empty();
        dirs = ConeDirs{};
        for (uint32_t k : members) {
            u = u.merge(in_parent[k]);
            dirs.merge(cones[k]);
        }
        if (dirs.any && !dirs.ok) return false;
        const double lo[3] = {u.min.x, u.min.y, u.min.z}, hi[3] = {u.max.x, u.max.y, u.max.z};
        for (int i = 0; i < 3; ++i) {
            if (!std::isfinite(lo[i]) || !std::isfinite(hi[i]) || lo[i] > hi[i]) return false;
            // what the device's range-skip test assumes (box_may_hit in rt_device.hip)
            const double pad = RT_SKIP_PAD * 0.5 * (hi[i] - lo[i]) + 1e-7 * std::fmax(std::fabs(lo[i]), std::fabs(hi[i])) + 1e-9;
            box[i] = lo[i] - pad;
            box[3 + i] = hi[i] + pad;
        }
        return true;
    }

    void push_skip(const double box[6], const ConeDirs &dirs, uint32_t depth) {
        RtNode head;
        std::memset(&head, 0, sizeof head);
        head.tag = RT_SKIPBOX | (int32_t)(depth << RT_NODE_LEVEL_SHIFT);
        head.xform = RT_SKIP_PLAIN;
        if (dirs.any) {  // inv[0..5] = q, inv[6] = bound on |d^T q d| below which the node lets a ray through
            head.xform = RT_SKIP_CONES;
            for (int i = 0; i < 6; ++i) head.inv[i] = dirs.q[i];
            head.inv[6] = 2.0 * 1e-5 / dirs.lambda_min;  // twice EPSILON (core/floats.rs:2) / lambda
        }
        head.leaf = -1;
        // The device tests the box in f32: six floats, (min, max) per axis, each rounded away from the box's inside.
        float f[6];
        for (int axis = 0; axis < 3; ++axis) {
            float lo = (float)box[axis], hi = (float)box[3 + axis];
            if ((double)lo > box[axis]) lo = std::nextafterf(lo, -INFINITY);
            if ((double)hi < box[3 + axis]) hi = std::nextafterf(hi, INFINITY);
            f[2 * axis] = lo;
            f[2 * axis + 1] = hi;
        }
        static_assert(sizeof f == 3 * sizeof(double), "six floats fill a[0..2]");
        std::memcpy(&head.a[0], f, sizeof f);
        nodes->push_back(head);
        if (counting) out.n_skip_nodes++;
    }

    // The order tree of a sibling list: the members with finite bounds, cut in two along one of the axes (build_tree
    // chooses where), recursively down to single members.  A table for rays travelling towards
    // +axis lists the low half of every split first, one for rays travelling towards -axis the high half: whatever
    // a ray meets first in space it then also meets first in the table, its closest hit is found early, and the
    // range-skip boxes behind that hit are rejected by distance.  Members without finite bounds (planes, unlimited
    // cylinders and cones: no box can hold them) stay in front, in the reference's order.
    struct Plan {
        int32_t outer;  // the tree node whose box was tested last before this one is reached; -1: the list's own bounds
        bool node;      // this tree node gets a range-skip node then
        double cost;    // expected cost of everything below, the node's own test included
    };
    struct OrderNode {
        int32_t left = -1, right = -1;
        uint32_t member = 0, count = 0;
        int axis = 0;
        Bounds box = Bounds::empty();  // around the members, in the list's space
        double work = 0.0;             // cost of testing every member once
        bool can_node = false;         // a conservative range-skip box exists (skipbox, dirs)
        double skipbox[6] = {0, 0, 0, 0, 0, 0};
        ConeDirs dirs;
        std::vector<Plan> plans;
    };
    struct ListLayout {
        std::vector<uint32_t> unbounded;
        std::vector<OrderNode> tree;
        int32_t root = -1;
        double walk = 0.0;  // expected cost of walking the list once
    };
    std::vector<ListLayout> layouts;  // per object (groups: their children); the last one is the world's own list

    bool finite_bounds(uint32_t k) const {
        const Bounds &bb = in_parent[k];
        return std::isfinite(bb.min.x) && std::isfinite(bb.min.y) && std::isfinite(bb.min.z) && std::isfinite(bb.max.x) &&
               std::isfinite(bb.max.y) && std::isfinite(bb.max.z);
    }
    double centre(uint32_t k, int axis) const {
        const Bounds &bb = in_parent[k];
        return axis == 0 ? bb.min.x + bb.max.x : axis == 1 ? bb.min.y + bb.max.y : bb.min.z + bb.max.z;
    }
    static double surface(const Bounds &b) {
        const double x = b.max.x - b.min.x, y = b.max.y - b.min.y, z = b.max.z - b.min.z;
        return x * y + y * z + z * x;
    }
    double member_cost(uint32_t m) const {
        const int kind = world.objects[m].kind;
        return COST_RECORD + ray_cost[m] + (kind == RT_GROUP ? COST_BOUNDS + layouts[m].walk : COST_LEAF[kind]);
    }
    // The order tree over the members [lo, hi) of a list.  The cut is the one, along any of the three axes (members
    // sorted by their centres), for which the two sides' boxes are smallest for what they hold: surface of the side's
    // box times the cost of testing its members, summed over both sides.  Long lists are not cut closer than an eighth from either
    // end, which bounds the depth of the tree.
    int32_t build_tree(ListLayout &L, std::vector<uint32_t>::iterator lo, std::vector<uint32_t>::iterator hi) {
        const int32_t at = (int32_t)L.tree.size();
        L.tree.emplace_back();
        const size_t n = (size_t)(hi - lo);
        {
            OrderNode &t = L.tree[(size_t)at];
            t.count = (uint32_t)n;
            const std::vector<uint32_t> members(lo, hi);
            for (uint32_t m : members) {
                t.box = t.box.merge(in_parent[m]);
                t.work += member_cost(m);
            }
            t.can_node = skip_box(members, t.skipbox, t.dirs);
            if (n == 1) {
                t.member = *lo;
                return at;
            }
        }
        int axis = 0;
        size_t best_at = n / 2;
        if (n >= 3) {
            const size_t margin = n >= 16 ? n / 8 : 1;
            double best = INFINITY;
            std::vector<uint32_t> sorted(lo, hi), best_order;
            std::vector<double> right_area(n);
            for (int a = 0; a < 3; ++a) {
                std::sort(sorted.begin(), sorted.end(), [&](uint32_t x, uint32_t y) {
                    const double cx = centre(x, a), cy = centre(y, a);
                    return cx < cy || (cx == cy && x < y);
                });
                Bounds acc = Bounds::empty();
                double work = 0.0;
                for (size_t k = n; k-- > 1;) {
                    acc = acc.merge(in_parent[sorted[k]]);
                    work += member_cost(sorted[k]);
                    right_area[k] = surface(acc) * work;
                }
                acc = Bounds::empty();
                work = 0.0;
                bool better = false;
                for (size_t k = 1; k < n; ++k) {
                    acc = acc.merge(in_parent[sorted[k - 1]]);
                    work += member_cost(sorted[k - 1]);
                    const double cost = surface(acc) * work + right_area[k];
                    if (k >= margin && n - k >= margin && cost < best) {
                        best = cost;
                        best_at = k;
                        axis = a;
                        better = true;
                    }
                }
                if (better || best_order.empty()) best_order = sorted;  // (no finite cost anywhere: the middle of the x order)
            }
            std::copy(best_order.begin(), best_order.end(), lo);
        } else if (centre(*lo, 0) > centre(*(lo + 1), 0)) {
            std::iter_swap(lo, lo + 1);
        }
        const auto mid = lo + (std::ptrdiff_t)best_at;
        const int32_t l = build_tree(L, lo, mid), r = build_tree(L, mid, hi);
        OrderNode &t = L.tree[(size_t)at];
        t.axis = axis;
        t.left = l;
        t.right = r;
        return at;
    }
    // Which tree nodes become range-skip nodes: expected cost, for a wave that has just passed the box of tree node
    // `outer`, of everything below `node` -- with a range-skip node here (its test, then the rest for the share of
    // waves that pass it: the ratio of the two boxes' surfaces) or without (the rest, for every wave).  A member on
    // its own is treated the same way: a node in front of it pays when its box is small inside the outer one and
    // its own test is long.
    double plan(ListLayout &L, int32_t node, int32_t outer) {
        {
            const OrderNode &t = L.tree[(size_t)node];
            for (const Plan &e : t.plans)
                if (e.outer == outer) return e.cost;
        }
        const double outer_area = surface(L.tree[(size_t)(outer < 0 ? L.root : outer)].box);
        const double area = surface(L.tree[(size_t)node].box);
        double pass = area / outer_area;
        if (!(pass < 1.0)) pass = 1.0;  // (also when the ratio is not a number: flat or unbounded boxes)
        const bool leaf = L.tree[(size_t)node].count == 1;
        const int32_t l = L.tree[(size_t)node].left, r = L.tree[(size_t)node].right;
        double without = leaf ? L.tree[(size_t)node].work : plan(L, l, outer) + plan(L, r, outer);
        double with = INFINITY;
        if (L.tree[(size_t)node].can_node && node != outer && pass < 1.0)
            with = COST_BOX + pass * (leaf ? L.tree[(size_t)node].work : plan(L, l, node) + plan(L, r, node));
        const bool here = with < without;
        L.tree[(size_t)node].plans.push_back(Plan{outer, here, here ? with : without});
        return here ? with : without;
    }
    void build_layout(ListLayout &L, const std::vector<uint32_t> &list) {
        std::vector<uint32_t> bounded;
        for (uint32_t k : list) (ordered && finite_bounds(k) ? bounded : L.unbounded).push_back(k);
        for (uint32_t k : L.unbounded) L.walk += member_cost(k);
        if (bounded.empty()) return;
        L.tree.reserve(2 * bounded.size());
        L.root = build_tree(L, bounded.begin(), bounded.end());
        L.walk += plan(L, L.root, -1);
        if (bounded.size() >= 2) out.reordered = true;
    }
    // The two halves of a split in the order a table for octant `oct` lists them (bit a of oct: rays travel towards -a)
    static void halves(const OrderNode &t, unsigned oct, int32_t &first, int32_t &second) {
        const bool flip = ((oct >> t.axis) & 1u) != 0;
        first = flip ? t.right : t.left;
        second = flip ? t.left : t.right;
    }

    // Emits the members below `node` of a list's order tree with the range-skip nodes plan() chose.
    bool emit_tree(const ListLayout &L, int32_t node, uint32_t depth, unsigned oct, int32_t outer) {
        const OrderNode &t = L.tree[(size_t)node];
        bool here = false;
        for (const Plan &e : t.plans)
            if (e.outer == outer) here = e.node;
        const size_t ni = nodes->size();
        if (here) push_skip(t.skipbox, t.dirs, depth);
        const int32_t inner = here ? node : outer;
        if (t.count == 1) {
            if (!emit(t.member, depth, oct)) return false;
        } else {
            int32_t a, b;
            halves(t, oct, a, b);
            if (!emit_tree(L, a, depth, oct, inner) || !emit_tree(L, b, depth, oct, inner)) return false;
        }
        if (here) (*nodes)[ni].skip = (int32_t)nodes->size();
        return true;
    }

    bool emit_list(const ListLayout &L, uint32_t depth, unsigned oct) {
        const size_t first = nodes->size();
        for (uint32_t m : L.unbounded)
            if (!emit(m, depth, oct)) return false;
        if (L.root >= 0 && !emit_tree(L, L.root, depth, oct, -1)) return false;
        // The range-skip nodes of THIS list that have cones below them: when their q's agree (they do whenever the
        // cones of the list differ by rotations about their axes, uniform scales and translations), all get the
        // first one's, and the device forms d^T q d once per visit of the list instead of once per node.
        std::vector<size_t> with_q;
        for (size_t i = first; i < nodes->size(); ++i) {
            const RtNode &n = (*nodes)[i];
            if ((n.tag & 0xff) == RT_SKIPBOX && (uint32_t)((n.tag >> RT_NODE_LEVEL_SHIFT) & RT_NODE_LEVEL_MASK) == depth && n.xform == RT_SKIP_CONES) with_q.push_back(i);
        }
        bool agree = with_q.size() > 1;
        for (size_t k = 1; k < with_q.size() && agree; ++k) {
            // inv[6] = 2 EPSILON / lambda_min of the node that is to take the first node's q
            const double tol = cone_q_tolerance(2e-5 / (*nodes)[with_q[k]].inv[6]);
            for (int c = 0; c < 6; ++c) agree = agree && std::fabs((*nodes)[with_q[k]].inv[c] - (*nodes)[with_q[0]].inv[c]) <= tol;
        }
        if (agree)
            for (size_t i : with_q) {
                for (int c = 0; c < 6; ++c) (*nodes)[i].inv[c] = (*nodes)[with_q[0]].inv[c];
                (*nodes)[i].tag |= RT_NODE_LIST_Q;
            }
        return true;
    }

    // Appends object `idx` and its subtree.
    bool emit(uint32_t idx, uint32_t depth, unsigned oct) {
        const rt_object_desc &o = world.objects[idx];
        Matrix4 T = Matrix4::from_array(o.transformation);
        auto Tinv = T.inverse();
        if (!Tinv) return fail(RT_ERR_NOT_INVERTIBLE, o.kind == RT_GROUP ? "group transformation should be invertible (object #%u)"
                                                                         : "shape transformation should be invertible (object #%u)", idx);
        size_t ni = nodes->size();
        nodes->emplace_back();
        {
            RtNode &n = (*nodes)[ni];
            std::memset(&n, 0, sizeof n);
            rows3(*Tinv, n.inv);
            n.xform = classify(n.inv);
            n.tag = o.kind | (o.closed ? RT_NODE_CLOSED : 0) | (int32_t)(depth << RT_NODE_LEVEL_SHIFT);
            n.leaf = -1;
            double centre[3], radius2;
            if (o.kind == RT_SPHERE && n.xform != RT_XFORM_GENERAL && ball_of(n.inv, centre, radius2)) {
                n.xform = RT_XFORM_DIAGONAL_BALL;
                n.inv[1] = centre[0]; n.inv[2] = centre[1]; n.inv[4] = centre[2];
                n.inv[6] = radius2;
            }
        }

        if (o.kind == RT_GROUP) {
            if (counting) out.n_groups++;
            if (!emit_list(layouts[idx], depth + 1, oct)) return false;
            const Bounds &gb = own[idx];
            RtNode &n = (*nodes)[ni];
            n.a[0] = gb.min.x; n.a[1] = gb.min.y; n.a[2] = gb.min.z;
            n.a[3] = gb.max.x; n.a[4] = gb.max.y; n.a[5] = gb.max.z;
            n.skip = (int32_t)nodes->size();
            return true;
        }

        Matrix4 W = Matrix4::from_array(o.world_transformation);
        auto Winv = W.inverse();
        if (!Winv) return fail(RT_ERR_NOT_INVERTIBLE, "world transformation should be invertible (object #%u)", idx);
        RtSurface s;
        std::memset(&s, 0, sizeof s);
        rows3(*Winv, s.inv_world);
        s.xform = classify(s.inv_world);
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
        RtNode &n = (*nodes)[ni];
        n.a[0] = o.minimum;
        n.a[1] = o.maximum;
        n.leaf = leaf_id[idx];
        n.skip = (int32_t)ni + 1;
        out.surfaces[(size_t)leaf_id[idx]] = s;
        return true;
    }
};

}  // namespace

// Range-skip nodes whose `skip` record is another range-skip node of the same list take that node's test along
// (RT_NODE_LOOKAHEAD in rt_device_types.h): about every fifth record a wave fetches is such a successor.
static void add_lookahead(std::vector<RtNode> &table) {
    const size_t n = table.size() - 1;  // without the inert record
    for (size_t i = 0; i < n; ++i) {
        RtNode &x = table[i];
        if ((x.tag & 0xff) != RT_SKIPBOX || x.skip <= 0 || (size_t)x.skip >= n) continue;
        const RtNode &r = table[(size_t)x.skip];
        if ((r.tag & 0xff) != RT_SKIPBOX) continue;
        if (((r.tag >> RT_NODE_LEVEL_SHIFT) & RT_NODE_LEVEL_MASK) != ((x.tag >> RT_NODE_LEVEL_SHIFT) & RT_NODE_LEVEL_MASK)) continue;
        double bound = 0.0;
        if (r.xform == RT_SKIP_CONES) {  // R's cones must see the directions X's do (one q per node: rt_flatten's ConeDirs)
            if (x.xform != RT_SKIP_CONES) continue;
            bool same = true;
            const double tol = cone_q_tolerance(2e-5 / r.inv[6]);  // R is tested with X's q
            for (int k = 0; k < 6; ++k) same = same && std::fabs(x.inv[k] - r.inv[k]) <= tol;
            if (!same) continue;
            bound = r.inv[6];
        }
        float box[6];  // R's own box, as R holds it
        std::memcpy(box, &r.a[0], sizeof box);
        bool finite = true;
        for (int k = 0; k < 6; ++k) finite = finite && std::isfinite(box[k]);
        if (!finite) continue;
        static_assert(sizeof(box) == 3 * sizeof(double), "six floats fill inv[7..9]");
        std::memcpy(&x.inv[7], box, sizeof box);
        x.inv[10] = bound;
        x.leaf = r.skip;
        x.tag |= RT_NODE_LOOKAHEAD;
    }
}

int flatten_world(const rt_world_desc &world, FlatScene &out, std::string &err) {
    out = FlatScene{};
    if (world.n_objects && !world.objects) {
        err = "rt_world_desc.objects is NULL";
        return RT_ERR_INVALID_ARGUMENT;
    }
    Builder b(world, out, err);
    b.own.resize(world.n_objects);
    b.in_parent.resize(world.n_objects);
    b.kids.resize(world.n_objects);
    b.cones.resize(world.n_objects);
    b.ray_cost.assign(world.n_objects, COST_RAY_GENERAL);
    b.leaf_id.assign(world.n_objects, -1);
    std::vector<uint32_t> top;
    uint32_t idx = 0;
    while (idx < world.n_objects) {
        uint32_t next = 0;
        if (!b.measure(idx, 0, next)) return b.code;
        top.push_back(idx);
        idx = next;
    }
    if (out.max_nesting > RT_MAX_GROUP_NESTING) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "groups are nested %u levels deep; at most %d are supported", out.max_nesting,
                      RT_MAX_GROUP_NESTING);
        err = buf;
        return RT_ERR_UNSUPPORTED;
    }
    out.surfaces.resize(out.n_leaves);
    bool any_group = false;
    for (uint32_t k = 0; k < world.n_objects; ++k) any_group = any_group || world.objects[k].kind == RT_GROUP;
    b.ordered = any_group || top.size() > SKIP_MIN_LIST;
    b.layouts.resize((size_t)world.n_objects + 1);
    for (uint32_t k = world.n_objects; k-- > 0;)  // children before their parents: a group's cost includes its list's
        if (world.objects[k].kind == RT_GROUP) b.build_layout(b.layouts[k], b.kids[k]);
    b.build_layout(b.layouts[world.n_objects], top);
    // One table per octant of ray directions (an ordered world), or the one table in the reference's order.  Octants
    // whose tables come out identical (no split along some axis anywhere in the world) share one.
    std::vector<std::vector<RtNode>> tables;
    const unsigned n_oct = b.ordered ? 8u : 1u;
    static const unsigned oct_mask = (unsigned)env_size("RT_OCT_MASK", 7);  // (developer override: axes whose sign picks a table)
    for (unsigned oct_index = 0; oct_index < n_oct; ++oct_index) {
        const unsigned oct = oct_index & oct_mask;
        std::vector<RtNode> t;
        t.reserve(world.n_objects + world.n_objects / 2 + 1);
        b.nodes = &t;
        b.counting = oct_index == 0;
        if (!b.emit_list(b.layouts[world.n_objects], 0, oct)) return b.code;
        t.emplace_back();  // one inert record past the end: the traversal fetches record i+1 while it works on record i
        std::memset(&t.back(), 0, sizeof(RtNode));
        add_lookahead(t);
        size_t same = tables.size();
        for (size_t k = 0; k < tables.size(); ++k)
            if (tables[k].size() == t.size() && std::memcmp(tables[k].data(), t.data(), t.size() * sizeof(RtNode)) == 0) same = k;
        out.oct_table[oct_index] = (uint8_t)same;
        if (same == tables.size()) tables.push_back(std::move(t));
    }
    for (unsigned oct = n_oct; oct < 8; ++oct) out.oct_table[oct] = 0;
    out.table_records = (uint32_t)tables[0].size();
    out.n_tables = (uint32_t)tables.size();
    out.nodes.clear();
    for (const auto &t : tables) out.nodes.insert(out.nodes.end(), t.begin(), t.end());
    out.light.has_light = world.has_light ? 1 : 0;
    for (int i = 0; i < 3; ++i) {
        out.light.position[i] = world.light_position[i];
        out.light.intensity[i] = world.light_intensity[i];
    }
    // RT_SURFACE_LIGHT_OUTSIDE: the light in the sphere's own space, |p|^2 >= 4 (anything not finite fails the test).
    // The device decides "strictly inside" from inverse(world_transformation) applied to a point on the sphere, the
    // reference from the local inverses of the enclosing groups one after the other (world.rs:53-58): the two agree to
    // within a few ulp of the largest term of the products, 1e-15 * sum_j |m_ij| |x_j|.  With that sum bounded by 1e6 for
    // every point of the sphere they agree to 1e-8 in the sphere's space, a hundredth of the margin the device demands.
    if (world.has_light)
        for (RtSurface &s : out.surfaces) {
            if (s.kind != RT_SPHERE) continue;
            const double *m = s.inv_world, *l = world.light_position;
            auto W = Matrix4::from_array(world.objects[s.object].world_transformation);
            const Vec3 centre = mul_point(W, Vec3{0.0, 0.0, 0.0});
            double radius = 0.0;  // no point of the sphere is further than this from the centre along any world axis (largest row sum)
            for (int r = 0; r < 3; ++r) {
                const double *w = world.objects[s.object].world_transformation + 4 * r;
                radius = std::fmax(radius, std::fabs(w[0]) + std::fabs(w[1]) + std::fabs(w[2]));
            }
            double p[3], terms = 0.0;
            for (int r = 0; r < 3; ++r) {
                p[r] = m[4 * r] * l[0] + m[4 * r + 1] * l[1] + m[4 * r + 2] * l[2] + m[4 * r + 3];
                const double t = std::fabs(m[4 * r]) * (std::fabs(centre.x) + radius) + std::fabs(m[4 * r + 1]) * (std::fabs(centre.y) + radius) +
                                 std::fabs(m[4 * r + 2]) * (std::fabs(centre.z) + radius) + std::fabs(m[4 * r + 3]);
                terms = std::fmax(terms, t);
            }
            const double d2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
            if (d2 >= 4.0 && d2 < 1e300 && terms <= 1e6) s.flags |= RT_SURFACE_LIGHT_OUTSIDE;
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
