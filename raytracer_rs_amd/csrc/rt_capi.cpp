// C ABI of include/raytracer_rs_amd.h over the HIP runtime.  No CPU rendering path exists in this library: every
// rendering entry point needs a HIP device and fails with RT_ERR_NO_DEVICE / RT_ERR_HIP otherwise.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cerrno>
#include <cstddef>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/raytracer_rs_amd.h"
#include "rt_capi_internal.hpp"
#include "rt_device.hpp"
#include "rt_flatten.hpp"
#include "rt_hostcopy.hpp"
#include "rt_loader.hpp"
#include "rt_math.hpp"

// What rt_render keeps between calls on one handle: the frame in device memory, pinned staging memory of the same
// size, two streams (kernels; device-to-host copies) and the events that order them.  Grown on demand, freed with the
// handle.
struct RenderPipe {
    enum { MAX_GROUPS = 8 };
    double *d_frame = nullptr, *h_stage = nullptr;
    size_t d_cap = 0, h_cap = 0;  // bytes
    hipStream_t s_render = nullptr, s_copy = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_rendered[MAX_GROUPS] = {}, ev_copied[MAX_GROUPS] = {};
};

struct rt_scene {
    int device = 0;
    rt::FlatScene flat;
    RtNode *d_nodes = nullptr;
    RtSurface *d_surfaces = nullptr;
    unsigned long long *d_counters = nullptr;
    uint32_t *d_tiles = nullptr;  // tile-ordering scratch of launch_render, grown on demand
    size_t tiles_cap = 0;         // in 32-bit words
    bool uneven = false;          // tiles differ in cost by more than a shadow ray: ordering them pays
    // Large hierarchies are rendered level by level (rt::launch_render_levels) instead of by one launch: decided once,
    // when the scene is created, so that every frame, row range and band of a scene comes from the same arithmetic.
    bool by_levels = false;
    rt::LevelArena levels;        // that renderer's queues and sums, grown on demand, kept between frames
    // Launch orders worked out so far, each in scratch of its own, keyed by the view they belong to (camera and rows):
    // a caller that renders the same rows through the same camera again -- a frame loop, the bands of rt_multi --
    // finds the order ready.  Least recently used entries make room (ORDER_CACHE_ENTRIES, ORDER_CACHE_BYTES).
    struct OrderEntry {
        RtCameraDev cam{};
        uint32_t rows[5] = {};
        uint32_t *d_tiles = nullptr;
        size_t cap = 0;  // in 32-bit words
        uint64_t used = 0;
    };
    std::vector<OrderEntry> orders;
    uint64_t order_clock = 0;
    hipEvent_t marks[32][3] = {};  // event triples of the launches made with RT_RENDER_TIMED, used round-robin
    uint32_t timed_launches = 0;
    RenderPipe pipe;
    rt::DeviceScene dev;
    std::vector<int32_t> object_to_leaf;
    uint64_t device_bytes = 0;
};

struct rt_scene_file {
    rt::SceneFile file;
    std::vector<rt_object_desc> objects;
};

namespace {

// Which worlds are rendered level by level (rt::launch_render_levels): every world with groups.  Measured on MI355X, GPU
// time per frame of a repeated view, single launch / levels: chapter_14 (327 leaves) 1.68 / 0.68 ms at 320x200, 2.01 /
// 0.71 at 1280x720, 4.30 / 3.21 at 3840x2160; generated hierarchies of 101 ... 10 001 leaves at 3840x2160 between 30 % and
// 56 % faster (the single launch waits for its longest waves: the tiles whose 64 ray trees are all deep).  The price: a
// view's first frame waits for the GPU after every level, and the handle keeps ray queues that grow with the frame.
// A world WITHOUT groups whose table has range-skip nodes (more than twelve objects) keeps the single launch: chapter_12
// (13 cubes) 0.89 / 1.09 ms at 3840x2160 -- its ray trees are shallow and a queue entry costs more than it saves.
// RT_RENDER_LEVELS=0 / 1 in the environment, read when a scene is created, forces the choice for every world whose table
// has groups or range-skip nodes (a table of leaves only always takes the single launch).
bool render_by_levels(int max_nesting, uint32_t n_groups) {
    if (max_nesting <= 0) return false;
    const char *e = getenv("RT_RENDER_LEVELS");
    if (e && (e[0] == '0' || e[0] == '1') && e[1] == 0) return e[0] == '1';
    return n_groups > 0;
}

thread_local std::string g_error;

int set_error(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
int hip_error(hipError_t e, const char *what) {
    return set_error(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? RT_ERR_NO_DEVICE : RT_ERR_HIP,
                     std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return hip_error(_e, #expr); \
    } while (0)

// Makes `device` current for the scope and restores the previous device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        status = hipGetDevice(&prev);
        if (status == hipSuccess && prev != device) {
            status = hipSetDevice(device);
            switched = status == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

template <typename T>
struct DeviceBuffer {
    T *p = nullptr;
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};

void stats_from_counters(const unsigned long long *c, rt_stats *s) {
    s->rays = c[0]; s->primary = c[1]; s->shadow = c[2]; s->reflect = c[3]; s->refract = c[4];
    for (int i = 0; i < 5; ++i) { s->prim_tests[i] = c[5 + i]; s->prim_hits[i] = c[10 + i]; }
    s->aabb_tests = c[15]; s->ray_transforms = c[16]; s->shades = c[17]; s->lit = c[18]; s->specular = c[19];
    s->patterns = c[20]; s->refract_setups = c[21]; s->schlick = c[22];
}

int clamp_depth(int d) {
    if (d <= 0) return RT_DEFAULT_DEPTH;
    return d > RT_MAX_DEPTH ? RT_MAX_DEPTH : d;
}

// Validates a render call and works out which rows it covers.
int render_rows(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts, rt::RowMap &rows,
                int &depth, RtCameraDev &cam) {
    if (!scene || !camera) return set_error(RT_ERR_INVALID_ARGUMENT, "scene and camera must not be NULL");
    if (camera->hsize == 0 || camera->vsize == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "camera has zero size");
    uint32_t row_begin = 0, row_end = camera->vsize;
    depth = RT_DEFAULT_DEPTH;
    rows = rt::RowMap{};
    if (opts) {
        if (opts->max_depth < 0 || opts->max_depth > RT_MAX_DEPTH)
            return set_error(RT_ERR_INVALID_ARGUMENT, "max_depth must be within 0.." + std::to_string(RT_MAX_DEPTH));
        depth = (opts->flags & RT_RENDER_EXACT_DEPTH) ? opts->max_depth : clamp_depth(opts->max_depth);
        if (opts->flags & ~(RT_RENDER_TIMED | RT_RENDER_QUANTIZED | RT_RENDER_EXACT_DEPTH | RT_RENDER_IN_PLACE)) return set_error(RT_ERR_INVALID_ARGUMENT, "unknown bits in flags");
        rows.in_place = (opts->flags & RT_RENDER_IN_PLACE) ? 1u : 0u;
        if (opts->band_rows) {
            if (opts->band_step == 0) return set_error(RT_ERR_INVALID_ARGUMENT, "band_step must be positive");
            const uint32_t n_bands = (camera->vsize + opts->band_rows - 1) / opts->band_rows;
            uint32_t mine = opts->band_first < n_bands ? (n_bands - opts->band_first + opts->band_step - 1) / opts->band_step : 0;
            rows.band_rows = opts->band_rows;
            rows.band_pitch = opts->band_step * opts->band_rows;
            rows.row_begin = opts->band_first * opts->band_rows;
            rows.row_end = camera->vsize;
            rows.n_rows = 0;
            if (mine) {
                uint32_t last_start = (opts->band_first + (mine - 1) * opts->band_step) * opts->band_rows;
                uint32_t last_len = camera->vsize - last_start < opts->band_rows ? camera->vsize - last_start : opts->band_rows;
                rows.n_rows = (mine - 1) * opts->band_rows + last_len;
            }
        } else if (opts->row_begin || opts->row_end) {
            row_begin = opts->row_begin;
            row_end = opts->row_end;
        }
    }
    if (!(opts && opts->band_rows)) {
        if (row_begin > row_end || row_end > camera->vsize)
            return set_error(RT_ERR_INVALID_ARGUMENT, "row range lies outside the frame");
        rows.row_begin = row_begin;
        rows.row_end = row_end;
        rows.n_rows = row_end - row_begin;
        rows.band_rows = rows.n_rows ? rows.n_rows : 1;
        rows.band_pitch = 0;
    }
    std::string err;
    int rc = rt::camera_to_device(*camera, cam, err);
    if (rc != RT_OK) return set_error(rc, err);
    return RT_OK;
}

}  // namespace

int rt::capi_set_error(int code, const std::string &msg) { return set_error(code, msg); }

static int render_device_impl(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts,
                              double *rgb_device, rt_stats *stats, bool accumulate);

int rt::capi_render_device_counting(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb_device) {
    return render_device_impl(scene, camera, opts, rgb_device, nullptr, true);
}

int rt::capi_counters_reset(const rt_scene *scene, void *stream) {
    if (!scene) return set_error(RT_ERR_INVALID_ARGUMENT, "scene must not be NULL");
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    HIP_TRY(hipMemsetAsync(scene->d_counters, 0, RT_NUM_COUNTERS * sizeof(unsigned long long), (hipStream_t)stream));
    return RT_OK;
}

int rt::capi_counters_read(const rt_scene *scene, void *stream, rt_stats *stats) {
    if (!scene || !stats) return set_error(RT_ERR_INVALID_ARGUMENT, "scene and stats must not be NULL");
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    unsigned long long c[RT_NUM_COUNTERS];
    HIP_TRY(hipMemcpyAsync(c, scene->d_counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::memset(stats, 0, sizeof *stats);
    stats_from_counters(c, stats);
    return RT_OK;
}

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(int *count) {
    if (!count) return set_error(RT_ERR_INVALID_ARGUMENT, "count must not be NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return hip_error(e, "hipGetDeviceCount");
    }
    *count = n;
    return RT_OK;
}

int rt_scene_create(const rt_world_desc *world, int device, rt_scene **out) {
    if (!world || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "world and out must not be NULL");
    *out = nullptr;
    auto sc = std::make_unique<rt_scene>();
    std::string err;
    int rc = rt::flatten_world(*world, sc->flat, err);
    if (rc != RT_OK) return set_error(rc, err);
    if (sc->flat.nodes.size() * sizeof(RtNode) >= (1ull << 32))  // the traversal addresses records by 32-bit byte offsets
        return set_error(RT_ERR_UNSUPPORTED, "the world has too many objects for one node table (4 GB of records)");
    sc->object_to_leaf.assign(world->n_objects, -1);
    for (size_t leaf = 0; leaf < sc->flat.surfaces.size(); ++leaf)
        sc->object_to_leaf[(size_t)sc->flat.surfaces[leaf].object] = (int32_t)leaf;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return set_error(RT_ERR_NO_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return set_error(RT_ERR_NO_DEVICE, "device ordinal out of range");
    sc->device = device;
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    // Set-up for rt_render_device_noalloc: the address of this GPU's copy of the static scratch array, looked up once
    // per device while that device is current.  A failure only disables that entry point (it reports it).
    (void)rt::static_render_scratch_resolve(device);

    // A table of leaves only (what the kernels built for worlds without groups walk) lies in device memory TWICE, one
    // copy after the other in front of the inert record: a walk that starts at any record and takes as many records
    // as the table has then needs no wrap-round (the shadow rays of rt::render_kernel start at the object they leave).
    const bool leaves_only = sc->flat.max_nesting == 0 && !sc->flat.n_skip_nodes && !sc->flat.n_groups && !sc->flat.reordered && sc->flat.n_tables == 1;
    std::vector<RtNode> doubled;
    if (leaves_only && sc->flat.table_records > 1) {
        const size_t n = sc->flat.table_records - 1;
        doubled.assign(sc->flat.nodes.begin(), sc->flat.nodes.begin() + (ptrdiff_t)n);
        doubled.insert(doubled.end(), sc->flat.nodes.begin(), sc->flat.nodes.end());  // the second copy and the inert record
    }
    const std::vector<RtNode> &table = doubled.empty() ? sc->flat.nodes : doubled;
    size_t nb = table.size() * sizeof(RtNode), sb = sc->flat.surfaces.size() * sizeof(RtSurface);
    e = hipMalloc((void **)&sc->d_nodes, nb);  // (every table carries its own inert trailing record)
    if (e == hipSuccess) e = hipMalloc((void **)&sc->d_surfaces, sb ? sb : sizeof(RtSurface));
    if (e == hipSuccess) e = hipMalloc((void **)&sc->d_counters, RT_NUM_COUNTERS * sizeof(unsigned long long));
    if (e == hipSuccess && nb) e = hipMemcpy(sc->d_nodes, table.data(), nb, hipMemcpyHostToDevice);
    if (e == hipSuccess && sb) e = hipMemcpy(sc->d_surfaces, sc->flat.surfaces.data(), sb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rt_scene *raw = sc.release();
        rt_scene_destroy(raw);
        return hip_error(e, "uploading the scene");
    }
    sc->device_bytes = nb + sb + RT_NUM_COUNTERS * sizeof(unsigned long long);
    sc->dev.nodes = sc->d_nodes;
    sc->dev.surfaces = sc->d_surfaces;
    sc->dev.n_nodes = (int32_t)sc->flat.table_records - 1;
    sc->dev.table_records = sc->flat.table_records;
    sc->dev.oct_tables = 0;
    for (int oct = 0; oct < 8; ++oct) sc->dev.oct_tables |= (unsigned long long)sc->flat.oct_table[oct] << (8 * oct);
    sc->dev.n_leaves = (int32_t)sc->flat.surfaces.size();
    // 0 selects the kernels built for tables of leaves only; a group (even a childless one at the top level) or a
    // range-skip node needs the general ones
    sc->dev.max_nesting = (int32_t)sc->flat.max_nesting;
    if (!leaves_only && sc->dev.max_nesting < 1) sc->dev.max_nesting = 1;
    sc->dev.any_order = sc->flat.reordered ? 1 : 0;
    sc->dev.cone_nodes = 0;
    for (const RtNode &n : sc->flat.nodes)
        if ((n.tag & 0xff) == RT_SKIPBOX && n.xform == RT_SKIP_CONES) sc->dev.cone_nodes = 1;
    sc->dev.light = sc->flat.light;
    // In a flat scene without reflective or transparent surfaces every pixel costs one or two rays over the same
    // few objects: nothing to gain from launching the tiles in any particular order.
    sc->uneven = sc->dev.max_nesting > 0;
    sc->by_levels = render_by_levels(sc->dev.max_nesting, sc->flat.n_groups);
    for (const RtSurface &s : sc->flat.surfaces) sc->uneven = sc->uneven || s.reflectivity != 0.0 || s.transparency != 0.0;
    sc->dev.no_secondary = 1;
    for (const RtSurface &s : sc->flat.surfaces)
        if (s.reflectivity != 0.0 || s.transparency != 0.0) sc->dev.no_secondary = 0;
    *out = sc.release();
    return RT_OK;
}

void rt_scene_destroy(rt_scene *scene) {
    if (!scene) return;
    {
        DeviceGuard guard(scene->device);
        if (scene->d_nodes) (void)hipFree(scene->d_nodes);
        if (scene->d_surfaces) (void)hipFree(scene->d_surfaces);
        if (scene->d_counters) (void)hipFree(scene->d_counters);
        if (scene->d_tiles) (void)hipFree(scene->d_tiles);
        rt::level_arena_release(scene->levels);
        for (auto &e : scene->orders)
            if (e.d_tiles) (void)hipFree(e.d_tiles);
        RenderPipe &p = scene->pipe;
        if (p.d_frame) (void)hipFree(p.d_frame);
        if (p.h_stage) (void)hipHostFree(p.h_stage);
        if (p.s_render) (void)hipStreamDestroy(p.s_render);
        if (p.s_copy) (void)hipStreamDestroy(p.s_copy);
        if (p.ev_begin) (void)hipEventDestroy(p.ev_begin);
        if (p.ev_end) (void)hipEventDestroy(p.ev_end);
        for (int g = 0; g < RenderPipe::MAX_GROUPS; ++g) {
            if (p.ev_rendered[g]) (void)hipEventDestroy(p.ev_rendered[g]);
            if (p.ev_copied[g]) (void)hipEventDestroy(p.ev_copied[g]);
        }
        for (auto &triple : scene->marks)
            for (hipEvent_t ev : triple)
                if (ev) (void)hipEventDestroy(ev);
    }
    delete scene;
}

int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out) {
    if (!scene || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "scene and out must not be NULL");
    std::memset(out, 0, sizeof *out);
    out->n_objects = (uint32_t)(scene->flat.table_records - 1 - scene->flat.n_skip_nodes);
    out->n_leaves = scene->flat.n_leaves;
    out->n_groups = scene->flat.n_groups;
    out->max_nesting = scene->flat.max_nesting;
    out->device_bytes = scene->device_bytes;
    out->device = scene->device;
    out->kernel_build = scene->dev.max_nesting == 0 ? (scene->dev.no_secondary ? 4 : 0) : scene->by_levels ? 3 : (scene->dev.any_order != 0 && scene->dev.cone_nodes == 0) ? 2 : 1;
    out->device_bytes += scene->levels.device_bytes();
    return RT_OK;
}

// Tile-ordering scratch of the handle, grown on demand (the handle is used by one render at a time).  RT_TILE_ORDER=0
// in the environment turns the ordering off (developer A/B switch).
static bool tile_order_enabled() {
    static const bool enabled = [] { const char *e = getenv("RT_TILE_ORDER"); return !(e && e[0] == '0'); }();
    return enabled;
}
static bool tile_order_cache_enabled() {
    static const bool enabled = [] { const char *e = getenv("RT_TILE_ORDER_CACHE"); return !(e && e[0] == '0'); }();
    return enabled;
}
// `cached` (optional): set when the scratch already holds the launch order of exactly these rows seen through exactly
// this camera -- the order is a function of world, camera and rows only, so a caller that renders the same view again
// (a frame loop, a benchmark) need not have it worked out again.  Callers that pass nullptr share the scratch between
// launches over different rows and get no caching.
enum : size_t { ORDER_CACHE_ENTRIES = 64, ORDER_CACHE_BYTES = 64u << 20 };
static hipError_t tile_scratch(const rt_scene *scene_c, const rt_camera_desc *camera, const rt::RowMap &rows, const RtCameraDev *cam,
                               uint32_t **out, bool *cached) {
    *out = nullptr;
    if (cached) *cached = false;
    if (!tile_order_enabled() || rows.n_rows == 0 || !scene_c->uneven) return hipSuccess;
    rt_scene *scene = const_cast<rt_scene *>(scene_c);
    const size_t need = rt::render_scratch_words(camera->hsize, rows.n_rows);
    if (!cached || !cam || !tile_order_cache_enabled()) {  // the handle's one shared scratch
        if (need > scene->tiles_cap) {
            if (scene->d_tiles) (void)hipFree(scene->d_tiles);
            scene->d_tiles = nullptr;
            scene->tiles_cap = 0;
            hipError_t e = hipMalloc((void **)&scene->d_tiles, need * sizeof(uint32_t));
            if (e != hipSuccess) return e;
            scene->tiles_cap = need;
        }
        *out = scene->d_tiles;
        return hipSuccess;
    }
    const uint32_t key[5] = {rows.row_begin, rows.row_end, rows.n_rows, rows.band_rows, rows.band_pitch};
    auto &orders = scene->orders;
    for (auto &e : orders)
        if (std::memcmp(&e.cam, cam, sizeof *cam) == 0 && std::memcmp(e.rows, key, sizeof key) == 0 && e.cap >= need) {
            e.used = ++scene->order_clock;
            *out = e.d_tiles;
            *cached = true;
            return hipSuccess;
        }
    // a new view: an entry of its own, after the least recently used ones have made room
    size_t total = need * sizeof(uint32_t);
    for (const auto &e : orders) total += e.cap * sizeof(uint32_t);
    while (!orders.empty() && (orders.size() >= ORDER_CACHE_ENTRIES || total > ORDER_CACHE_BYTES)) {
        size_t lru = 0;
        for (size_t k = 1; k < orders.size(); ++k)
            if (orders[k].used < orders[lru].used) lru = k;
        total -= orders[lru].cap * sizeof(uint32_t);
        (void)hipFree(orders[lru].d_tiles);  // (waits for the device: nothing that reads the entry is still running)
        orders.erase(orders.begin() + (std::ptrdiff_t)lru);
    }
    rt_scene::OrderEntry e;
    hipError_t err = hipMalloc((void **)&e.d_tiles, need * sizeof(uint32_t));
    if (err != hipSuccess) return err;
    e.cap = need;
    e.cam = *cam;
    std::memcpy(e.rows, key, sizeof key);
    e.used = ++scene->order_clock;
    orders.push_back(e);
    *out = e.d_tiles;
    return hipSuccess;
}

// rt_render_device, and the variant rt_multi uses when it collects statistics: `accumulate` launches the counting
// kernel so that it adds to the handle's counters as they stand -- no clearing before, no read-back and therefore no
// host synchronisation after (rt::capi_counters_reset / rt::capi_counters_read bracket a run of such launches).
static int render_device_impl(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts,
                              double *rgb_device, rt_stats *stats, bool accumulate) {
    rt::RowMap rows;
    int depth;
    RtCameraDev cam;
    int rc = render_rows(scene, camera, opts, rows, depth, cam);
    if (rc != RT_OK) return rc;
    if (!rgb_device && rows.n_rows) return set_error(RT_ERR_INVALID_ARGUMENT, "rgb_device must not be NULL");
    const bool want_stats = accumulate || (opts && opts->collect_stats);
    if (want_stats && !stats && !accumulate) return set_error(RT_ERR_INVALID_ARGUMENT, "collect_stats needs a stats pointer");
    hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (want_stats && !accumulate) HIP_TRY(hipMemsetAsync(scene->d_counters, 0, RT_NUM_COUNTERS * sizeof(unsigned long long), stream));
    const bool by_levels = scene->by_levels && !want_stats;  // (the counting kernel is the single launch: it visits what the reference visits)
    uint32_t *tiles = nullptr;
    bool order_cached = false;
    if (!by_levels) HIP_TRY(tile_scratch(scene, camera, rows, &cam, &tiles, &order_cached));  // (levels need no launch order)
    if (opts && (opts->flags & RT_RENDER_QUANTIZED)) rows.quantized = reinterpret_cast<uint8_t *>(rgb_device);
    hipEvent_t *marks = nullptr;
    if (opts && (opts->flags & RT_RENDER_TIMED)) {
        rt_scene *mut = const_cast<rt_scene *>(scene);  // one render at a time per handle
        marks = mut->marks[mut->timed_launches % 32];
        for (int k = 0; k < 3; ++k)
            if (!marks[k]) HIP_TRY(hipEventCreate(&marks[k]));
        mut->timed_launches++;
    }
    if (by_levels)
        HIP_TRY(rt::launch_render_levels(scene->dev, cam, rows, depth, rgb_device, const_cast<rt_scene *>(scene)->levels, stream, marks));
    else
        HIP_TRY(rt::launch_render(scene->dev, cam, rows, depth, rgb_device, want_stats ? scene->d_counters : nullptr, tiles, stream, marks, order_cached));
    if (want_stats && !accumulate) {
        unsigned long long c[RT_NUM_COUNTERS];
        HIP_TRY(hipMemcpyAsync(c, scene->d_counters, sizeof c, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        stats_from_counters(c, stats);
    }
    return RT_OK;
}

int rt_render_device(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts,
                     double *rgb_device, rt_stats *stats) {
    return render_device_impl(scene, camera, opts, rgb_device, stats, false);
}

int rt_render_device_noalloc(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts,
                             double *rgb_device) {
    rt::RowMap rows;
    int depth;
    RtCameraDev cam;
    int rc = render_rows(scene, camera, opts, rows, depth, cam);
    if (rc != RT_OK) return rc;
    if (!rgb_device && rows.n_rows) return set_error(RT_ERR_INVALID_ARGUMENT, "rgb_device must not be NULL");
    if (opts && opts->collect_stats)
        return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc cannot collect stats (that needs a host synchronisation)");
    if (opts && (opts->flags & RT_RENDER_TIMED))
        return set_error(RT_ERR_INVALID_ARGUMENT, "rt_render_device_noalloc cannot time the launch (that needs event creation)");
    if (scene->by_levels) {
        // A world rendered level by level needs ray queues that grow with the frame and, for a view it has not rendered
        // before, a wait after every level.  Neither is allowed here: the call is served only when this very view has
        // been rendered through rt_render_device on this handle before (its counts are remembered, its buffers there).
        hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;
        DeviceGuard guard(scene->device);
        if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
        if (opts && (opts->flags & RT_RENDER_QUANTIZED)) rows.quantized = reinterpret_cast<uint8_t *>(rgb_device);
        rt::LevelArena &arena = const_cast<rt_scene *>(scene)->levels;  // one render at a time per handle
        arena.frozen = true;
        const hipError_t e = rt::launch_render_levels(scene->dev, cam, rows, depth, rgb_device, arena, stream, nullptr);
        arena.frozen = false;
        if (e == hipErrorNotReady)
            return set_error(RT_ERR_UNSUPPORTED, "this world is rendered level by level (rt_scene_info.kernel_build 3), through ray queues in device memory "
                                                 "that grow with the frame: rt_render_device_noalloc serves it only for a view (camera, rows, depth) that "
                                                 "rt_render_device has rendered on this handle before");
        HIP_TRY(e);
        return RT_OK;
    }
    const unsigned long long tiles = (unsigned long long)((camera->hsize + 7) / 8) * ((rows.n_rows + 7) / 8);
    if (tiles > RT_NOALLOC_MAX_TILES)
        return set_error(RT_ERR_CAPACITY, "the rows to render cover " + std::to_string(tiles) + " 8x8-pixel tiles; rt_render_device_noalloc's static scratch holds " +
                                              std::to_string(RT_NOALLOC_MAX_TILES) + " (render fewer rows per call, or use rt_render_device)");
    hipStream_t stream = opts ? (hipStream_t)opts->stream : nullptr;
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    // The device pointer that hipGetSymbolAddress returned for this GPU when the scene was created -- never the array's
    // name, which is only a host-side handle.
    uint32_t *tiles_ptr = nullptr;
    if (scene->uneven && rows.n_rows && tile_order_enabled()) {
        tiles_ptr = rt::static_render_scratch(scene->device);
        if (!tiles_ptr) return set_error(RT_ERR_HIP, "the static scratch array could not be located on device " + std::to_string(scene->device));
    }
    if (opts && (opts->flags & RT_RENDER_QUANTIZED)) rows.quantized = reinterpret_cast<uint8_t *>(rgb_device);
    // classify_tiles_kernel clears the words of the scratch that must start at zero (the census and the cursors);
    // everything else the launch reads from it, it has written itself first.
    HIP_TRY(rt::launch_render(scene->dev, cam, rows, depth, rgb_device, nullptr, tiles_ptr, stream, nullptr));
    return RT_OK;
}

int rt_scene_render_times(const rt_scene *scene, uint32_t back, double *ordering_ms, double *render_ms) {
    if (!scene || !ordering_ms || !render_ms) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (back >= 32 || back >= scene->timed_launches)
        return set_error(RT_ERR_INVALID_ARGUMENT, "no launch with RT_RENDER_TIMED that far back");
    const hipEvent_t *marks = scene->marks[(scene->timed_launches - 1 - back) % 32];
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, marks[0], marks[1]));
    HIP_TRY(hipEventElapsedTime(&b, marks[1], marks[2]));
    *ordering_ms = a;
    *render_ms = b;
    return RT_OK;
}

// Makes sure the handle's pipeline can hold `bytes` of frame.
static hipError_t pipe_prepare(RenderPipe &p, size_t bytes) {
    hipError_t e = hipSuccess;
    if (!p.s_render) e = hipStreamCreateWithFlags(&p.s_render, hipStreamNonBlocking);
    if (e == hipSuccess && !p.s_copy) e = hipStreamCreateWithFlags(&p.s_copy, hipStreamNonBlocking);
    if (e == hipSuccess && !p.ev_begin) e = hipEventCreate(&p.ev_begin);
    if (e == hipSuccess && !p.ev_end) e = hipEventCreate(&p.ev_end);
    for (int g = 0; e == hipSuccess && g < RenderPipe::MAX_GROUPS; ++g) {
        if (!p.ev_rendered[g]) e = hipEventCreateWithFlags(&p.ev_rendered[g], hipEventDisableTiming);
        if (e == hipSuccess && !p.ev_copied[g]) e = hipEventCreateWithFlags(&p.ev_copied[g], hipEventDisableTiming);
    }
    if (e == hipSuccess && bytes > p.d_cap) {
        if (p.d_frame) (void)hipFree(p.d_frame);
        p.d_frame = nullptr;
        p.d_cap = 0;
        e = hipMalloc((void **)&p.d_frame, bytes);
        if (e == hipSuccess) p.d_cap = bytes;
    }
    if (e == hipSuccess && bytes > p.h_cap) {
        if (p.h_stage) (void)hipHostFree(p.h_stage);
        p.h_stage = nullptr;
        p.h_cap = 0;
        e = hipHostMalloc((void **)&p.h_stage, bytes, hipHostMallocDefault);
        if (e == hipSuccess) p.h_cap = bytes;
    }
    return e;
}

// Camera::render into host memory.  The rows are rendered in a few groups; while the kernels of group k+1 run, group
// k travels to pinned staging memory over PCIe and group k-1 is copied from there into the caller's frame by a few
// host threads: the call costs about what the slowest of the three stages costs for the whole frame (for a
// 3840x2160 f64 frame, 199 MB, that is the PCIe transfer), not their sum.
int rt_render(const rt_scene *scene_c, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb,
              rt_stats *stats) {
    rt::RowMap rows;
    int depth;
    RtCameraDev cam;
    int rc = render_rows(scene_c, camera, opts, rows, depth, cam);
    if (rc != RT_OK) return rc;
    if (!rgb) return set_error(RT_ERR_INVALID_ARGUMENT, "rgb must not be NULL");
    if (opts && (opts->flags & ~RT_RENDER_EXACT_DEPTH)) return set_error(RT_ERR_INVALID_ARGUMENT, "these flags are for rt_render_device");
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (rows.n_rows == 0) return RT_OK;
    const bool want_stats = opts && opts->collect_stats;
    if (want_stats && !stats) return set_error(RT_ERR_INVALID_ARGUMENT, "collect_stats needs a stats pointer");
    rt_scene *scene = const_cast<rt_scene *>(scene_c);  // one call at a time per handle: the pipeline is the handle's
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    const size_t row_vals = (size_t)camera->hsize * 3;
    const size_t row_bytes = row_vals * sizeof(double);
    const size_t bytes = (size_t)rows.n_rows * row_bytes;
    RenderPipe &p = scene->pipe;
    HIP_TRY(pipe_prepare(p, bytes));
    uint32_t *tiles = nullptr;
    HIP_TRY(tile_scratch(scene, camera, rows, nullptr, &tiles, nullptr));

    // Groups of packed rows [begin, end): whole bands (band mode) or multiples of the 8-row tile height, at least
    // 8 MB each so that no stage is dominated by its fixed costs.
    const bool banded = rows.band_pitch != 0;
    const uint32_t unit = banded ? rows.band_rows : 8u;
    uint32_t n_groups = (uint32_t)std::min<size_t>(RenderPipe::MAX_GROUPS, std::max<size_t>(1, bytes / (8u << 20)));
    const uint32_t units = (rows.n_rows + unit - 1) / unit;
    // (a world rendered level by level pays six launches and five host round trips per group, and small groups leave its
    // levels short of rays: four groups cost 4 % of the rendering time of an 8K frame, sixteen 36 %)
    if (scene->by_levels && n_groups > 4) n_groups = 4;
    if (n_groups > units) n_groups = units;
    uint32_t begin[RenderPipe::MAX_GROUPS + 1];
    for (uint32_t g = 0; g <= n_groups; ++g) begin[g] = std::min<uint64_t>(rows.n_rows, (uint64_t)units * g / n_groups * unit);

    hipError_t e = hipSuccess;
    if (want_stats) e = hipMemsetAsync(scene->d_counters, 0, RT_NUM_COUNTERS * sizeof(unsigned long long), p.s_render);
    if (e == hipSuccess) e = hipEventRecord(p.ev_begin, p.s_render);
    for (uint32_t g = 0; e == hipSuccess && g < n_groups; ++g) {
        rt::RowMap part = rows;
        part.n_rows = begin[g + 1] - begin[g];
        if (banded) {
            part.row_begin = rows.row_begin + (begin[g] / rows.band_rows) * rows.band_pitch;
        } else {
            part.row_begin = rows.row_begin + begin[g];
            part.row_end = rows.row_begin + begin[g + 1];
            part.band_rows = part.n_rows;
        }
        const size_t at = (size_t)begin[g] * row_vals;
        if (scene->by_levels && !want_stats) e = rt::launch_render_levels(scene->dev, cam, part, depth, p.d_frame + at, scene->levels, p.s_render);
        else e = rt::launch_render(scene->dev, cam, part, depth, p.d_frame + at, want_stats ? scene->d_counters : nullptr, tiles, p.s_render);
        if (e == hipSuccess && g + 1 == n_groups) e = hipEventRecord(p.ev_end, p.s_render);
        if (e == hipSuccess) e = hipEventRecord(p.ev_rendered[g], p.s_render);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.s_copy, p.ev_rendered[g], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(p.h_stage + at, p.d_frame + at, (size_t)part.n_rows * row_bytes, hipMemcpyDeviceToHost, p.s_copy);
        if (e == hipSuccess) e = hipEventRecord(p.ev_copied[g], p.s_copy);
    }
    // staged rows -> their place in the caller's whole-frame buffer
    std::vector<rt::CopySegment> segments;
    for (uint32_t g = 0; e == hipSuccess && g < n_groups; ++g) {
        e = hipEventSynchronize(p.ev_copied[g]);
        if (e != hipSuccess) break;
        segments.clear();
        for (uint32_t r = begin[g]; r < begin[g + 1]; r += rows.band_rows) {
            const uint32_t y = rows.row_begin + (r / rows.band_rows) * rows.band_pitch + (banded ? 0 : r % rows.band_rows);
            const uint32_t len = std::min(begin[g + 1] - r, rows.band_rows);
            segments.push_back(rt::CopySegment{rgb + (size_t)y * row_vals, p.h_stage + (size_t)r * row_vals, (size_t)len * row_bytes});
        }
        rt::HostCopyPool::instance().run(segments);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(p.s_render);  // nothing of this call may still be running when the caller's buffers go away
        (void)hipStreamSynchronize(p.s_copy);
        return hip_error(e, "rt_render");
    }
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p.ev_begin, p.ev_end);
    unsigned long long c[RT_NUM_COUNTERS];
    if (e == hipSuccess && want_stats) e = hipMemcpy(c, scene->d_counters, sizeof c, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_error(e, "rt_render");
    if (stats) {
        if (want_stats) stats_from_counters(c, stats);
        stats->kernel_ms = ms;
    }
    return RT_OK;
}

int rt_color_at(const rt_scene *scene, const double *rays, uint32_t n, int32_t remaining, double *colors,
                rt_stats *stats) {
    if (!scene || ((!rays || !colors) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (remaining < 0 || remaining > RT_MAX_DEPTH)
        return set_error(RT_ERR_INVALID_ARGUMENT, "remaining must be within 0.." + std::to_string(RT_MAX_DEPTH));
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (!n) return RT_OK;
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<double> d_rays, d_colors;
    HIP_TRY(d_rays.alloc((size_t)n * 6));
    HIP_TRY(d_colors.alloc((size_t)n * 3));
    HIP_TRY(hipMemcpy(d_rays.p, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
    if (stats) HIP_TRY(hipMemset(scene->d_counters, 0, RT_NUM_COUNTERS * sizeof(unsigned long long)));
    HIP_TRY(rt::launch_color_at(scene->dev, d_rays.p, n, remaining, d_colors.p, stats ? scene->d_counters : nullptr, nullptr));
    HIP_TRY(hipMemcpy(colors, d_colors.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (stats) {
        unsigned long long c[RT_NUM_COUNTERS];
        HIP_TRY(hipMemcpy(c, scene->d_counters, sizeof c, hipMemcpyDeviceToHost));
        stats_from_counters(c, stats);
    }
    return RT_OK;
}

int rt_intersect(const rt_scene *scene, const double *rays, uint32_t n, double *t, int32_t *object) {
    if (!scene || ((!rays || !t || !object) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!n) return RT_OK;
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<double> d_rays, d_t;
    DeviceBuffer<int32_t> d_obj;
    HIP_TRY(d_rays.alloc((size_t)n * 6));
    HIP_TRY(d_t.alloc(n));
    HIP_TRY(d_obj.alloc(n));
    HIP_TRY(hipMemcpy(d_rays.p, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_intersect(scene->dev, d_rays.p, n, d_t.p, d_obj.p, nullptr));
    HIP_TRY(hipMemcpy(t, d_t.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(object, d_obj.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_intersect_all(const rt_scene *scene, const double *rays, uint32_t n, uint32_t capacity, uint32_t *count,
                     double *t, int32_t *object) {
    if (!scene || ((!rays || !count) && n) || ((!t || !object) && n && capacity))
        return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!n) return RT_OK;
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    const size_t total = (size_t)n * capacity;
    DeviceBuffer<double> d_rays, d_t;
    DeviceBuffer<int32_t> d_obj;
    DeviceBuffer<uint32_t> d_count;
    HIP_TRY(d_rays.alloc((size_t)n * 6));
    HIP_TRY(d_t.alloc(total));
    HIP_TRY(d_obj.alloc(total));
    HIP_TRY(d_count.alloc(n));
    HIP_TRY(hipMemcpy(d_rays.p, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_intersect_all(scene->dev, d_rays.p, n, capacity, d_count.p, d_t.p, d_obj.p, nullptr));
    HIP_TRY(hipMemcpy(count, d_count.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (total) {
        HIP_TRY(hipMemcpy(t, d_t.p, total * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(object, d_obj.p, total * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < n; ++i)
        if (count[i] > capacity)
            return set_error(RT_ERR_CAPACITY, "ray " + std::to_string(i) + " has " + std::to_string(count[i]) +
                                                  " intersections, capacity is " + std::to_string(capacity));
    return RT_OK;
}

namespace {

static_assert(sizeof(rt_computations) == sizeof(RtComps) && offsetof(rt_computations, n1) == offsetof(RtComps, n1) &&
                  offsetof(rt_computations, point) == offsetof(RtComps, point),
              "rt_computations and RtComps must have one layout");

// The leaf a Computations record refers to, or -1 (groups have no surface: the reference panics in normal_at).
int32_t leaf_of(const rt_scene *scene, int32_t object) {
    if (object < 0 || (size_t)object >= scene->object_to_leaf.size()) return -1;
    return scene->object_to_leaf[(size_t)object];
}

// n1 and n2 of the hit (t, object) within the list xs (rendering/intersections.rs:65-103): walk the list keeping the
// objects the ray is inside of; n1 is the last container's index before the hit is applied, n2 the one after.
void refractive_indices(const rt_scene *scene, double t, int32_t object, const double *xs_t, const int32_t *xs_object,
                        uint32_t xs_n, double &n1, double &n2) {
    std::vector<int32_t> containers;
    auto last_index = [&]() {
        if (containers.empty()) return 1.0;
        const int32_t leaf = leaf_of(scene, containers.back());
        return leaf >= 0 ? scene->flat.surfaces[(size_t)leaf].refractive_index : 1.0;
    };
    n1 = 1.0;
    n2 = 1.0;
    for (uint32_t k = 0; k < xs_n; ++k) {
        const bool is_hit = xs_object[k] == object && std::fabs(xs_t[k] - t) < 1e-5;  // float_equal, core/floats.rs:4-6
        if (is_hit) n1 = last_index();
        bool found = false;
        for (size_t c = 0; c < containers.size(); ++c)
            if (containers[c] == xs_object[k]) {
                containers.erase(containers.begin() + (std::ptrdiff_t)c);
                found = true;
                break;
            }
        if (!found) containers.push_back(xs_object[k]);
        if (is_hit) {
            n2 = last_index();
            break;
        }
    }
}

int shade_part(const rt_scene *scene, int part, const rt_computations *comps, uint32_t n, int32_t remaining,
               double *colors) {
    if (!scene || ((!comps || !colors) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (remaining < 0 || remaining > RT_MAX_DEPTH)
        return set_error(RT_ERR_INVALID_ARGUMENT, "remaining must be within 0.." + std::to_string(RT_MAX_DEPTH));
    if (!n) return RT_OK;
    std::vector<RtComps> dev(n);
    std::memcpy(dev.data(), comps, (size_t)n * sizeof(RtComps));
    for (uint32_t i = 0; i < n; ++i) {
        const int32_t leaf = leaf_of(scene, comps[i].object);
        if (leaf < 0) return set_error(RT_ERR_INVALID_ARGUMENT, "Computations.object must be a leaf object of the world");
        dev[i].object = leaf;
    }
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<RtComps> d_comps;
    DeviceBuffer<double> d_colors;
    HIP_TRY(d_comps.alloc(n));
    HIP_TRY(d_colors.alloc((size_t)n * 3));
    HIP_TRY(hipMemcpy(d_comps.p, dev.data(), (size_t)n * sizeof(RtComps), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_shade_part(scene->dev, part, d_comps.p, n, remaining, d_colors.p, nullptr));
    HIP_TRY(hipMemcpy(colors, d_colors.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // namespace

int rt_prepare_computations(const rt_scene *scene, const double *rays, const double *t, const int32_t *object,
                            uint32_t n, const uint32_t *xs_begin, const double *xs_t, const int32_t *xs_object,
                            rt_computations *out) {
    if (!scene || ((!rays || !t || !object || !out) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (xs_begin && (!xs_t || !xs_object) && n && xs_begin[n] > xs_begin[0])
        return set_error(RT_ERR_INVALID_ARGUMENT, "xs_begin given without xs_t / xs_object");
    if (!n) return RT_OK;
    std::vector<RtComps> dev(n);
    for (uint32_t i = 0; i < n; ++i) {
        const int32_t leaf = leaf_of(scene, object[i]);
        if (leaf < 0)
            return set_error(RT_ERR_INVALID_ARGUMENT, "Groups do not have a normal - normal_at should be called on child objects");
        RtComps &c = dev[i];
        std::memset(&c, 0, sizeof c);
        c.time = t[i];
        c.object = leaf;
        if (xs_begin) {
            if (xs_begin[i + 1] < xs_begin[i]) return set_error(RT_ERR_INVALID_ARGUMENT, "xs_begin must not decrease");
            refractive_indices(scene, t[i], object[i], xs_t + xs_begin[i], xs_object + xs_begin[i], xs_begin[i + 1] - xs_begin[i], c.n1, c.n2);
        } else {
            refractive_indices(scene, t[i], object[i], t + i, object + i, 1, c.n1, c.n2);
        }
    }
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<double> d_rays;
    DeviceBuffer<RtComps> d_comps;
    HIP_TRY(d_rays.alloc((size_t)n * 6));
    HIP_TRY(d_comps.alloc(n));
    HIP_TRY(hipMemcpy(d_rays.p, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_comps.p, dev.data(), (size_t)n * sizeof(RtComps), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_prepare_computations(scene->dev, d_rays.p, n, d_comps.p, nullptr));
    HIP_TRY(hipMemcpy(out, d_comps.p, (size_t)n * sizeof(RtComps), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) out[i].object = object[i];
    return RT_OK;
}

int rt_shade_hit(const rt_scene *scene, const rt_computations *comps, uint32_t n, int32_t remaining, double *colors) {
    return shade_part(scene, RT_PART_SHADE_HIT, comps, n, remaining, colors);
}
int rt_reflected_color(const rt_scene *scene, const rt_computations *comps, uint32_t n, int32_t remaining,
                       double *colors) {
    return shade_part(scene, RT_PART_REFLECTED, comps, n, remaining, colors);
}
int rt_refracted_color(const rt_scene *scene, const rt_computations *comps, uint32_t n, int32_t remaining,
                       double *colors) {
    return shade_part(scene, RT_PART_REFRACTED, comps, n, remaining, colors);
}

int rt_is_shadowed(const rt_scene *scene, const double *points, uint32_t n, uint8_t *shadowed) {
    if (!scene || ((!points || !shadowed) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!n) return RT_OK;
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<double> d_points;
    DeviceBuffer<uint8_t> d_out;
    HIP_TRY(d_points.alloc((size_t)n * 3));
    HIP_TRY(d_out.alloc(n));
    HIP_TRY(hipMemcpy(d_points.p, points, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_is_shadowed(scene->dev, d_points.p, n, d_out.p, nullptr));
    HIP_TRY(hipMemcpy(shadowed, d_out.p, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_normal_at(const rt_scene *scene, const int32_t *object, const double *points, uint32_t n, double *normals) {
    if (!scene || ((!object || !points || !normals) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!n) return RT_OK;
    std::vector<int32_t> leaf(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (object[i] < 0 || (size_t)object[i] >= scene->object_to_leaf.size() || scene->object_to_leaf[(size_t)object[i]] < 0)
            return set_error(RT_ERR_INVALID_ARGUMENT, "Groups do not have a normal - normal_at should be called on child objects");
        leaf[i] = scene->object_to_leaf[(size_t)object[i]];
    }
    DeviceGuard guard(scene->device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<int32_t> d_leaf;
    DeviceBuffer<double> d_points, d_normals;
    HIP_TRY(d_leaf.alloc(n));
    HIP_TRY(d_points.alloc((size_t)n * 3));
    HIP_TRY(d_normals.alloc((size_t)n * 3));
    HIP_TRY(hipMemcpy(d_leaf.p, leaf.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_points.p, points, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_normal_at(scene->dev, d_leaf.p, d_points.p, n, d_normals.p, nullptr));
    HIP_TRY(hipMemcpy(normals, d_normals.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_ray_for_pixel(const rt_camera_desc *camera, const uint32_t *px, const uint32_t *py, uint32_t n, double *rays,
                     int device) {
    if (!camera || ((!px || !py || !rays) && n)) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    RtCameraDev cam;
    std::string err;
    int rc = rt::camera_to_device(*camera, cam, err);
    if (rc != RT_OK) return set_error(rc, err);
    if (!n) return RT_OK;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    DeviceGuard guard(device);
    if (guard.status != hipSuccess) return hip_error(guard.status, "hipSetDevice");
    DeviceBuffer<uint32_t> d_px, d_py;
    DeviceBuffer<double> d_rays;
    HIP_TRY(d_px.alloc(n));
    HIP_TRY(d_py.alloc(n));
    HIP_TRY(d_rays.alloc((size_t)n * 6));
    HIP_TRY(hipMemcpy(d_px.p, px, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_py.p, py, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_ray_for_pixel(cam, d_px.p, d_py.p, n, d_rays.p, nullptr));
    HIP_TRY(hipMemcpy(rays, d_rays.p, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_canvas_quantize_device(const double *rgb_device, size_t n, uint8_t *out_device, void *stream) {
    if ((!rgb_device || !out_device) && n) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(rt::launch_quantize(rgb_device, n, out_device, (hipStream_t)stream));
    return RT_OK;
}

int rt_camera_new(uint32_t hsize, uint32_t vsize, double field_of_view, rt_camera_desc *out) {
    if (!out) return set_error(RT_ERR_INVALID_ARGUMENT, "out must not be NULL");
    rt::camera_new(hsize, vsize, field_of_view, *out);
    return RT_OK;
}

int rt_view_transform(const double from[3], const double to[3], const double up[3], double out[16]) {
    if (!from || !to || !up || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    rt::view_transform(rt::Vec3{from[0], from[1], from[2]}, rt::Vec3{to[0], to[1], to[2]}, rt::Vec3{up[0], up[1], up[2]})
        .to_array(out);
    return RT_OK;
}

int rt_scene_file_load(const char *path, rt_scene_file **out) {
    if (!path || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "path and out must not be NULL");
    *out = nullptr;
    auto f = std::make_unique<rt_scene_file>();
    std::string err;
    int rc = rt::SceneFile::from_file(path, f->file, err);
    if (rc != RT_OK) return set_error(rc, err);
    *out = f.release();
    return RT_OK;
}

int rt_scene_file_parse(const char *text, size_t len, rt_scene_file **out) {
    if (!text || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "text and out must not be NULL");
    *out = nullptr;
    auto f = std::make_unique<rt_scene_file>();
    std::string err;
    int rc = rt::SceneFile::from_str(std::string(text, len), f->file, err);
    if (rc != RT_OK) return set_error(rc, err);
    *out = f.release();
    return RT_OK;
}

void rt_scene_file_free(rt_scene_file *file) { delete file; }

int rt_scene_file_build_camera(const rt_scene_file *file, rt_camera_desc *out) {
    if (!file || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "file and out must not be NULL");
    file->file.build_camera(*out);
    return RT_OK;
}

int rt_scene_file_build_world(rt_scene_file *file, rt_world_desc *out) {
    if (!file || !out) return set_error(RT_ERR_INVALID_ARGUMENT, "file and out must not be NULL");
    std::string err;
    int rc = file->file.build_world(file->objects, *out, err);
    if (rc != RT_OK) return set_error(rc, err);
    return RT_OK;
}

}  // extern "C"
