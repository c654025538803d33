// Launchers of the gfx950 kernels (implemented in rt_device.hip).  Host code calls these; all pointers are device
// pointers on the current HIP device, all launches are asynchronous on `stream`.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/raytracer_rs_amd.h"
#include "rt_device_types.h"

namespace rt {

struct DeviceScene {
    const RtNode *nodes = nullptr;
    const RtSurface *surfaces = nullptr;
    int32_t n_nodes = 0;
    int32_t n_leaves = 0;
    int32_t max_nesting = 0;  // deepest nesting level of any node (0: no groups)
    int32_t any_order = 0;    // non-zero: some sibling list of the table is not in the reference's order
    int32_t cone_nodes = 0;   // non-zero: some range-skip node has cones below it (RT_SKIP_CONES)
    int32_t no_secondary = 0; // non-zero: no surface of the world reflects or transmits: a pixel is its primary ray and one shadow ray
    uint32_t table_records = 0;  // records per node table, the inert trailing one included (n_nodes + 1)
    unsigned long long oct_tables = 0;  // byte k: which of the node tables (one after the other at `nodes`) suits rays of
                                        // direction octant k (bit a of k set: travelling towards -a); all zero: one table
    RtLight light{};
};

// Which rows of the frame one launch renders, and where they land in the packed output buffer: packed row r
// (0 <= r < n_rows) is frame row  row_begin + (r / band_rows) * band_pitch + (r % band_rows), rendered only when
// that is < row_end.  A contiguous range is band_rows = n_rows, band_pitch = 0.
struct RowMap {
    uint32_t row_begin = 0, row_end = 0;
    uint32_t n_rows = 0;
    uint32_t band_rows = 1, band_pitch = 0;
    uint32_t n_tiles = 0;                 // set by launch_render
    const uint32_t *tile_list = nullptr;  // set by launch_render
    uint8_t *quantized = nullptr;         // when set, the frame is stored here as 8-bit values (3 bytes per pixel, the
                                          // canvas quantiser applied in the kernel) and `rgb` is not written
    uint32_t in_place = 0;                // non-zero: the output buffer is the WHOLE frame and a row lands at its own place
                                          // in it (index = frame row), not packed after the rows rendered before it
};

// Device memory of the level-by-level renderer (launch_render_levels), owned by whoever owns the scene (one arena per
// scene handle; one frame at a time).  Buffers grow on demand and are kept between frames.
struct ArenaBuffer {
    void *p = nullptr;
    size_t bytes = 0;
};
constexpr int LEVELS_MAX = RT_MAX_DEPTH + 1;  // level 0 = the rays through the pixels, level k = their k-th generation of children
struct LevelArena {
    ArenaBuffer queue_f[2];                // the rays a level reads and the rays it writes: 64 bytes each (origin, direction, weight, remaining budget)
    ArenaBuffer counts, offsets;           // children per kind and wave of the level being rendered, and the same scanned
    ArenaBuffer terms[LEVELS_MAX];         // per ray of a level: its term of its pixel's sum, then the sum over its subtree
    ArenaBuffer links[LEVELS_MAX];         // per ray of a level: which of its wave's children are its own
    ArenaBuffer scan_temp;
    ArenaBuffer key_of_slot, sort_buf[3];   // sorting a level's rays: key per queue slot; keys in, keys out, slots out
    uint32_t *host_totals = nullptr;        // pinned: the level's child totals as the host reads them
    uint32_t levels = 0;                    // of the last frame: levels rendered, rays per level
    uint32_t level_rays[LEVELS_MAX] = {};
    // How many rays each level of a frame has, remembered per view (camera, rows, budget): the scene behind a handle
    // never changes, so the same view has the same ray trees, and a frame whose counts are known is launched without
    // waiting for the GPU between levels.  A few entries, least recently used first to go.
    struct Known {
        RtCameraDev cam;
        uint32_t rows[5];
        int32_t max_depth;
        uint32_t levels;
        uint32_t level_rays[LEVELS_MAX];
        uint64_t used;
    };
    static constexpr int KNOWN_MAX = 32;
    Known known[KNOWN_MAX] = {};
    uint32_t n_known = 0;
    uint64_t clock = 0;
    // Set by a caller that must not allocate or wait (rt_render_device_noalloc) for the duration of its call: a frame
    // whose view is not remembered, or whose buffers would have to grow, is then refused (hipErrorNotReady) instead.
    bool frozen = false;
    // Which remembered view the numbering in index[] / ray_of_slot[] belongs to (-1: none).  The numbering of a view's
    // rays is the same in every frame, so a frame of the view rendered last skips the counting, the scans and the
    // numbering kernels altogether.
    ArenaBuffer wave_cost;                  // shader clocks per wave of the level just rendered
    // The numbering of a view's rays: which queue slot holds which ray of every level, and the launch order of every
    // level's waves.  It is the same in every frame of the view, so a frame of a view whose numbering is still in place
    // skips the counting, the scans, the sorts and the numbering kernels altogether.  A few are kept (a frame rendered
    // in row groups is several views), least recently used first to go; every buffer grows to the largest size ever
    // asked of its kind, so that after the largest view has been rendered nothing grows again.
    struct Numbering {
        ArenaBuffer index[LEVELS_MAX];        // per level: ray number in the next level -> its slot in the queue
        ArenaBuffer ray_of_slot[LEVELS_MAX];  // the other way round (also for the sums: where a ray's children are)
        ArenaBuffer wave_order[LEVELS_MAX];   // per level: the waves in launch order
        int32_t view = -1;                    // which remembered view it belongs to (index into known[]; -1: none)
        bool sorted = true;                   // the sorted numbering (RT_LEVELS_SORT)
        bool ordered = false;                 // wave_order[] is filled (RT_LEVELS_ORDER)
        uint64_t used = 0;
    };
    static constexpr int NUMBERINGS_MAX = 4;
    Numbering numberings[NUMBERINGS_MAX];
    size_t numbering_cap[3][LEVELS_MAX] = {};  // largest index / ray_of_slot / wave_order ever asked for, per level
    bool capture_seen = false;              // a call on this handle was captured into a HIP graph: no numbering is ever reused again
    size_t device_bytes() const;
};
void level_arena_release(LevelArena &arena);

// Number of 32-bit words of device scratch launch_render needs for a frame of hsize x n_rows pixels.
size_t render_scratch_words(uint32_t hsize, uint32_t n_rows);
// Camera::render over the rows of `rows`: rgb receives n_rows*hsize*3 doubles.
// counters == nullptr launches the plain kernel; otherwise the counting variant adds into counters[RT_NUM_COUNTERS].
// marks (three events, or nullptr) are recorded before the tile-ordering pass, between it and the render kernel, and
// after the render kernel.
// tile_scratch (render_scratch_words() words, or nullptr) lets the launch order its 8x8 pixel tiles by expected cost,
// expensive first, which shortens the under-occupied end of the launch; the image does not depend on it.
// order_cached: tile_scratch still holds the order a previous launch computed for the same scene, camera and rows;
// the three ordering kernels are not launched again.
hipError_t launch_render(const DeviceScene &scene, const RtCameraDev &camera, const RowMap &rows, int max_depth,
                         double *rgb, unsigned long long *counters, uint32_t *tile_scratch, hipStream_t stream,
                         hipEvent_t *marks = nullptr, bool order_cached = false);
// The same frame rendered level by level: one launch for the rays through the pixels, then one per generation of
// reflection and refraction rays, each wave taking 64 rays of one kind from neighbouring pixels off a queue in device
// memory, and a pass that adds every ray's term to its parent's, children before parents.  For scenes with groups
// only (the kernels walk node tables with groups).  Blocks the calling thread between levels (it reads how many rays
// the next level has) and may allocate device memory (the arena grows to the frame's needs).  marks as above; the
// tile-ordering pass does not exist here, so marks[0] and marks[1] are recorded together.
hipError_t launch_render_levels(const DeviceScene &scene, const RtCameraDev &camera, const RowMap &rows, int max_depth,
                                double *rgb, LevelArena &arena, hipStream_t stream, hipEvent_t *marks = nullptr);
// The tile-ordering scratch that lives in a static device array (one copy per GPU) instead of allocated memory:
// static_render_scratch_resolve(device) looks its address up once per device -- call it with `device` current, at
// set-up; static_render_scratch(device) returns that address afterwards (nullptr when it was never resolved).  The
// array holds render_scratch_words() words for frames of up to RT_NOALLOC_MAX_TILES tiles.
hipError_t static_render_scratch_resolve(int device);
uint32_t *static_render_scratch(int device);
// Stripes of rows that lie packed one after the other at `src` go to their places in a frame: stripe k (stripe_bytes
// long, the last one possibly shorter; total_bytes in all) to dst + k * dst_pitch_bytes.  Both on the current device.
hipError_t launch_place_stripes(const void *src, void *dst, size_t stripe_bytes, size_t dst_pitch_bytes, size_t total_bytes,
                                hipStream_t stream);
// The canvas quantiser (rendering/canvas.rs:123-130) over n values in device memory.
hipError_t launch_quantize(const double *rgb, size_t n, uint8_t *out, hipStream_t stream);
// World::color_at for n rays.
hipError_t launch_color_at(const DeviceScene &scene, const double *rays, uint32_t n, int remaining, double *colors,
                           unsigned long long *counters, hipStream_t stream);
// World::intersect + Intersection::hit for n rays.
hipError_t launch_intersect(const DeviceScene &scene, const double *rays, uint32_t n, double *t, int32_t *object,
                            hipStream_t stream);
// World::intersect for n rays: all intersections, sorted; lane i fills slice i of t / object (cap entries each) and
// count[i] (which may exceed cap: then the slice is incomplete).
hipError_t launch_intersect_all(const DeviceScene &scene, const double *rays, uint32_t n, uint32_t cap, uint32_t *count,
                                double *t, int32_t *object, hipStream_t stream);
// Fills the vectors of n Computations whose time, object (a leaf index), n1 and n2 are set.
hipError_t launch_prepare_computations(const DeviceScene &scene, const double *rays, uint32_t n, RtComps *comps,
                                       hipStream_t stream);
// World::shade_hit / reflected_color / refracted_color (RT_PART_*) of n Computations (object = leaf index).
hipError_t launch_shade_part(const DeviceScene &scene, int part, const RtComps *comps, uint32_t n, int remaining,
                             double *colors, hipStream_t stream);
// World::is_shadowed for n points.
hipError_t launch_is_shadowed(const DeviceScene &scene, const double *points, uint32_t n, uint8_t *out,
                              hipStream_t stream);
// Object::normal_at for n (leaf, point) pairs; leaf indexes RtSurface[].
hipError_t launch_normal_at(const DeviceScene &scene, const int32_t *leaf, const double *points, uint32_t n,
                            double *normals, hipStream_t stream);
// Camera::ray_for_pixel for n pixels.
hipError_t launch_ray_for_pixel(const RtCameraDev &camera, const uint32_t *px, const uint32_t *py, uint32_t n,
                                double *rays, hipStream_t stream);

}  // namespace rt
