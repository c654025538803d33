/*
 * raytracer_rs_amd.h -- C ABI of the MI355X (gfx950) renderer for raytracer-rs's per-pixel render path.
 *
 * The reference crate has no FFI of its own: the boundary it offers is the Rust API
 * `SceneFile -> (Camera, World) -> Camera::render(&World) -> Canvas`.  These entry points are what a `-sys`
 * binding under that API calls; each one names the reference item it stands in for (file:line under the
 * reference checkout, `src/` omitted).
 *
 * Conventions
 *   - plain C types only; every function returns RT_OK (0) or a negative RT_ERR_* code and never aborts or
 *     throws across the boundary; rt_last_error() returns the message of the calling thread's last failure.
 *   - matrices are 16 doubles, row-major, exactly `Matrix4.data` (core/matrices.rs:7-9).
 *   - colours/framebuffers are 3 doubles per pixel, row-major, index (y*width + x)*3: the layout of
 *     `Canvas.pixels` (rendering/canvas.rs:7-11,40-42).
 *   - objects are listed in depth-first pre-order: a group (rendering/objects.rs:34-41, geometry/groups.rs:15-21)
 *     is followed by its `n_children` children, each of which may itself be a group.
 *   - all arithmetic on the path is IEEE f64 without multiply-add contraction, in the reference's operation order.
 *   - the caller owns every output buffer; handles are opaque; a handle is bound to the HIP device it was created
 *     on and serves ONE CALL AT A TIME: a scene handle owns device scratch (tile order, counters, timing events, the
 *     staging pipeline of rt_render) that every render through it rewrites, whatever the `const` in the signatures
 *     suggests.  Calls from several host threads on one handle, and asynchronous launches through one handle that
 *     may overlap on the device (different streams), must be serialised by the caller; different handles -- also of
 *     the same world on the same device -- are independent.
 *   - there is no CPU fallback: without a HIP device every rendering entry point fails with RT_ERR_NO_DEVICE.
 */
#ifndef RAYTRACER_RS_AMD_H
#define RAYTRACER_RS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 4

enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NOT_INVERTIBLE = -2, /* the reference panics: "... should be invertible" (objects.rs:88-101,123-125; world.rs:54-57; camera.rs:46-49; patterns.rs:140-143) */
    RT_ERR_MATERIAL = -3,       /* the reference panics: "Material values must be positive" (scene/materials.rs:94-96) */
    RT_ERR_HIP = -4,
    RT_ERR_IO = -5,             /* SceneFile::from_file Err: fs::read_to_string (scenes/loader.rs:191) */
    RT_ERR_PARSE = -6,          /* SceneFile::from_file Err: toml::from_str (scenes/loader.rs:192) */
    RT_ERR_UNSUPPORTED = -7,    /* e.g. groups nested deeper than RT_MAX_GROUP_NESTING */
    RT_ERR_NO_DEVICE = -8,
    RT_ERR_CAPACITY = -9        /* an output list was too short; the counts say how long it has to be */
};

/* enum Object (rendering/objects.rs:34-41) */
enum { RT_SPHERE = 0, RT_PLANE = 1, RT_CUBE = 2, RT_CYLINDER = 3, RT_CONE = 4, RT_GROUP = 5 };
/* enum PatternKind (scene/patterns.rs:13-32); RT_PATTERN_NONE is Option::None (scene/materials.rs:18) */
enum { RT_PATTERN_NONE = -1, RT_PATTERN_STRIPE = 0, RT_PATTERN_GRADIENT = 1, RT_PATTERN_RING = 2, RT_PATTERN_CHECKERS = 3 };

/* Groups may be nested this deep (a leaf inside a top-level group is at depth 1).  The reference recurses without a
 * limit (rendering/world.rs:41-78); the device walks any depth with one loop and parks one ray per level, this many. */
#define RT_MAX_GROUP_NESTING 16
/* DEFAULT_MAX_REFLECTION_DEPTH == DEFAULT_MAX_REFRACTION_DEPTH (rendering/world.rs:11-12) */
#define RT_DEFAULT_DEPTH 5
#define RT_MAX_DEPTH 8

typedef struct rt_pattern_desc { /* struct Pattern (scene/patterns.rs:7-10) */
    int32_t kind;                /* RT_PATTERN_* */
    int32_t _pad;
    double transform[16];
    double color_a[3];
    double color_b[3];
} rt_pattern_desc;

typedef struct rt_material_desc { /* struct Material (scene/materials.rs:9-19) */
    double color[3];
    double ambient, diffuse, specular, shininess, reflectivity, transparency, refractive_index;
    rt_pattern_desc pattern;
} rt_material_desc;

typedef struct rt_object_desc { /* one `Object` (rendering/objects.rs:34-41) */
    int32_t kind;               /* RT_SPHERE .. RT_GROUP */
    int32_t n_children;         /* groups only: Group.children.len() (geometry/groups.rs:20) */
    int32_t closed;             /* Cylinder/Cone.closed (geometry/cylinders.rs:21, cones.rs:21) */
    int32_t _pad;
    double minimum, maximum;    /* Cylinder/Cone.minimum/maximum; ignored for other kinds */
    double transformation[16];       /* Transformable::transformation() (objects.rs:212-221) */
    double world_transformation[16]; /* Object::world_transformation() (objects.rs:241-250) */
    rt_material_desc material;       /* HasMaterial::material() (objects.rs:188-197) */
} rt_object_desc;

typedef struct rt_world_desc { /* struct World (rendering/world.rs:14-17) */
    const rt_object_desc *objects; /* all objects, depth-first pre-order */
    uint32_t n_objects;
    int32_t has_light;             /* light_source.is_some() */
    double light_position[3];      /* PointLight.position (scene/lights.rs:5-8) */
    double light_intensity[3];     /* PointLight.intensity */
} rt_world_desc;

typedef struct rt_camera_desc { /* struct Camera (rendering/camera.rs:8-16) */
    uint32_t hsize, vsize;
    double field_of_view;
    double transform[16];
    double pixel_size, half_width, half_height;
} rt_camera_desc;

/* Operation counts of one call.  `rays` counts World::intersect invocations (rendering/world.rs:27): camera rays,
 * shadow rays, reflection rays and refraction rays. */
typedef struct rt_stats {
    uint64_t rays, primary, shadow, reflect, refract;
    uint64_t prim_tests[5]; /* Shape::local_intersect calls per leaf kind */
    uint64_t prim_hits[5];  /* ... that returned at least one t */
    uint64_t aabb_tests;    /* Bounds::intersects calls */
    uint64_t ray_transforms;/* Ray::transform calls */
    uint64_t shades;        /* shade_hit calls */
    uint64_t lit;           /* lighting() calls that took the diffuse branch */
    uint64_t specular;      /* ... that evaluated powf */
    uint64_t patterns;      /* Pattern::color_at_object calls */
    uint64_t refract_setups;/* refracted_color calls past the two early-outs */
    uint64_t schlick;       /* Computations::schlick calls */
    double kernel_ms;       /* device time of the kernel(s), by HIP events (0 for the asynchronous entry point);
                               rt_multi_*: wall-clock time from the start of the call until the last band's kernel had
                               finished on every GPU */
    double gather_ms;       /* rt_multi_*: wall-clock time from then until the frame was complete where the caller wants it
                               (the part of the collection that rendering did not hide); 0 elsewhere */
} rt_stats;

typedef struct rt_render_opts {
    uint32_t row_begin, row_end; /* rows [row_begin, row_end) of the frame; both 0 = the whole frame.  A pixel comes
                                    out bit-identical to the whole frame's when the range lies on the 8-row grid the
                                    kernels tile the frame with (row_begin, and row_end unless it is the frame's last
                                    row, multiples of 8; likewise band_rows below); other ranges may differ from it in
                                    the last bits of a colour in worlds with groups or long object lists, where the
                                    lanes of a tile share rays and a pixel's terms are then summed in another order */
    int32_t max_depth;           /* recursion budget per pixel (world.rs:117 `remaining`).  0 = RT_DEFAULT_DEPTH, so that
                                    zeroed options mean the reference's behaviour; a budget of zero itself (no
                                    reflection or refraction rays at all) is max_depth = 0 with RT_RENDER_EXACT_DEPTH */
    int32_t collect_stats;       /* non-zero: run the counting variant of the kernel and fill rt_stats */
    void *stream;                /* hipStream_t for rt_render_device; NULL = the null stream */
    /* Row-band sharding (one frame over several GPUs): when band_rows > 0 the frame is cut into bands of band_rows
     * rows and this call renders bands band_first, band_first + band_step, band_first + 2*band_step, ...
     * (row_begin/row_end are ignored).  band_rows = 0 renders the contiguous range above. */
    uint32_t band_rows, band_first, band_step;
    uint32_t flags; /* RT_RENDER_* */
} rt_render_opts;
/* rt_render_device: record events round the launch so that rt_scene_render_times can report it (no synchronisation). */
#define RT_RENDER_TIMED 1u
/* rt_render_device: store the frame as 8-bit values -- `rgb_device` then points at uint8_t[rows * hsize * 3] -- with the
 * canvas quantiser (rendering/canvas.rs:123-130) applied by the render kernel itself; the same bytes as rendering
 * doubles and running rt_canvas_quantize_device over them, without the f64 frame ever being written. */
#define RT_RENDER_QUANTIZED 2u
/* rt_render, rt_render_device: max_depth is the budget exactly as given, 0 included (without this flag 0 selects
 * RT_DEFAULT_DEPTH). */
#define RT_RENDER_EXACT_DEPTH 4u

/* rt_render_device: `rgb_device` is the WHOLE frame (vsize*hsize*3 values) and every row rendered is stored at its own
 * place in it, instead of packed from the start of the buffer: a row range or a selection of bands then fills in its
 * part of a frame that other calls -- other GPUs -- complete. */
#define RT_RENDER_IN_PLACE 8u

typedef struct rt_scene_info {
    uint32_t n_objects, n_leaves, n_groups, max_nesting;
    uint64_t device_bytes;
    int32_t device;
    int32_t kernel_build; /* which build of the render kernel serves this world: 0 = a table of leaves only (no groups, at
                             most 12 objects), 1 = any table, 2 = a table in which no range of nodes holds a cone,
                             3 = a world with groups, rendered level by level of its ray trees through queues in device
                             memory (rt_render_device_noalloc serves such a world only for views rendered before);
                             1 and 2 serve worlds with groups only when RT_RENDER_LEVELS=0 was in the environment
                             when the scene was created; 4 = a table of leaves only in which nothing reflects or
                             transmits: a kernel of one round (primary ray, shading, shadow ray) */
} rt_scene_info;

/* struct Computations (rendering/intersections.rs:12-24); `object` is an index into rt_world_desc.objects. */
typedef struct rt_computations {
    double time;
    int32_t object;
    int32_t inside;
    double point[3], eye_vector[3], normal_vector[3], reflect_vector[3], over_point[3], under_point[3];
    double n1, n2;
} rt_computations;

typedef struct rt_scene rt_scene;           /* a World resident on one GPU */
typedef struct rt_scene_file rt_scene_file; /* struct SceneFile (scenes/loader.rs:21-26) */

const char *rt_last_error(void);
int rt_abi_version(void);
/* Number of HIP devices visible to the process. */
int rt_device_count(int *count);

/* Copies a World into device memory: pre-inverts every transformation with the reference's cofactor algorithm
 * (core/matrices.rs:94-162), computes every group's bounds (geometry/groups.rs:130-149, geometry/bounds.rs:36-89)
 * and uploads the flattened tables.  Stands in for the `&World` borrow of Camera::render (rendering/camera.rs:56).
 * device < 0 selects the process's current HIP device. */
int rt_scene_create(const rt_world_desc *world, int device, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);

/* Camera::render (rendering/camera.rs:56-67): writes rows [row_begin,row_end) of the frame into the host buffer
 * `rgb` (which holds the WHOLE frame: hsize*vsize*3 doubles; rows outside the range are left untouched).  The handle
 * keeps a device frame and pinned staging memory of the size of the largest frame rendered through it, so that
 * kernels, the transfer over PCIe and the copy into `rgb` overlap. */
int rt_render(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb,
              rt_stats *stats);
/* The same on device memory, asynchronously on opts->stream: `rgb_device` receives only the rows rendered, packed:
 * (row_end-row_begin)*hsize*3 doubles starting at row_begin, or with bands the selected bands one after another
 * (band_rows*hsize*3 doubles each, the frame's last band possibly shorter).  No host synchronisation unless
 * collect_stats is set -- or the world has groups (rt_scene_info.kernel_build 3): such a world is rendered level by
 * level of its ray trees through ray queues in device memory; the FIRST frame of a view (camera, rows, budget) waits on
 * opts->stream once per level to learn how many rays the next level has, sorts every level's rays, and lets the handle
 * allocate (and keep) device memory that grows with the frame, about 0.5 KB per pixel; the handle remembers the counts
 * of 32 views, whose later frames are launched without waiting, like any world's, and for the four rendered last also
 * the order of their rays and the cost of their waves, so that their later frames run the level kernels alone, most expensive
 * waves first (only scheduling is remembered, never a result).  The frame is complete in stream order when the call
 * returns, as for any world.
 * A handle serves ONE render at a time (as `&World` is borrowed for the duration of Camera::render): the scratch in
 * which a launch orders its tiles, the counters and the timing marks belong to the handle, which is `const` here only
 * in the sense that the world it holds does not change.  Launches of one handle on one stream are ordered by the
 * stream; launches that may overlap in time (two streams, two host threads) need a handle each -- rt_scene_create is
 * cheap, the world is a few kilobytes to a few megabytes. */
int rt_render_device(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts,
                     double *rgb_device, rt_stats *stats);

/* rt_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  No counterpart in the reference.  Same arguments and the same pixels as rt_render_device,
 * but the call makes no hipMalloc / hipFree / event or stream creation: the tile-ordering scratch its kernels use (the
 * one buffer rt_render_device allocates on demand) lives in a static device array of RT_NOALLOC_SCRATCH_BYTES that
 * every GPU holds from the library's first use on it until the process ends; its address is looked up when a scene is
 * created on that GPU.  Limits: the rows rendered by one call may cover at most RT_NOALLOC_MAX_TILES 8x8-pixel tiles
 * (ceil(hsize/8) * ceil(rows/8); 7680x4320 is 518 400) -- more fails with RT_ERR_CAPACITY; collect_stats and
 * RT_RENDER_TIMED are refused (RT_ERR_INVALID_ARGUMENT: they need a host synchronisation resp. event creation).
 * All calls on one GPU -- through any scene handle -- share that GPU's array: calls that overlap in time on one GPU
 * race on the tile order (the caller serialises them, e.g. by using one stream); the array keeps the last call's
 * contents and every call rewrites what it reads.
 * A world rendered level by level (rt_scene_info.kernel_build 3) does not use the static array: its frames go through
 * ray queues that the handle owns.  For such a world the call is served -- again without allocating or waiting -- for
 * a view (camera, rows, depth) that rt_render_device has rendered on this handle before, and refused with
 * RT_ERR_UNSUPPORTED otherwise (a view's first frame sizes the queues and learns its ray counts from the GPU). */
#define RT_NOALLOC_MAX_TILES 524288u
#define RT_NOALLOC_SCRATCH_BYTES ((256u + 3u * RT_NOALLOC_MAX_TILES) * 4u)
int rt_render_device_noalloc(const rt_scene *scene, const rt_camera_desc *camera, const rt_render_opts *opts,
                             double *rgb_device);

/* Measurement aid, no counterpart in the reference: GPU time of a launch made with RT_RENDER_TIMED, `back` launches
 * ago (0 = the latest; the handle remembers 32): the tile-ordering pass and the render kernel, in milliseconds.
 * The caller must have synchronised the launch's stream first (RT_ERR_HIP otherwise). */
int rt_scene_render_times(const rt_scene *scene, uint32_t back, double *ordering_ms, double *render_ms);

/* Camera::render's row loop (rendering/camera.rs:58-64) over several GPUs of one node, driven from one process.  The
 * reference renders its rows one after the other on one core; here the frame is cut into stripes of rows, unit u of U
 * is the set of stripes u, u + U, u + 2U, ... (every unit holds the same mix of cheap and expensive rows), and every GPU
 * takes the next unit from a shared cursor whenever it has fewer than two in flight, so that the GPUs finish together
 * whatever their speed.  `devices` lists HIP device ordinals, one worker (host thread, two streams, copies of the
 * world) each; an ordinal may appear more than once (several workers on one GPU).  The frame is assembled on
 * devices[0].  The handle keeps its worker threads until it is destroyed; it serves one call at a time. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_world_desc *world, const int *devices, uint32_t n_devices, rt_multi **out);
void rt_multi_destroy(rt_multi *multi);
/* How units rendered on other GPUs reach the frame on devices[0] (rt_multi_render_device); chosen when the handle is
 * created.  Such a GPU renders a unit's stripes packed, the packed unit travels as one message into staging memory on
 * devices[0], and a small kernel there puts the stripes in place.  RT_GATHER_RCCL: a worker that has issued its last
 * unit sends its units in one group of ncclSend/ncclRecv pairs over its own xGMI link while other GPUs are still
 * rendering (a copy of RCCL already in the process is used, else librccl.so.1 is loaded).  RT_GATHER_COPY: every unit is
 * copied behind its kernel by the GPU that rendered it (hipMemcpyPeerAsync) -- taken when RCCL cannot be loaded or
 * initialised, or when the environment says RT_MULTI_GATHER=copy (RT_MULTI_GATHER=rccl makes a missing RCCL an error
 * instead).  RT_GATHER_NONE: every worker is on devices[0] and renders straight into the frame. */
enum { RT_GATHER_NONE = 0, RT_GATHER_RCCL = 1, RT_GATHER_COPY = 2 };
typedef struct rt_multi_info {
    uint32_t n_workers, n_devices; /* workers; distinct GPUs among them */
    int32_t gather;                /* RT_GATHER_* */
    uint32_t rccl_ranks;           /* size of the RCCL communicator, 0 without one */
    uint32_t band_rows, n_units;   /* of the last call: height of a stripe, number of units */
    char note[160];                /* the RCCL library in use, or why RCCL is not */
} rt_multi_info;
int rt_multi_get_info(const rt_multi *multi, rt_multi_info *out);
/* The whole frame into device memory of devices[0] (vsize*hsize*3 doubles, or bytes with RT_RENDER_QUANTIZED), the
 * call returning when it is complete.  Of opts: max_depth, collect_stats, flags (RT_RENDER_QUANTIZED,
 * RT_RENDER_EXACT_DEPTH), band_rows = the height of a stripe (a multiple of 8; 0: 16 rows) and band_step = the number
 * of units (0: two per worker, fewer for small frames; one worker alone renders the frame in one launch);
 * everything else must be zero.  stats->kernel_ms and stats->gather_ms are wall-clock times (see rt_stats). */
int rt_multi_render_device(rt_multi *multi, const rt_camera_desc *camera, const rt_render_opts *opts, void *frame_device,
                           rt_stats *stats);
/* The same into host memory (Canvas.pixels: vsize*hsize*3 doubles): every GPU sends its units to the host itself, over
 * its own PCIe link, and nothing travels between GPUs. */
int rt_multi_render(rt_multi *multi, const rt_camera_desc *camera, const rt_render_opts *opts, double *rgb, rt_stats *stats);
/* Measurement aids: how many units each worker took in the last call, and when each worker's last kernel had finished
 * (milliseconds since the start of that call); n = number of workers. */
int rt_multi_units_taken(const rt_multi *multi, uint32_t *per_worker, uint32_t n);
int rt_multi_worker_ms(const rt_multi *multi, double *per_worker, uint32_t n);

/* World::color_at (rendering/world.rs:113-126) for n rays (6 doubles each: origin xyz, direction xyz) with the
 * given recursion budget (world.rs:117 `remaining`; pass RT_DEFAULT_DEPTH for color_at itself). */
int rt_color_at(const rt_scene *scene, const double *rays, uint32_t n, int32_t remaining, double *colors,
                rt_stats *stats);
/* World::intersect + Intersection::hit (rendering/world.rs:27-36, rendering/intersections.rs:37-39) for n rays:
 * t[i] and the index (into rt_world_desc.objects) of the hit object, or -1 / NaN when nothing is hit. */
int rt_intersect(const rt_scene *scene, const double *rays, uint32_t n, double *t, int32_t *object);
/* World::intersect itself (rendering/world.rs:27-36) for n rays: the whole list of intersections of ray i, stably
 * sorted by t (negative t's included), in t[i*capacity ..] / object[i*capacity ..]; count[i] entries.  When some
 * ray has more than `capacity` intersections the call returns RT_ERR_CAPACITY with every count[i] set to the
 * length that ray needs (lists of rays within capacity are still complete).  A list that contains a NaN t comes
 * back in an unspecified order (the reference's comparator calls NaN equal to everything, world.rs:34). */
int rt_intersect_all(const rt_scene *scene, const double *rays, uint32_t n, uint32_t capacity, uint32_t *count,
                     double *t, int32_t *object);
/* Intersection::prepare_computations_for_intersections (rendering/intersections.rs:45-118) for n hits: hit i is
 * (t[i], object[i]) on rays[i] and its intersection list is xs_t / xs_object [xs_begin[i], xs_begin[i+1])
 * (xs_begin has n+1 entries).  xs_begin = NULL gives every hit the one-element list [hit], which is
 * Intersection::prepare_computations (intersections.rs:41-43), the form World::color_at uses (world.rs:121). */
int rt_prepare_computations(const rt_scene *scene, const double *rays, const double *t, const int32_t *object,
                            uint32_t n, const uint32_t *xs_begin, const double *xs_t, const int32_t *xs_object,
                            rt_computations *out);
/* World::shade_hit, World::reflected_color, World::refracted_color (rendering/world.rs:80-111, 151-168, 170-198)
 * for n Computations, with the recursion budget `remaining` of their _internal forms (RT_DEFAULT_DEPTH for the
 * public ones). */
int rt_shade_hit(const rt_scene *scene, const rt_computations *comps, uint32_t n, int32_t remaining, double *colors);
int rt_reflected_color(const rt_scene *scene, const rt_computations *comps, uint32_t n, int32_t remaining,
                       double *colors);
int rt_refracted_color(const rt_scene *scene, const rt_computations *comps, uint32_t n, int32_t remaining,
                       double *colors);
/* World::is_shadowed (rendering/world.rs:128-149) for n points (3 doubles each). */
int rt_is_shadowed(const rt_scene *scene, const double *points, uint32_t n, uint8_t *shadowed);
/* Object::normal_at (rendering/objects.rs:171-184) of object[i] at points[i]. */
int rt_normal_at(const rt_scene *scene, const int32_t *object, const double *points, uint32_t n, double *normals);
/* Camera::ray_for_pixel (rendering/camera.rs:39-54) for n pixels, evaluated by the device code path. */
int rt_ray_for_pixel(const rt_camera_desc *camera, const uint32_t *px, const uint32_t *py, uint32_t n, double *rays,
                     int device);

/* Camera::new (rendering/camera.rs:19-37). */
int rt_camera_new(uint32_t hsize, uint32_t vsize, double field_of_view, rt_camera_desc *out);
/* view_transform (scene/transformations.rs:4-17). */
int rt_view_transform(const double from[3], const double to[3], const double up[3], double out[16]);

/* Canvas::to_ppm_string (rendering/canvas.rs:44-90): P3 text into `out` (capacity cap); returns the length
 * needed, so a first call with out = NULL sizes the buffer. */
size_t rt_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap);
/* Canvas::to_ppm (rendering/canvas.rs:116-120). */
int rt_canvas_write_ppm(const double *rgb, uint32_t width, uint32_t height, const char *path);
/* Canvas::to_rgb_image + the encoder behind Canvas::to_jpeg (rendering/canvas.rs:92-114: `image` 0.24.6,
 * ImageOutputFormat::Jpeg(100)): the bytes of the JPEG file into `out` (capacity cap); returns the length needed (0
 * for an empty frame or one beyond 65535 pixels a side), so a first call with out = NULL sizes the buffer.  Baseline,
 * quality 100, no chroma subsampling; byte-identical to the files the reference ships beside its PPMs. */
size_t rt_canvas_to_jpeg(const double *rgb, uint32_t width, uint32_t height, unsigned char *out, size_t cap);
/* Canvas::to_jpeg (rendering/canvas.rs:109-114). */
int rt_canvas_write_jpeg(const double *rgb, uint32_t width, uint32_t height, const char *path);
/* convert_canvas_color_value_to_decimal_rgb_value (rendering/canvas.rs:123-130) over n values. */
int rt_canvas_quantize(const double *rgb, size_t n, uint8_t *out);
/* The same over a frame still in device memory (both pointers are device pointers on the current device),
 * asynchronously on `stream` (a hipStream_t; NULL = the null stream): 8-bit frames are what a PPM needs and an
 * eighth of the bytes to move between GPUs or to the host. */
int rt_canvas_quantize_device(const double *rgb_device, size_t n, uint8_t *out_device, void *stream);

/* SceneFile::from_file (scenes/loader.rs:190-194): reads and parses a TOML scene. */
int rt_scene_file_load(const char *path, rt_scene_file **out);
/* The same from memory (the text of a scene file). */
int rt_scene_file_parse(const char *text, size_t len, rt_scene_file **out);
void rt_scene_file_free(rt_scene_file *file);
/* SceneFile::build_camera (scenes/loader.rs:196-212). */
int rt_scene_file_build_camera(const rt_scene_file *file, rt_camera_desc *out);
/* SceneFile::build_world (scenes/loader.rs:214-247).  The object array belongs to the file handle and stays
 * valid until rt_scene_file_free. */
int rt_scene_file_build_world(rt_scene_file *file, rt_world_desc *out);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_RS_AMD_H */
