/*
 * rt_oracle.h -- CPU restatement of the raytracer-rs per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker for the HIP
 * renderer: it may be loaded by tests/, by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py, and by nothing else.  The product library
 * (raytracer_rs_amd/librt_amd.so) never links, loads or calls it.
 *
 * Every function follows one function of the reference (Rust, f64, no FMA
 * contraction) operation by operation; the reference file:line is cited at
 * each definition in rt_oracle.c.  Parity is pinned by the reference's own
 * golden images (outputs/chapter_{5,6,10,11,13}*.ppm, bit-exact at 8 bits)
 * and by the known answers of its inline unit tests (tests/test_oracle_*.py).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_EPSILON 1e-5 /* core/floats.rs:2 */

enum { ORC_SPHERE = 0, ORC_PLANE = 1, ORC_CUBE = 2, ORC_CYLINDER = 3, ORC_CONE = 4, ORC_GROUP = 5 };
enum { ORC_PAT_NONE = -1, ORC_PAT_STRIPE = 0, ORC_PAT_GRADIENT = 1, ORC_PAT_RING = 2, ORC_PAT_CHECKERS = 3,
       ORC_PAT_TEST = 4 };

typedef struct { double m[4][4]; } orc_mat4;       /* core/matrices.rs:7 (row-major) */
typedef struct { double x, y, z; } orc_vec3;       /* Point (w=1) or Vector (w=0): core/tuples.rs:14,136 */
typedef struct { orc_vec3 origin, direction; } orc_ray; /* rendering/rays.rs:5 */
typedef struct { double r, g, b; } orc_color;      /* core/color.rs:4 */
typedef struct { orc_vec3 min, max; } orc_bounds;  /* geometry/bounds.rs:7 */

typedef struct {                                   /* scene/patterns.rs:7 */
    int32_t kind;                                  /* ORC_PAT_* */
    int32_t _pad;
    orc_mat4 transform;
    orc_color a, b;
} orc_pattern;

typedef struct {                                   /* scene/materials.rs:9 */
    orc_color color;
    double ambient, diffuse, specular, shininess, reflectivity, transparency, refractive_index;
    orc_pattern pattern;                           /* kind == ORC_PAT_NONE: Option::None */
} orc_material;

typedef struct {                                   /* rendering/camera.rs:8 */
    uint32_t hsize, vsize;
    double field_of_view;
    orc_mat4 transform;
    double pixel_size, half_width, half_height;
} orc_camera;

typedef struct {                                   /* rendering/intersections.rs:12 */
    double time;
    int32_t object;
    int32_t inside;
    orc_vec3 point, eye_vector, normal_vector, reflect_vector, over_point, under_point;
    double n1, n2;
} orc_comps;

/* Operation counters of one render (or one batch of queries).  "rays" is the
 * number of World::intersect invocations (rendering/world.rs:27); the others
 * count the per-operation events the algorithmic flop model multiplies. */
typedef struct {
    uint64_t rays;            /* World::intersect calls: primary + shadow + reflection + refraction */
    uint64_t primary;         /* camera rays */
    uint64_t shadow;          /* is_shadowed casts */
    uint64_t reflect;         /* reflection rays cast */
    uint64_t refract;         /* refraction rays cast */
    uint64_t prim_tests[5];   /* local_intersect calls per leaf kind */
    uint64_t prim_hits[5];    /* ... that returned at least one t */
    uint64_t aabb_tests;      /* Bounds::intersects calls (one per group visited) */
    uint64_t ray_transforms;  /* Ray::transform calls (one per node visited) */
    uint64_t shades;          /* shade_hit calls (= hits that were lit/shaded) */
    uint64_t lit;             /* lighting() calls that took the diffuse/specular branch */
    uint64_t specular;        /* ... that evaluated powf */
    uint64_t patterns;        /* Pattern::color_at_object calls */
    uint64_t refract_setups;  /* refracted_color calls that got past the two early-outs */
    uint64_t schlick;         /* Computations::schlick calls */
} orc_counters;

/* ---- scene configuration: a binary image of the TOML schema (scenes/loader.rs:21-187) ---- */
typedef struct {
    int32_t has_translate, has_scale, has_rotate_x, has_rotate_y, has_rotate_z;
    int32_t _pad;
    double translate[3], scale[3];
    double rotate_x, rotate_y, rotate_z; /* degrees */
} orc_transform_cfg;

typedef struct {
    double color[3];
    double ambient, diffuse, specular, shininess, reflectivity, transparency, refractive_index;
    int32_t pattern_kind; /* ORC_PAT_NONE when the table has no [pattern] */
    int32_t _pad;
    double color_a[3], color_b[3];
    orc_transform_cfg pattern_transform;
} orc_material_cfg;

typedef struct {
    int32_t kind;         /* ORC_* */
    int32_t n_children;   /* groups: direct children, which follow in depth-first pre-order */
    int32_t has_transform, has_material;
    int32_t has_minimum, has_maximum, has_closed, closed;
    double minimum, maximum;
    orc_transform_cfg transform;
    orc_material_cfg material;
} orc_object_cfg;

typedef struct {
    uint32_t width, height;
    double fov; /* degrees */
    double from[3], to[3], up[3];
} orc_camera_cfg;

typedef struct orc_world orc_world;

/* ---- core/matrices.rs ---- */
void orc_mat4_identity(orc_mat4 *out);
void orc_mat4_translate(double x, double y, double z, orc_mat4 *out);
void orc_mat4_scale(double x, double y, double z, orc_mat4 *out);
void orc_mat4_rotate_x(double r, orc_mat4 *out);
void orc_mat4_rotate_y(double r, orc_mat4 *out);
void orc_mat4_rotate_z(double r, orc_mat4 *out);
void orc_mat4_shear(double xy, double xz, double yx, double yz, double zx, double zy, orc_mat4 *out);
void orc_mat4_mul(const orc_mat4 *a, const orc_mat4 *b, orc_mat4 *out);
void orc_mat4_mul_point(const orc_mat4 *a, const orc_vec3 *p, orc_vec3 *out);
void orc_mat4_mul_vector(const orc_mat4 *a, const orc_vec3 *v, orc_vec3 *out);
void orc_mat4_transpose(const orc_mat4 *a, orc_mat4 *out);
double orc_mat4_determinant(const orc_mat4 *a);
double orc_mat4_minor(const orc_mat4 *a, int row, int col);
double orc_mat4_cofactor(const orc_mat4 *a, int row, int col);
int orc_mat4_inverse(const orc_mat4 *a, orc_mat4 *out); /* 1 = Some, 0 = None */
int orc_mat4_equal(const orc_mat4 *a, const orc_mat4 *b);

/* ---- core/tuples.rs ---- */
double orc_vec_magnitude(const orc_vec3 *v);
void orc_vec_normalize(const orc_vec3 *v, orc_vec3 *out);
double orc_vec_dot(const orc_vec3 *a, const orc_vec3 *b);
void orc_vec_cross(const orc_vec3 *a, const orc_vec3 *b, orc_vec3 *out);
void orc_vec_reflect(const orc_vec3 *v, const orc_vec3 *n, orc_vec3 *out);

/* ---- scene/transformations.rs ---- */
void orc_view_transform(const orc_vec3 *from, const orc_vec3 *to, const orc_vec3 *up, orc_mat4 *out);

/* ---- rendering/rays.rs ---- */
void orc_ray_position(const orc_ray *r, double t, orc_vec3 *out);
void orc_ray_transform(const orc_ray *r, const orc_mat4 *m, orc_ray *out);

/* ---- geometry/{sphere,planes,cubes,cylinders,cones}.rs ---- */
int orc_local_intersect(int kind, double minimum, double maximum, int closed, const orc_ray *ray, double ts[4]);
void orc_local_normal_at(int kind, double minimum, double maximum, const orc_vec3 *p, orc_vec3 *out);
void orc_shape_bounds(int kind, double minimum, double maximum, orc_bounds *out);

/* ---- geometry/bounds.rs ---- */
void orc_bounds_empty(orc_bounds *out);
void orc_bounds_infinite(orc_bounds *out);
void orc_bounds_add_point(orc_bounds *b, const orc_vec3 *p);
void orc_bounds_merge(const orc_bounds *a, const orc_bounds *b, orc_bounds *out);
void orc_bounds_transform(const orc_bounds *b, const orc_mat4 *m, orc_bounds *out);
int orc_bounds_intersects(const orc_bounds *b, const orc_ray *ray);

/* ---- scene/materials.rs, scene/patterns.rs ---- */
void orc_material_default(orc_material *out);
void orc_pattern_color_at(const orc_pattern *p, const orc_vec3 *point, orc_color *out);

/* ---- world construction ---- */
orc_world *orc_world_new(void);
void orc_world_free(orc_world *w);
void orc_world_set_light(orc_world *w, const orc_vec3 *position, const orc_color *intensity);
void orc_world_clear_light(orc_world *w);
/* Adds an object (parent = -1: top level; otherwise a group's id).  Returns its id (>= 0), or -1 on error. */
int orc_world_add(orc_world *w, int parent, int kind, const orc_mat4 *transformation, const orc_material *material,
                  double minimum, double maximum, int closed);
/* Recomputes cached world transforms (scenes/loader.rs:228-241), inverses and group bounds.  Returns 0, or -1 when
 * a matrix the renderer would have to invert is singular (the reference panics at render time). */
int orc_world_finalize(orc_world *w);
/* mode 0 ("cached"): inverses/bounds computed once, closest hit by running minimum.
 * mode 1 ("faithful"): per-ray matrix inverses, per-ray Group::bounds(), intersection list + stable sort,
 * exactly the work the reference repeats (world.rs:27-78, objects.rs:121-136). */
void orc_world_set_mode(orc_world *w, int mode);
int orc_world_object_count(const orc_world *w);
int orc_world_object_world_transform(const orc_world *w, int id, orc_mat4 *out);
int orc_world_object_material(const orc_world *w, int id, orc_material *out);
int orc_world_group_bounds(const orc_world *w, int id, orc_bounds *out);
const char *orc_last_error(void);

/* ---- scenes/loader.rs ---- */
void orc_build_transform(const orc_transform_cfg *cfg, int present, orc_mat4 *out);
int orc_build_camera(const orc_camera_cfg *cfg, orc_camera *out);
orc_world *orc_build_world(const orc_object_cfg *objects, int n_objects, const double light_position[3],
                           const double light_intensity[3]);

/* ---- rendering/world.rs, intersections.rs, objects.rs ---- */
int orc_world_intersect(const orc_world *w, const orc_ray *ray, double *ts, int32_t *ids, int cap, orc_counters *c);
int orc_world_hit(const orc_world *w, const orc_ray *ray, double *t, int32_t *id, orc_counters *c);
void orc_object_normal_at(const orc_world *w, int id, const orc_vec3 *world_point, orc_vec3 *out);
void orc_prepare_computations(const orc_world *w, int id, double t, const orc_ray *ray, orc_comps *out);
void orc_prepare_computations_for_intersections(const orc_world *w, int id, double t, const orc_ray *ray,
                                                const double *ts, const int32_t *ids, int n, orc_comps *out);
double orc_schlick(const orc_comps *c);
void orc_lighting(const orc_world *w, int id, const orc_material *m, const orc_vec3 *light_position,
                  const orc_color *light_intensity, const orc_vec3 *point, const orc_vec3 *eye, const orc_vec3 *normal,
                  int in_shadow, orc_color *out, orc_counters *c);
void orc_pattern_color_at_object(const orc_world *w, int id, const orc_pattern *p, const orc_vec3 *world_point,
                                 orc_color *out);
int orc_is_shadowed(const orc_world *w, const orc_vec3 *point, orc_counters *c);
void orc_shade_hit(const orc_world *w, const orc_comps *comps, int remaining, orc_color *out, orc_counters *c);
void orc_color_at(const orc_world *w, const orc_ray *ray, int remaining, orc_color *out, orc_counters *c);
void orc_reflected_color(const orc_world *w, const orc_comps *comps, int remaining, orc_color *out, orc_counters *c);
void orc_refracted_color(const orc_world *w, const orc_comps *comps, int remaining, orc_color *out, orc_counters *c);

/* ---- rendering/camera.rs ---- */
void orc_camera_new(uint32_t hsize, uint32_t vsize, double field_of_view, orc_camera *out);
void orc_camera_ray_for_pixel(const orc_camera *cam, uint32_t px, uint32_t py, orc_ray *out);
/* Renders rows [y0, y1) into rgb (full-frame buffer, idx = (y*hsize + x)*3).  Single-threaded like the reference;
 * disjoint row ranges may be rendered concurrently from several threads. */
void orc_camera_render_rows(const orc_camera *cam, const orc_world *w, uint32_t y0, uint32_t y1, double *rgb,
                            orc_counters *c);

/* ---- rendering/canvas.rs ---- */
uint32_t orc_quantize(double value);
void orc_canvas_quantize(const double *rgb, size_t n_values, uint8_t *out);
/* Writes the P3 text of canvas.rs:44-90 into out (capacity cap); returns the byte length needed. */
size_t orc_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
