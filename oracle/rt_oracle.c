/*
 * rt_oracle.c -- CPU restatement of the raytracer-rs per-pixel render path (plain C, f64).
 *
 * TEST INFRASTRUCTURE ONLY: see rt_oracle.h.  Build with -ffp-contract=off and without
 * -ffast-math / -march=native so that no multiply-add is fused: the reference (rustc) never
 * contracts, and a fused operation changes decisions such as `discriminant < 0.0`.
 *
 * Citations are file:line under the reference checkout (src/ prefix omitted).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char g_err[256];
const char *orc_last_error(void) { return g_err; }
static void set_err(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

/* core/floats.rs:4-6 */
static int float_equal(double a, double b) { return fabs(a - b) < ORC_EPSILON; }

/* Rust f64::max / f64::min ignore a NaN operand; C fmax/fmin have the same contract. */
static double rmax(double a, double b) { return fmax(a, b); }
static double rmin(double a, double b) { return fmin(a, b); }

/* ------------------------------------------------------------------ core/tuples.rs */

static orc_vec3 v3(double x, double y, double z) { orc_vec3 v = {x, y, z}; return v; }
static orc_vec3 vadd(orc_vec3 a, orc_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* tuples.rs:90-96,183-197 */
static orc_vec3 vsub(orc_vec3 a, orc_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* tuples.rs:98-104,199-213 */
static orc_vec3 vneg(orc_vec3 a) { return v3(-a.x, -a.y, -a.z); }                              /* tuples.rs:106-112 */
static orc_vec3 vmul(orc_vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }           /* tuples.rs:114-119 */
static orc_vec3 vdiv(orc_vec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }           /* tuples.rs:121-126 */
/* tuples.rs:59-61 */
static double vdot(orc_vec3 a, orc_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* tuples.rs:35-37: powi(2) is x*x */
static double vmag(orc_vec3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
/* tuples.rs:40-42 */
static orc_vec3 vnorm(orc_vec3 a) { return vdiv(a, vmag(a)); }
/* tuples.rs:45-51 */
static orc_vec3 vcross(orc_vec3 a, orc_vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* tuples.rs:54-56: *self - *normal * 2.0 * self.dot(normal)  ==  self - ((normal*2.0) * dot) */
static orc_vec3 vreflect(orc_vec3 v, orc_vec3 n) { return vsub(v, vmul(vmul(n, 2.0), vdot(v, n))); }

double orc_vec_magnitude(const orc_vec3 *v) { return vmag(*v); }
void orc_vec_normalize(const orc_vec3 *v, orc_vec3 *out) { *out = vnorm(*v); }
double orc_vec_dot(const orc_vec3 *a, const orc_vec3 *b) { return vdot(*a, *b); }
void orc_vec_cross(const orc_vec3 *a, const orc_vec3 *b, orc_vec3 *out) { *out = vcross(*a, *b); }
void orc_vec_reflect(const orc_vec3 *v, const orc_vec3 *n, orc_vec3 *out) { *out = vreflect(*v, *n); }

/* ------------------------------------------------------------------ core/color.rs */

static orc_color col(double r, double g, double b) { orc_color c = {r, g, b}; return c; }
static orc_color cadd(orc_color a, orc_color b) { return col(a.r + b.r, a.g + b.g, a.b + b.b); } /* color.rs:42-52 */
static orc_color csub(orc_color a, orc_color b) { return col(a.r - b.r, a.g - b.g, a.b - b.b); } /* color.rs:54-64 */
static orc_color cmul(orc_color a, double s) { return col(a.r * s, a.g * s, a.b * s); }          /* color.rs:66-71 */
static orc_color chad(orc_color a, orc_color b) { return col(a.r * b.r, a.g * b.g, a.b * b.b); } /* color.rs:75-84 */
static const orc_color BLACK = {0.0, 0.0, 0.0};

/* ------------------------------------------------------------------ core/matrices.rs */

void orc_mat4_identity(orc_mat4 *o) { /* matrices.rs:17-26 */
    memset(o, 0, sizeof *o);
    o->m[0][0] = o->m[1][1] = o->m[2][2] = o->m[3][3] = 1.0;
}
void orc_mat4_translate(double x, double y, double z, orc_mat4 *o) { /* matrices.rs:28-37 */
    orc_mat4_identity(o);
    o->m[0][3] = x; o->m[1][3] = y; o->m[2][3] = z;
}
void orc_mat4_scale(double x, double y, double z, orc_mat4 *o) { /* matrices.rs:39-48 */
    orc_mat4_identity(o);
    o->m[0][0] = x; o->m[1][1] = y; o->m[2][2] = z;
}
void orc_mat4_rotate_x(double r, orc_mat4 *o) { /* matrices.rs:50-59 */
    orc_mat4_identity(o);
    o->m[1][1] = cos(r); o->m[1][2] = -sin(r);
    o->m[2][1] = sin(r); o->m[2][2] = cos(r);
}
void orc_mat4_rotate_y(double r, orc_mat4 *o) { /* matrices.rs:61-70 */
    orc_mat4_identity(o);
    o->m[0][0] = cos(r); o->m[0][2] = sin(r);
    o->m[2][0] = -sin(r); o->m[2][2] = cos(r);
}
void orc_mat4_rotate_z(double r, orc_mat4 *o) { /* matrices.rs:72-81 */
    orc_mat4_identity(o);
    o->m[0][0] = cos(r); o->m[0][1] = -sin(r);
    o->m[1][0] = sin(r); o->m[1][1] = cos(r);
}
void orc_mat4_shear(double xy, double xz, double yx, double yz, double zx, double zy, orc_mat4 *o) { /* matrices.rs:83-92 */
    orc_mat4_identity(o);
    o->m[0][1] = xy; o->m[0][2] = xz;
    o->m[1][0] = yx; o->m[1][2] = yz;
    o->m[2][0] = zx; o->m[2][1] = zy;
}

/* matrices.rs:178-192: sum starts at 0.0 and adds the four products in order */
void orc_mat4_mul(const orc_mat4 *a, const orc_mat4 *b, orc_mat4 *out) {
    orc_mat4 r;
    for (int row = 0; row < 4; ++row)
        for (int c = 0; c < 4; ++c) {
            double sum = 0.0;
            for (int i = 0; i < 4; ++i) sum += a->m[row][i] * b->m[i][c];
            r.m[row][c] = sum;
        }
    *out = r;
}

/* matrices.rs:195-221: row[0]*x + row[1]*y + row[2]*z + row[3]*w; the fourth row is computed and dropped */
static orc_vec3 mat_mul_tuple(const orc_mat4 *a, orc_vec3 t, double w) {
    double d[4];
    for (int i = 0; i < 4; ++i) d[i] = a->m[i][0] * t.x + a->m[i][1] * t.y + a->m[i][2] * t.z + a->m[i][3] * w;
    return v3(d[0], d[1], d[2]);
}
static orc_vec3 mat_mul_point(const orc_mat4 *a, orc_vec3 p) { return mat_mul_tuple(a, p, 1.0); }
static orc_vec3 mat_mul_vector(const orc_mat4 *a, orc_vec3 v) { return mat_mul_tuple(a, v, 0.0); }
void orc_mat4_mul_point(const orc_mat4 *a, const orc_vec3 *p, orc_vec3 *out) { *out = mat_mul_point(a, *p); }
void orc_mat4_mul_vector(const orc_mat4 *a, const orc_vec3 *v, orc_vec3 *out) { *out = mat_mul_vector(a, *v); }

void orc_mat4_transpose(const orc_mat4 *a, orc_mat4 *out) { /* matrices.rs:137-145 */
    orc_mat4 r;
    for (int row = 0; row < 4; ++row)
        for (int c = 0; c < 4; ++c) r.m[c][row] = a->m[row][c];
    *out = r;
}

/* matrices.rs:310-312 */
static double det2(double m[2][2]) { return m[0][0] * m[1][1] - m[0][1] * m[1][0]; }
/* matrices.rs:268-278, 255-257, 246-253 */
static double minor3(double m[3][3], int row, int c) {
    double s[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) s[i][j] = m[i + (i >= row ? 1 : 0)][j + (j >= c ? 1 : 0)];
    return det2(s);
}
static double cofactor3(double m[3][3], int row, int c) {
    double mi = minor3(m, row, c);
    return ((row + c) % 2 == 0) ? mi : -mi;
}
/* matrices.rs:259-265 */
static double det3(double m[3][3]) {
    double result = 0.0;
    for (int c = 0; c < 3; ++c) result += m[0][c] * cofactor3(m, 0, c);
    return result;
}
/* matrices.rs:103-113, 160-162 */
double orc_mat4_minor(const orc_mat4 *a, int row, int c) {
    double s[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s[i][j] = a->m[i + (i >= row ? 1 : 0)][j + (j >= c ? 1 : 0)];
    return det3(s);
}
/* matrices.rs:151-158 */
double orc_mat4_cofactor(const orc_mat4 *a, int row, int c) {
    double mi = orc_mat4_minor(a, row, c);
    return ((row + c) % 2 == 0) ? mi : -mi;
}
/* matrices.rs:94-100 */
double orc_mat4_determinant(const orc_mat4 *a) {
    double result = 0.0;
    for (int c = 0; c < 4; ++c) result += a->m[0][c] * orc_mat4_cofactor(a, 0, c);
    return result;
}
/* matrices.rs:116-134 */
int orc_mat4_inverse(const orc_mat4 *a, orc_mat4 *out) {
    double determinant = orc_mat4_determinant(a);
    if (float_equal(determinant, 0.0)) return 0;
    orc_mat4 r;
    double inv_det = 1.0 / determinant;
    for (int row = 0; row < 4; ++row)
        for (int c = 0; c < 4; ++c) r.m[c][row] = orc_mat4_cofactor(a, row, c) * inv_det;
    *out = r;
    return 1;
}
/* matrices.rs:223-234 */
int orc_mat4_equal(const orc_mat4 *a, const orc_mat4 *b) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            if (!float_equal(a->m[r][c], b->m[r][c])) return 0;
    return 1;
}

/* ------------------------------------------------------------------ scene/transformations.rs:4-17 */
void orc_view_transform(const orc_vec3 *from, const orc_vec3 *to, const orc_vec3 *up, orc_mat4 *out) {
    orc_vec3 forward = vnorm(vsub(*to, *from));
    orc_vec3 up_normalized = vnorm(*up);
    orc_vec3 left = vcross(forward, up_normalized);
    orc_vec3 true_up = vcross(left, forward);
    orc_mat4 orientation, tr;
    orc_mat4_identity(&orientation);
    orientation.m[0][0] = left.x; orientation.m[0][1] = left.y; orientation.m[0][2] = left.z;
    orientation.m[1][0] = true_up.x; orientation.m[1][1] = true_up.y; orientation.m[1][2] = true_up.z;
    orientation.m[2][0] = -forward.x; orientation.m[2][1] = -forward.y; orientation.m[2][2] = -forward.z;
    orc_mat4_translate(-from->x, -from->y, -from->z, &tr);
    orc_mat4_mul(&orientation, &tr, out);
}

/* ------------------------------------------------------------------ rendering/rays.rs */
static orc_vec3 ray_position(orc_ray r, double t) { return vadd(r.origin, vmul(r.direction, t)); } /* rays.rs:17-19 */
static orc_ray ray_transform(orc_ray r, const orc_mat4 *m) {                                       /* rays.rs:22-24 */
    orc_ray o;
    o.origin = mat_mul_point(m, r.origin);
    o.direction = mat_mul_vector(m, r.direction);
    return o;
}
void orc_ray_position(const orc_ray *r, double t, orc_vec3 *out) { *out = ray_position(*r, t); }
void orc_ray_transform(const orc_ray *r, const orc_mat4 *m, orc_ray *out) { *out = ray_transform(*r, m); }

/* ------------------------------------------------------------------ geometry: local_intersect */

/* sphere.rs:51-74 */
static int sphere_intersect(orc_ray ray, double ts[4]) {
    orc_vec3 sphere_to_ray = vsub(ray.origin, v3(0.0, 0.0, 0.0));
    double a = vdot(ray.direction, ray.direction);
    double b = 2.0 * vdot(ray.direction, sphere_to_ray);
    double c = vdot(sphere_to_ray, sphere_to_ray) - 1.0;
    double discriminant = b * b - 4.0 * a * c; /* b.powf(2.0) == b*b */
    if (discriminant < 0.0) return 0;
    double t1 = (-b - sqrt(discriminant)) / (2.0 * a);
    double t2 = (-b + sqrt(discriminant)) / (2.0 * a);
    if (t1 <= t2) { ts[0] = t1; ts[1] = t2; } else { ts[0] = t2; ts[1] = t1; }
    return 2;
}

/* planes.rs:46-56 */
static int plane_intersect(orc_ray ray, double ts[4]) {
    if (fabs(ray.direction.y) < ORC_EPSILON) return 0;
    ts[0] = -ray.origin.y / ray.direction.y;
    return 1;
}

/* cubes.rs:41-59 (threshold EPSILON) and bounds.rs:110-128 (threshold 1e-10, arbitrary extents) */
static void check_axis(double origin, double direction, double lo, double hi, double threshold, double *tmin,
                       double *tmax) {
    double tmin_numerator = lo - origin;
    double tmax_numerator = hi - origin;
    double a, b;
    if (fabs(direction) >= threshold) {
        a = tmin_numerator / direction;
        b = tmax_numerator / direction;
    } else {
        a = tmin_numerator * INFINITY;
        b = tmax_numerator * INFINITY;
    }
    if (a > b) { *tmin = b; *tmax = a; } else { *tmin = a; *tmax = b; }
}

/* cubes.rs:63-86 */
static int cube_intersect(orc_ray ray, double ts[4]) {
    double tmin, tmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(ray.origin.x, ray.direction.x, -1.0, 1.0, ORC_EPSILON, &tmin, &tmax);
    check_axis(ray.origin.y, ray.direction.y, -1.0, 1.0, ORC_EPSILON, &ytmin, &ytmax);
    tmin = rmax(tmin, ytmin);
    tmax = rmin(tmax, ytmax);
    if (tmin > tmax) return 0;
    check_axis(ray.origin.z, ray.direction.z, -1.0, 1.0, ORC_EPSILON, &ztmin, &ztmax);
    tmin = rmax(tmin, ztmin);
    tmax = rmin(tmax, ztmax);
    if (tmin > tmax) return 0;
    ts[0] = tmin; ts[1] = tmax;
    return 2;
}

/* cylinders.rs:140-144 */
static int cyl_check_cap(orc_ray ray, double t) {
    double x = ray.origin.x + t * ray.direction.x;
    double z = ray.origin.z + t * ray.direction.z;
    return (x * x + z * z) <= 1.0;
}
/* cylinders.rs:124-138 */
static int cyl_intersect_caps(orc_ray ray, double minimum, double maximum, int closed, double ts[4], int n) {
    if (!closed || float_equal(ray.direction.y, 0.0)) return n;
    double t = (minimum - ray.origin.y) / ray.direction.y;
    if (cyl_check_cap(ray, t)) ts[n++] = t;
    t = (maximum - ray.origin.y) / ray.direction.y;
    if (cyl_check_cap(ray, t)) ts[n++] = t;
    return n;
}
/* cylinders.rs:57-99 */
static int cylinder_intersect(orc_ray ray, double minimum, double maximum, int closed, double ts[4]) {
    double a = ray.direction.x * ray.direction.x + ray.direction.z * ray.direction.z;
    int n = 0;
    if (float_equal(a, 0.0)) {
        if (closed) n = cyl_intersect_caps(ray, minimum, maximum, closed, ts, n);
        return n;
    }
    double b = 2.0 * ray.origin.x * ray.direction.x + 2.0 * ray.origin.z * ray.direction.z;
    double c = ray.origin.x * ray.origin.x + ray.origin.z * ray.origin.z - 1.0;
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return 0;
    double sqrt_disc = sqrt(discriminant);
    double denom = 2.0 * a;
    double t0 = (-b - sqrt_disc) / denom;
    double t1 = (-b + sqrt_disc) / denom;
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    double y0 = ray.origin.y + t0 * ray.direction.y;
    double y1 = ray.origin.y + t1 * ray.direction.y;
    if (minimum < y0 && y0 < maximum) ts[n++] = t0;
    if (minimum < y1 && y1 < maximum) ts[n++] = t1;
    return cyl_intersect_caps(ray, minimum, maximum, closed, ts, n);
}

/* cones.rs:156-161 */
static int cone_check_cap(orc_ray ray, double t, double y) {
    double x = ray.origin.x + t * ray.direction.x;
    double z = ray.origin.z + t * ray.direction.z;
    double radius = fabs(y);
    return (x * x + z * z) <= radius * radius;
}
/* cones.rs:140-154 */
static int cone_intersect_caps(orc_ray ray, double minimum, double maximum, int closed, double ts[4], int n) {
    if (!closed || float_equal(ray.direction.y, 0.0)) return n;
    double t = (minimum - ray.origin.y) / ray.direction.y;
    if (cone_check_cap(ray, t, minimum)) ts[n++] = t;
    t = (maximum - ray.origin.y) / ray.direction.y;
    if (cone_check_cap(ray, t, maximum)) ts[n++] = t;
    return n;
}
/* cones.rs:57-107 */
static int cone_intersect(orc_ray ray, double minimum, double maximum, int closed, double ts[4]) {
    orc_vec3 o = ray.origin, d = ray.direction;
    double a = d.x * d.x - d.y * d.y + d.z * d.z;
    double b = 2.0 * o.x * d.x - 2.0 * o.y * d.y + 2.0 * o.z * d.z;
    double c = o.x * o.x - o.y * o.y + o.z * o.z;
    int n = 0;
    if (float_equal(a, 0.0)) {
        if (float_equal(b, 0.0)) return cone_intersect_caps(ray, minimum, maximum, closed, ts, n);
        double t = -c / (2.0 * b);
        double y = o.y + t * d.y;
        if (minimum < y && y < maximum) ts[n++] = t;
        return cone_intersect_caps(ray, minimum, maximum, closed, ts, n);
    }
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return cone_intersect_caps(ray, minimum, maximum, closed, ts, n);
    double sqrt_disc = sqrt(discriminant);
    double denom = 2.0 * a;
    double t0 = (-b - sqrt_disc) / denom;
    double t1 = (-b + sqrt_disc) / denom;
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    double y0 = o.y + t0 * d.y;
    if (minimum < y0 && y0 < maximum) ts[n++] = t0;
    double y1 = o.y + t1 * d.y;
    if (minimum < y1 && y1 < maximum) ts[n++] = t1;
    return cone_intersect_caps(ray, minimum, maximum, closed, ts, n);
}

static int local_intersect(int kind, double minimum, double maximum, int closed, orc_ray ray, double ts[4]) {
    switch (kind) {
    case ORC_SPHERE: return sphere_intersect(ray, ts);
    case ORC_PLANE: return plane_intersect(ray, ts);
    case ORC_CUBE: return cube_intersect(ray, ts);
    case ORC_CYLINDER: return cylinder_intersect(ray, minimum, maximum, closed, ts);
    case ORC_CONE: return cone_intersect(ray, minimum, maximum, closed, ts);
    default: return 0;
    }
}
int orc_local_intersect(int kind, double minimum, double maximum, int closed, const orc_ray *ray, double ts[4]) {
    return local_intersect(kind, minimum, maximum, closed, *ray, ts);
}

/* ------------------------------------------------------------------ geometry: local_normal_at */
static orc_vec3 local_normal_at(int kind, double minimum, double maximum, orc_vec3 p) {
    switch (kind) {
    case ORC_SPHERE: /* sphere.rs:77-80 */
        return vsub(p, v3(0.0, 0.0, 0.0));
    case ORC_PLANE: /* planes.rs:58-60 */
        return v3(0.0, 1.0, 0.0);
    case ORC_CUBE: { /* cubes.rs:88-98 */
        double maxc = rmax(rmax(fabs(p.x), fabs(p.y)), fabs(p.z));
        if (fabs(maxc - fabs(p.x)) < ORC_EPSILON) return v3(p.x, 0.0, 0.0);
        if (fabs(maxc - fabs(p.y)) < ORC_EPSILON) return v3(0.0, p.y, 0.0);
        return v3(0.0, 0.0, p.z);
    }
    case ORC_CYLINDER: { /* cylinders.rs:101-111 */
        double dist = p.x * p.x + p.z * p.z;
        if (dist < 1.0 && p.y >= maximum - ORC_EPSILON) return v3(0.0, 1.0, 0.0);
        if (dist < 1.0 && p.y <= minimum + ORC_EPSILON) return v3(0.0, -1.0, 0.0);
        return v3(p.x, 0.0, p.z);
    }
    case ORC_CONE: { /* cones.rs:109-126 */
        double dist = p.x * p.x + p.z * p.z;
        if (dist < 1.0 && p.y >= maximum - ORC_EPSILON) return v3(0.0, 1.0, 0.0);
        if (dist < 1.0 && p.y <= minimum + ORC_EPSILON) return v3(0.0, -1.0, 0.0);
        double y = sqrt(p.x * p.x + p.z * p.z);
        if (p.y > 0.0) y = -y;
        return v3(p.x, y, p.z);
    }
    default:
        return v3(NAN, NAN, NAN); /* groups.rs:125-127 panics */
    }
}
void orc_local_normal_at(int kind, double minimum, double maximum, const orc_vec3 *p, orc_vec3 *out) {
    *out = local_normal_at(kind, minimum, maximum, *p);
}

/* ------------------------------------------------------------------ geometry/bounds.rs */
void orc_bounds_empty(orc_bounds *o) { /* bounds.rs:20-25 */
    o->min = v3(INFINITY, INFINITY, INFINITY);
    o->max = v3(-INFINITY, -INFINITY, -INFINITY);
}
void orc_bounds_infinite(orc_bounds *o) { /* bounds.rs:28-33 */
    o->min = v3(-INFINITY, -INFINITY, -INFINITY);
    o->max = v3(INFINITY, INFINITY, INFINITY);
}
void orc_bounds_add_point(orc_bounds *b, const orc_vec3 *p) { /* bounds.rs:36-44 */
    b->min.x = rmin(b->min.x, p->x); b->min.y = rmin(b->min.y, p->y); b->min.z = rmin(b->min.z, p->z);
    b->max.x = rmax(b->max.x, p->x); b->max.y = rmax(b->max.y, p->y); b->max.z = rmax(b->max.z, p->z);
}
void orc_bounds_merge(const orc_bounds *a, const orc_bounds *b, orc_bounds *out) { /* bounds.rs:47-52 */
    orc_bounds r = *a;
    orc_bounds_add_point(&r, &b->min);
    orc_bounds_add_point(&r, &b->max);
    *out = r;
}
void orc_bounds_transform(const orc_bounds *b, const orc_mat4 *m, orc_bounds *out) { /* bounds.rs:57-89 */
    int inf_min = isinf(b->min.x) || isinf(b->min.y) || isinf(b->min.z);
    int inf_max = isinf(b->max.x) || isinf(b->max.y) || isinf(b->max.z);
    if (inf_min || inf_max) { orc_bounds_infinite(out); return; }
    orc_vec3 corners[8] = {
        b->min,
        v3(b->min.x, b->min.y, b->max.z), v3(b->min.x, b->max.y, b->min.z), v3(b->min.x, b->max.y, b->max.z),
        v3(b->max.x, b->min.y, b->min.z), v3(b->max.x, b->min.y, b->max.z), v3(b->max.x, b->max.y, b->min.z),
        b->max,
    };
    orc_bounds r;
    orc_bounds_empty(&r);
    for (int i = 0; i < 8; ++i) {
        orc_vec3 p = mat_mul_point(m, corners[i]);
        orc_bounds_add_point(&r, &p);
    }
    *out = r;
}
int orc_bounds_intersects(const orc_bounds *b, const orc_ray *ray) { /* bounds.rs:94-106 */
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(ray->origin.x, ray->direction.x, b->min.x, b->max.x, 1e-10, &xtmin, &xtmax);
    check_axis(ray->origin.y, ray->direction.y, b->min.y, b->max.y, 1e-10, &ytmin, &ytmax);
    check_axis(ray->origin.z, ray->direction.z, b->min.z, b->max.z, 1e-10, &ztmin, &ztmax);
    double tmin = rmax(rmax(xtmin, ytmin), ztmin);
    double tmax = rmin(rmin(xtmax, ytmax), ztmax);
    return tmin <= tmax;
}
void orc_shape_bounds(int kind, double minimum, double maximum, orc_bounds *o) {
    switch (kind) {
    case ORC_SPHERE: /* sphere.rs:83-85 */
    case ORC_CUBE:   /* cubes.rs:101-103 */
        o->min = v3(-1.0, -1.0, -1.0); o->max = v3(1.0, 1.0, 1.0);
        break;
    case ORC_PLANE: /* planes.rs:63-68 */
        o->min = v3(-INFINITY, 0.0, -INFINITY); o->max = v3(INFINITY, 0.0, INFINITY);
        break;
    case ORC_CYLINDER: /* cylinders.rs:115-120 */
        o->min = v3(-1.0, minimum, -1.0); o->max = v3(1.0, maximum, 1.0);
        break;
    case ORC_CONE: { /* cones.rs:130-136 */
        double limit = rmax(fabs(minimum), fabs(maximum));
        o->min = v3(-limit, minimum, -limit); o->max = v3(limit, maximum, limit);
        break;
    }
    default:
        orc_bounds_empty(o);
    }
}

/* ------------------------------------------------------------------ scene/materials.rs, patterns.rs */
void orc_material_default(orc_material *m) { /* materials.rs:35-45 */
    memset(m, 0, sizeof *m);
    m->color = col(1.0, 1.0, 1.0);
    m->ambient = 0.1; m->diffuse = 0.9; m->specular = 0.9; m->shininess = 200.0;
    m->reflectivity = 0.0; m->transparency = 0.0; m->refractive_index = 1.0;
    m->pattern.kind = ORC_PAT_NONE;
    orc_mat4_identity(&m->pattern.transform);
}

/* Rust `as i32`: saturating, NaN -> 0 */
static int32_t sat_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}

void orc_pattern_color_at(const orc_pattern *p, const orc_vec3 *pt, orc_color *out) { /* patterns.rs:118-135 */
    switch (p->kind) {
    case ORC_PAT_STRIPE: /* patterns.rs:81-87: `%` on f64 is fmod */
        *out = (fmod(floor(pt->x), 2.0) == 0.0) ? p->a : p->b;
        break;
    case ORC_PAT_GRADIENT: { /* patterns.rs:89-93 */
        orc_color distance = csub(p->b, p->a);
        double fraction = pt->x;
        if (fraction < 0.0) fraction = 0.0; else if (fraction > 1.0) fraction = 1.0; /* f64::clamp keeps NaN */
        *out = cadd(p->a, cmul(distance, fraction));
        break;
    }
    case ORC_PAT_RING: /* patterns.rs:95-102 */
        *out = (fmod(floor(sqrt(pt->x * pt->x + pt->z * pt->z)), 2.0) == 0.0) ? p->a : p->b;
        break;
    case ORC_PAT_CHECKERS: { /* patterns.rs:104-116 */
        const double eps = 1e-6;
        uint32_t x = (uint32_t)sat_i32(floor(pt->x + eps));
        uint32_t y = (uint32_t)sat_i32(floor(pt->y + eps));
        uint32_t z = (uint32_t)sat_i32(floor(pt->z + eps));
        *out = (((x + y + z) & 1u) == 0u) ? p->a : p->b;
        break;
    }
    case ORC_PAT_TEST: /* patterns.rs:133 */
        *out = col(pt->x, pt->y, pt->z);
        break;
    default:
        *out = BLACK;
    }
}

/* ------------------------------------------------------------------ world storage */
typedef struct {
    int kind, parent;
    orc_mat4 T, W;       /* Object::transformation(), Object::world_transformation() */
    orc_material mat;
    double minimum, maximum;
    int closed;
    int n_children, cap_children;
    int *children;
    /* pure functions of the scene, cached by orc_world_finalize */
    orc_mat4 Tinv, Winv;
    int Tinv_ok, Winv_ok;
    orc_bounds bounds;   /* groups: Group::bounds() */
} obj_t;

struct orc_world {
    obj_t *objs;
    int n, cap;
    int *top;
    int n_top, cap_top;
    int has_light;
    orc_vec3 light_position;
    orc_color light_intensity;
    int mode;
    int finalized;
};

orc_world *orc_world_new(void) {
    orc_world *w = (orc_world *)calloc(1, sizeof *w);
    return w;
}
void orc_world_free(orc_world *w) {
    if (!w) return;
    for (int i = 0; i < w->n; ++i) free(w->objs[i].children);
    free(w->objs);
    free(w->top);
    free(w);
}
void orc_world_set_light(orc_world *w, const orc_vec3 *p, const orc_color *i) {
    w->has_light = 1; w->light_position = *p; w->light_intensity = *i;
}
void orc_world_clear_light(orc_world *w) { w->has_light = 0; }
void orc_world_set_mode(orc_world *w, int mode) { w->mode = mode; }
int orc_world_object_count(const orc_world *w) { return w->n; }

static int push_int(int **arr, int *n, int *cap, int v) {
    if (*n == *cap) {
        int nc = *cap ? *cap * 2 : 8;
        int *na = (int *)realloc(*arr, (size_t)nc * sizeof(int));
        if (!na) return -1;
        *arr = na; *cap = nc;
    }
    (*arr)[(*n)++] = v;
    return 0;
}

int orc_world_add(orc_world *w, int parent, int kind, const orc_mat4 *T, const orc_material *mat, double minimum,
                  double maximum, int closed) {
    if (kind < ORC_SPHERE || kind > ORC_GROUP) { set_err("unknown object kind"); return -1; }
    if (parent >= w->n || (parent >= 0 && w->objs[parent].kind != ORC_GROUP)) { set_err("parent is not a group"); return -1; }
    if (w->n == w->cap) {
        int nc = w->cap ? w->cap * 2 : 16;
        obj_t *no = (obj_t *)realloc(w->objs, (size_t)nc * sizeof(obj_t));
        if (!no) { set_err("out of memory"); return -1; }
        w->objs = no; w->cap = nc;
    }
    int id = w->n++;
    obj_t *o = &w->objs[id];
    memset(o, 0, sizeof *o);
    o->kind = kind; o->parent = parent;
    o->T = *T; orc_mat4_identity(&o->W);
    o->mat = *mat;
    o->minimum = minimum; o->maximum = maximum; o->closed = closed;
    if (parent < 0) push_int(&w->top, &w->n_top, &w->cap_top, id);
    else push_int(&w->objs[parent].children, &w->objs[parent].n_children, &w->objs[parent].cap_children, id);
    w->finalized = 0;
    return id;
}

/* groups.rs:96-105 via objects.rs:284-288: children of a group get parent_world * local, recursively */
static void propagate_world(orc_world *w, int gid) {
    obj_t *g = &w->objs[gid];
    for (int i = 0; i < g->n_children; ++i) {
        obj_t *c = &w->objs[g->children[i]];
        orc_mat4_mul(&g->W, &c->T, &c->W);
        if (c->kind == ORC_GROUP) propagate_world(w, g->children[i]);
    }
}

/* groups.rs:130-149 */
static void group_bounds(const orc_world *w, int gid, orc_bounds *out) {
    const obj_t *g = &w->objs[gid];
    orc_bounds gb;
    orc_bounds_empty(&gb);
    for (int i = 0; i < g->n_children; ++i) {
        const obj_t *c = &w->objs[g->children[i]];
        orc_bounds cb, tb;
        if (c->kind == ORC_GROUP) group_bounds(w, g->children[i], &cb);
        else orc_shape_bounds(c->kind, c->minimum, c->maximum, &cb);
        orc_bounds_transform(&cb, &c->T, &tb);
        orc_bounds_merge(&gb, &tb, &gb);
    }
    *out = gb;
}

int orc_world_finalize(orc_world *w) {
    /* loader.rs:229-241: top-level world = local; a top-level group then rebuilds its children with
     * (identity * group.transformation) * child.transformation (groups.rs:85-93), and update_transforms
     * propagates below that (objects.rs:284-288). */
    orc_mat4 ident;
    orc_mat4_identity(&ident);
    for (int k = 0; k < w->n_top; ++k) {
        int id = w->top[k];
        obj_t *o = &w->objs[id];
        o->W = o->T;
        if (o->kind == ORC_GROUP) {
            orc_mat4 pg;
            orc_mat4_mul(&ident, &o->T, &pg);
            for (int i = 0; i < o->n_children; ++i) {
                obj_t *c = &w->objs[o->children[i]];
                orc_mat4_mul(&pg, &c->T, &c->W);
                if (c->kind == ORC_GROUP) propagate_world(w, o->children[i]);
            }
        }
    }
    int ok = 0;
    for (int i = 0; i < w->n; ++i) {
        obj_t *o = &w->objs[i];
        o->Tinv_ok = orc_mat4_inverse(&o->T, &o->Tinv);
        o->Winv_ok = orc_mat4_inverse(&o->W, &o->Winv);
        if (!o->Tinv_ok) { set_err("object transformation is not invertible"); ok = -1; }
        if (o->kind != ORC_GROUP && !o->Winv_ok) { set_err("object world transformation is not invertible"); ok = -1; }
        if (o->kind != ORC_GROUP && o->mat.pattern.kind != ORC_PAT_NONE) {
            orc_mat4 tmp;
            if (!orc_mat4_inverse(&o->mat.pattern.transform, &tmp)) { set_err("pattern transform is not invertible"); ok = -1; }
        }
        if (o->kind == ORC_GROUP) group_bounds(w, i, &o->bounds);
    }
    w->finalized = 1;
    return ok;
}

int orc_world_object_world_transform(const orc_world *w, int id, orc_mat4 *out) {
    if (id < 0 || id >= w->n) return -1;
    *out = w->objs[id].W;
    return 0;
}
int orc_world_object_material(const orc_world *w, int id, orc_material *out) {
    if (id < 0 || id >= w->n) return -1;
    *out = w->objs[id].mat;
    return 0;
}
int orc_world_group_bounds(const orc_world *w, int id, orc_bounds *out) {
    if (id < 0 || id >= w->n || w->objs[id].kind != ORC_GROUP) return -1;
    group_bounds(w, id, out);
    return 0;
}

/* ------------------------------------------------------------------ rendering/world.rs: intersect */

typedef struct {
    double *ts;
    int32_t *ids;
    int n, cap;
    double small_ts[64];
    int32_t small_ids[64];
} xs_list;

static void xs_init(xs_list *l) { l->ts = l->small_ts; l->ids = l->small_ids; l->n = 0; l->cap = 64; }
static void xs_free(xs_list *l) {
    if (l->ts != l->small_ts) { free(l->ts); free(l->ids); }
}
static void xs_push(xs_list *l, double t, int id) {
    if (l->n == l->cap) {
        int nc = l->cap * 2;
        double *nt = (double *)malloc((size_t)nc * sizeof(double));
        int32_t *ni = (int32_t *)malloc((size_t)nc * sizeof(int32_t));
        memcpy(nt, l->ts, (size_t)l->n * sizeof(double));
        memcpy(ni, l->ids, (size_t)l->n * sizeof(int32_t));
        xs_free(l);
        l->ts = nt; l->ids = ni; l->cap = nc;
    }
    l->ts[l->n] = t; l->ids[l->n] = id; l->n++;
}

static orc_mat4 obj_Tinv(const orc_world *w, const obj_t *o) {
    if (w->mode == 1) { /* objects.rs:122-126 / world.rs:53-58: inverse recomputed per ray */
        orc_mat4 inv;
        if (!orc_mat4_inverse(&o->T, &inv)) memset(&inv, 0xff, sizeof inv);
        return inv;
    }
    return o->Tinv;
}
static orc_mat4 obj_Winv(const orc_world *w, const obj_t *o) {
    if (w->mode == 1) { /* objects.rs:87-102 */
        orc_mat4 inv;
        if (!orc_mat4_inverse(&o->W, &inv)) memset(&inv, 0xff, sizeof inv);
        return inv;
    }
    return o->Winv;
}

/* world.rs:41-78 */
static void collect_intersections(const orc_world *w, int id, orc_ray ray, xs_list *xs, orc_counters *c) {
    const obj_t *o = &w->objs[id];
    if (o->kind == ORC_GROUP) {
        orc_bounds bounds;
        if (w->mode == 1) group_bounds(w, id, &bounds); else bounds = o->bounds;
        orc_mat4 inv = obj_Tinv(w, o);
        orc_ray local_ray = ray_transform(ray, &inv);
        if (c) { c->ray_transforms++; c->aabb_tests++; }
        if (!orc_bounds_intersects(&bounds, &local_ray)) return;
        for (int i = 0; i < o->n_children; ++i) collect_intersections(w, o->children[i], local_ray, xs, c);
    } else {
        orc_mat4 inv = obj_Tinv(w, o); /* objects.rs:121-136 */
        orc_ray local_ray = ray_transform(ray, &inv);
        double ts[4];
        int n = local_intersect(o->kind, o->minimum, o->maximum, o->closed, local_ray, ts);
        if (c) { c->ray_transforms++; c->prim_tests[o->kind]++; if (n) c->prim_hits[o->kind]++; }
        for (int i = 0; i < n; ++i) xs_push(xs, ts[i], id);
    }
}

/* world.rs:34: stable sort, comparator partial_cmp().unwrap_or(Equal) */
static void xs_sort(xs_list *l) {
    for (int i = 1; i < l->n; ++i) {
        double t = l->ts[i];
        int32_t id = l->ids[i];
        int j = i;
        while (j > 0 && l->ts[j - 1] > t) { l->ts[j] = l->ts[j - 1]; l->ids[j] = l->ids[j - 1]; --j; }
        l->ts[j] = t; l->ids[j] = id;
    }
}

/* world.rs:27-36 */
static void world_intersect(const orc_world *w, orc_ray ray, xs_list *xs, orc_counters *c) {
    if (c) c->rays++;
    for (int k = 0; k < w->n_top; ++k) collect_intersections(w, w->top[k], ray, xs, c);
    xs_sort(xs);
}

/* intersections.rs:37-39 on the sorted list */
static int xs_hit(const xs_list *xs) {
    for (int i = 0; i < xs->n; ++i)
        if (xs->ts[i] >= 0.0) return i;
    return -1;
}

/* Closest hit without materialising the list: the first element with t >= 0 of a stable sort is the smallest
 * t >= 0, ties going to the earliest emitted, i.e. a running minimum updated on strict `<`. */
typedef struct { int found; double t; int id; } best_t;
static void closest_collect(const orc_world *w, int id, orc_ray ray, best_t *best, orc_counters *c) {
    const obj_t *o = &w->objs[id];
    orc_ray local_ray = ray_transform(ray, &o->Tinv);
    if (c) c->ray_transforms++;
    if (o->kind == ORC_GROUP) {
        if (c) c->aabb_tests++;
        if (!orc_bounds_intersects(&o->bounds, &local_ray)) return;
        for (int i = 0; i < o->n_children; ++i) closest_collect(w, o->children[i], local_ray, best, c);
    } else {
        double ts[4];
        int n = local_intersect(o->kind, o->minimum, o->maximum, o->closed, local_ray, ts);
        if (c) { c->prim_tests[o->kind]++; if (n) c->prim_hits[o->kind]++; }
        for (int i = 0; i < n; ++i)
            if (ts[i] >= 0.0 && (!best->found || ts[i] < best->t)) { best->found = 1; best->t = ts[i]; best->id = id; }
    }
}

static int world_hit(const orc_world *w, orc_ray ray, double *t, int *id, orc_counters *c) {
    if (w->mode == 1) {
        xs_list xs;
        xs_init(&xs);
        world_intersect(w, ray, &xs, c);
        int h = xs_hit(&xs);
        if (h >= 0) { *t = xs.ts[h]; *id = xs.ids[h]; }
        xs_free(&xs);
        return h >= 0;
    }
    best_t best = {0, 0.0, -1};
    if (c) c->rays++;
    for (int k = 0; k < w->n_top; ++k) closest_collect(w, w->top[k], ray, &best, c);
    if (best.found) { *t = best.t; *id = best.id; }
    return best.found;
}

int orc_world_intersect(const orc_world *w, const orc_ray *ray, double *ts, int32_t *ids, int cap, orc_counters *c) {
    xs_list xs;
    xs_init(&xs);
    /* always the list form here: callers want the whole sorted list */
    world_intersect(w, *ray, &xs, c);
    int n = xs.n;
    for (int i = 0; i < n && i < cap; ++i) { ts[i] = xs.ts[i]; ids[i] = xs.ids[i]; }
    xs_free(&xs);
    return n;
}
int orc_world_hit(const orc_world *w, const orc_ray *ray, double *t, int32_t *id, orc_counters *c) {
    int iid = -1;
    double tt = 0.0;
    int r = world_hit(w, *ray, &tt, &iid, c);
    if (r) { *t = tt; *id = iid; }
    return r;
}

/* ------------------------------------------------------------------ rendering/objects.rs:87-105,171-184 */
static orc_vec3 world_to_object(const orc_world *w, const obj_t *o, orc_vec3 p) {
    orc_mat4 inv = obj_Winv(w, o);
    return mat_mul_point(&inv, p);
}
static orc_vec3 normal_at(const orc_world *w, const obj_t *o, orc_vec3 world_point) {
    orc_vec3 local_point = world_to_object(w, o, world_point);
    orc_vec3 local_normal = local_normal_at(o->kind, o->minimum, o->maximum, local_point);
    orc_mat4 inv = obj_Winv(w, o), tr;
    orc_mat4_transpose(&inv, &tr);
    orc_vec3 transformed = mat_mul_vector(&tr, local_normal);
    return vnorm(v3(transformed.x, transformed.y, transformed.z));
}
void orc_object_normal_at(const orc_world *w, int id, const orc_vec3 *p, orc_vec3 *out) {
    *out = normal_at(w, &w->objs[id], *p);
}

/* ------------------------------------------------------------------ rendering/intersections.rs */

/* intersections.rs:45-118.  (ts, ids, n) is the intersection list handed in; the renderer passes the
 * one-element list [self] (intersections.rs:41-43), so n1 = 1.0 and n2 = the hit object's refractive index. */
void orc_prepare_computations_for_intersections(const orc_world *w, int id, double t, const orc_ray *ray,
                                                const double *ts, const int32_t *ids, int n, orc_comps *out) {
    const obj_t *o = &w->objs[id];
    orc_vec3 normal_vector = normal_at(w, o, ray_position(*ray, t));
    orc_vec3 eye_vector = vneg(ray->direction);
    int inside = 0;
    if (vdot(normal_vector, eye_vector) < 0.0) { inside = 1; normal_vector = vneg(normal_vector); }
    orc_vec3 reflect_vector = vreflect(ray->direction, normal_vector);
    orc_vec3 point = ray_position(*ray, t);
    orc_vec3 over_point = vadd(point, vmul(normal_vector, ORC_EPSILON));
    orc_vec3 under_point = vsub(point, vmul(normal_vector, ORC_EPSILON));

    double n1 = 1.0, n2 = 1.0;
    int cap = n > 0 ? n : 1;
    int32_t *containers = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
    int nc = 0;
    for (int i = 0; i < n; ++i) {
        int is_hit = (ids[i] == id) && float_equal(ts[i], t);
        if (is_hit) n1 = nc ? w->objs[containers[nc - 1]].mat.refractive_index : 1.0;
        int pos = -1;
        for (int k = 0; k < nc; ++k)
            if (containers[k] == ids[i]) { pos = k; break; }
        if (pos >= 0) {
            memmove(&containers[pos], &containers[pos + 1], (size_t)(nc - pos - 1) * sizeof(int32_t));
            nc--;
        } else {
            containers[nc++] = ids[i];
        }
        if (is_hit) {
            n2 = nc ? w->objs[containers[nc - 1]].mat.refractive_index : 1.0;
            break;
        }
    }
    free(containers);

    out->time = t; out->object = id; out->inside = inside;
    out->point = point; out->eye_vector = eye_vector; out->normal_vector = normal_vector;
    out->reflect_vector = reflect_vector; out->over_point = over_point; out->under_point = under_point;
    out->n1 = n1; out->n2 = n2;
}

/* intersections.rs:41-43 */
void orc_prepare_computations(const orc_world *w, int id, double t, const orc_ray *ray, orc_comps *out) {
    int32_t ids[1] = {id};
    double ts[1] = {t};
    orc_prepare_computations_for_intersections(w, id, t, ray, ts, ids, 1, out);
}

/* powi with the constant exponents used on the path: x.powi(2) = x*x; x.powi(5) = x*((x*x)*(x*x)) (binary
 * exponentiation, as both LLVM's constant expansion and compiler-rt's __powidf2 do it). */
static double powi2(double x) { return x * x; }
static double powi5(double x) { double x2 = x * x; double x4 = x2 * x2; return x * x4; }

/* intersections.rs:123-136 */
double orc_schlick(const orc_comps *c) {
    double cosv = vdot(c->eye_vector, c->normal_vector);
    if (c->n1 > c->n2) {
        double n = c->n1 / c->n2;
        double sin_2t = powi2(n) * (1.0 - powi2(cosv));
        if (sin_2t > 1.0) return 1.0;
        cosv = sqrt(1.0 - sin_2t);
    }
    double r0 = powi2((c->n1 - c->n2) / (c->n1 + c->n2));
    return r0 + (1.0 - r0) * powi5(1.0 - cosv);
}

/* ------------------------------------------------------------------ scene/patterns.rs:137-146, materials.rs:122-172 */
void orc_pattern_color_at_object(const orc_world *w, int id, const orc_pattern *p, const orc_vec3 *world_point,
                                 orc_color *out) {
    orc_vec3 object_point = world_to_object(w, &w->objs[id], *world_point);
    orc_mat4 inv;
    if (!orc_mat4_inverse(&p->transform, &inv)) memset(&inv, 0xff, sizeof inv);
    orc_vec3 pattern_point = mat_mul_point(&inv, object_point);
    orc_pattern_color_at(p, &pattern_point, out);
}

void orc_lighting(const orc_world *w, int id, const orc_material *m, const orc_vec3 *light_position,
                  const orc_color *light_intensity, const orc_vec3 *point, const orc_vec3 *eye, const orc_vec3 *normal,
                  int in_shadow, orc_color *out, orc_counters *c) {
    orc_color color;
    if (m->pattern.kind != ORC_PAT_NONE) {
        orc_pattern_color_at_object(w, id, &m->pattern, point, &color);
        if (c) c->patterns++;
    } else {
        color = m->color;
    }
    orc_color effective_color = chad(color, *light_intensity);
    orc_vec3 light_vector = vnorm(vsub(*light_position, *point));
    orc_color ambient = cmul(effective_color, m->ambient);
    double light_dot_normal = vdot(light_vector, *normal);
    orc_color diffuse = BLACK, specular = BLACK;
    if (light_dot_normal >= 0.0 && !in_shadow) {
        if (c) c->lit++;
        diffuse = cmul(cmul(effective_color, m->diffuse), light_dot_normal);
        orc_vec3 reflect_vector = vreflect(vneg(light_vector), *normal);
        double reflect_dot_eye = vdot(reflect_vector, *eye);
        if (reflect_dot_eye > 0.0) {
            if (c) c->specular++;
            double factor = pow(reflect_dot_eye, m->shininess);
            specular = cmul(cmul(*light_intensity, m->specular), factor);
        }
    }
    *out = cadd(cadd(ambient, diffuse), specular);
}

/* ------------------------------------------------------------------ rendering/world.rs:80-198 */
static orc_color color_at_internal(const orc_world *w, orc_ray ray, int remaining, orc_counters *c);

/* world.rs:128-149 */
static int is_shadowed(const orc_world *w, orc_vec3 point, orc_counters *c) {
    if (!w->has_light) return 0;
    orc_vec3 v = vsub(w->light_position, point);
    double distance = vmag(v);
    orc_vec3 direction = vnorm(v);
    orc_ray ray = {point, direction};
    double t = 0.0;
    int id = -1;
    if (c) c->shadow++;
    if (world_hit(w, ray, &t, &id, c)) return t < distance;
    return 0;
}
int orc_is_shadowed(const orc_world *w, const orc_vec3 *p, orc_counters *c) { return is_shadowed(w, *p, c); }

/* world.rs:155-168 */
static orc_color reflected_color_internal(const orc_world *w, const orc_comps *comps, int remaining, orc_counters *c) {
    if (remaining == 0) return BLACK;
    double reflectivity = w->objs[comps->object].mat.reflectivity;
    if (reflectivity == 0.0) return BLACK;
    orc_ray reflect_ray = {comps->over_point, comps->reflect_vector};
    if (c) c->reflect++;
    orc_color color = color_at_internal(w, reflect_ray, remaining - 1, c);
    return cmul(color, reflectivity);
}

/* world.rs:174-198 */
static orc_color refracted_color_internal(const orc_world *w, const orc_comps *comps, int remaining, orc_counters *c) {
    if (remaining == 0) return BLACK;
    double transparency = w->objs[comps->object].mat.transparency;
    if (transparency == 0.0) return BLACK;
    if (c) c->refract_setups++;
    double n_ratio = comps->n1 / comps->n2;
    double cos_i = vdot(comps->eye_vector, comps->normal_vector);
    double sin2_t = powi2(n_ratio) * (1.0 - powi2(cos_i));
    if (sin2_t > 1.0) return BLACK;
    double cos_t = sqrt(1.0 - sin2_t);
    orc_vec3 direction = vsub(vmul(comps->normal_vector, n_ratio * cos_i - cos_t), vmul(comps->eye_vector, n_ratio));
    orc_ray refract_ray = {comps->under_point, direction};
    if (c) c->refract++;
    orc_color color = color_at_internal(w, refract_ray, remaining - 1, c);
    return cmul(color, transparency);
}

/* world.rs:84-111 */
static orc_color shade_hit_internal(const orc_world *w, const orc_comps *comps, int remaining, orc_counters *c) {
    if (!w->has_light) return BLACK;
    if (c) c->shades++;
    int in_shadow = is_shadowed(w, comps->over_point, c);
    const orc_material *m = &w->objs[comps->object].mat;
    orc_color surface;
    orc_lighting(w, comps->object, m, &w->light_position, &w->light_intensity, &comps->point, &comps->eye_vector,
                 &comps->normal_vector, in_shadow, &surface, c);
    orc_color reflected = reflected_color_internal(w, comps, remaining, c);
    orc_color refracted = refracted_color_internal(w, comps, remaining, c);
    if (m->reflectivity > 0.0 && m->transparency > 0.0) {
        if (c) c->schlick++;
        double reflectance = orc_schlick(comps);
        return cadd(cadd(surface, cmul(reflected, reflectance)), cmul(refracted, 1.0 - reflectance));
    }
    return cadd(cadd(surface, reflected), refracted);
}

/* world.rs:117-126 */
static orc_color color_at_internal(const orc_world *w, orc_ray ray, int remaining, orc_counters *c) {
    double t = 0.0;
    int id = -1;
    if (!world_hit(w, ray, &t, &id, c)) return BLACK;
    orc_comps comps;
    orc_prepare_computations(w, id, t, &ray, &comps);
    return shade_hit_internal(w, &comps, remaining, c);
}

void orc_shade_hit(const orc_world *w, const orc_comps *comps, int remaining, orc_color *out, orc_counters *c) {
    *out = shade_hit_internal(w, comps, remaining, c);
}
void orc_color_at(const orc_world *w, const orc_ray *ray, int remaining, orc_color *out, orc_counters *c) {
    *out = color_at_internal(w, *ray, remaining, c);
}
void orc_reflected_color(const orc_world *w, const orc_comps *comps, int remaining, orc_color *out, orc_counters *c) {
    *out = reflected_color_internal(w, comps, remaining, c);
}
void orc_refracted_color(const orc_world *w, const orc_comps *comps, int remaining, orc_color *out, orc_counters *c) {
    *out = refracted_color_internal(w, comps, remaining, c);
}

/* ------------------------------------------------------------------ rendering/camera.rs */
void orc_camera_new(uint32_t hsize, uint32_t vsize, double field_of_view, orc_camera *out) { /* camera.rs:19-37 */
    double half_view = tan(field_of_view / 2.0);
    double aspect = (double)hsize / (double)vsize;
    double half_width, half_height;
    if (aspect > 1.0) { half_width = half_view; half_height = half_view / aspect; }
    else { half_width = half_view * aspect; half_height = half_view; }
    out->hsize = hsize; out->vsize = vsize; out->field_of_view = field_of_view;
    orc_mat4_identity(&out->transform);
    out->pixel_size = (half_width * 2.0) / (double)hsize;
    out->half_width = half_width; out->half_height = half_height;
}

static orc_ray ray_for_pixel_inv(const orc_camera *cam, const orc_mat4 *inverse_transform, uint32_t px, uint32_t py) {
    double x_offset = ((double)px + 0.5) * cam->pixel_size;
    double y_offset = ((double)py + 0.5) * cam->pixel_size;
    double world_x = cam->half_width - x_offset;
    double world_y = cam->half_height - y_offset;
    orc_vec3 pixel = mat_mul_point(inverse_transform, v3(world_x, world_y, -1.0));
    orc_vec3 origin = mat_mul_point(inverse_transform, v3(0.0, 0.0, 0.0));
    orc_ray r;
    r.origin = origin;
    r.direction = vnorm(vsub(pixel, origin));
    return r;
}
void orc_camera_ray_for_pixel(const orc_camera *cam, uint32_t px, uint32_t py, orc_ray *out) { /* camera.rs:39-54 */
    orc_mat4 inv;
    if (!orc_mat4_inverse(&cam->transform, &inv)) memset(&inv, 0xff, sizeof inv);
    *out = ray_for_pixel_inv(cam, &inv, px, py);
}

/* camera.rs:56-67 restricted to rows [y0, y1); world.rs:113-115 starts every pixel at depth 5 */
void orc_camera_render_rows(const orc_camera *cam, const orc_world *w, uint32_t y0, uint32_t y1, double *rgb,
                            orc_counters *c) {
    orc_mat4 inv;
    int inv_ok = orc_mat4_inverse(&cam->transform, &inv);
    if (!inv_ok) memset(&inv, 0xff, sizeof inv);
    for (uint32_t y = y0; y < y1 && y < cam->vsize; ++y)
        for (uint32_t x = 0; x < cam->hsize; ++x) {
            orc_ray ray;
            if (w->mode == 1) orc_camera_ray_for_pixel(cam, x, y, &ray); /* inverse per pixel, camera.rs:46-49 */
            else ray = ray_for_pixel_inv(cam, &inv, x, y);
            if (c) c->primary++;
            orc_color color = color_at_internal(w, ray, 5, c);
            size_t idx = ((size_t)y * cam->hsize + x) * 3; /* canvas.rs:40-42 */
            rgb[idx] = color.r; rgb[idx + 1] = color.g; rgb[idx + 2] = color.b;
        }
}

/* ------------------------------------------------------------------ rendering/canvas.rs */
uint32_t orc_quantize(double value) { /* canvas.rs:123-130: round half away from zero, saturating cast, clamp */
    double r = round(value * 255.0);
    uint32_t q;
    if (r != r || r <= 0.0) q = 0;
    else if (r >= 4294967295.0) q = 4294967295u;
    else q = (uint32_t)r;
    return q > 255u ? 255u : q;
}
void orc_canvas_quantize(const double *rgb, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)orc_quantize(rgb[i]);
}

typedef struct { char *out; size_t cap, len; } sbuf;
static void sb_put(sbuf *s, const char *p, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (s->len < s->cap) s->out[s->len] = p[i];
        s->len++;
    }
}
/* canvas.rs:44-90.  A row's tokens are joined by single spaces; when the joined row exceeds 70 characters it is
 * re-flowed greedily: a token moves to a new line when the running length, counting each token plus one space,
 * would exceed 70. */
size_t orc_canvas_to_ppm(const double *rgb, uint32_t width, uint32_t height, char *out, size_t cap) {
    sbuf s = {out, cap, 0};
    char tmp[64];
    int n = snprintf(tmp, sizeof tmp, "P3\n%u %u\n255\n", width, height);
    sb_put(&s, tmp, (size_t)n);
    size_t row_vals = (size_t)width * 3;
    for (uint32_t y = 0; y < height; ++y) {
        const double *row = rgb + (size_t)y * row_vals;
        size_t total = 0; /* length of the unsplit line */
        for (size_t i = 0; i < row_vals; ++i) {
            uint32_t q = orc_quantize(row[i]);
            total += (q >= 100 ? 3 : q >= 10 ? 2 : 1) + 1;
        }
        if (total) total -= 1;
        int split = total > 70;
        size_t line_length = 0;
        for (size_t i = 0; i < row_vals; ++i) {
            uint32_t q = orc_quantize(row[i]);
            int wl = snprintf(tmp, sizeof tmp, "%u", q);
            if (split) {
                line_length += (size_t)wl + 1;
                if (line_length > 70) {
                    sb_put(&s, "\n", 1);
                    line_length = (size_t)wl + 1;
                } else if (i) sb_put(&s, " ", 1);
            } else if (i) sb_put(&s, " ", 1);
            sb_put(&s, tmp, (size_t)wl);
        }
        sb_put(&s, "\n", 1);
    }
    sb_put(&s, "\n", 1);
    return s.len;
}

/* ------------------------------------------------------------------ scenes/loader.rs */
#define ORC_PI 3.14159265358979323846264338327950288 /* std::f64::consts::PI */

/* loader.rs:372-396 */
void orc_build_transform(const orc_transform_cfg *t, int present, orc_mat4 *out) {
    orc_mat4 matrix, m;
    orc_mat4_identity(&matrix);
    if (present) {
        if (t->has_translate) { orc_mat4_translate(t->translate[0], t->translate[1], t->translate[2], &m); orc_mat4_mul(&matrix, &m, &matrix); }
        if (t->has_rotate_z) { orc_mat4_rotate_z(t->rotate_z * ORC_PI / 180.0, &m); orc_mat4_mul(&matrix, &m, &matrix); }
        if (t->has_rotate_y) { orc_mat4_rotate_y(t->rotate_y * ORC_PI / 180.0, &m); orc_mat4_mul(&matrix, &m, &matrix); }
        if (t->has_rotate_x) { orc_mat4_rotate_x(t->rotate_x * ORC_PI / 180.0, &m); orc_mat4_mul(&matrix, &m, &matrix); }
        if (t->has_scale) { orc_mat4_scale(t->scale[0], t->scale[1], t->scale[2], &m); orc_mat4_mul(&matrix, &m, &matrix); }
    }
    *out = matrix;
}

/* loader.rs:196-212 */
int orc_build_camera(const orc_camera_cfg *cfg, orc_camera *out) {
    orc_camera_new(cfg->width, cfg->height, cfg->fov * ORC_PI / 180.0, out);
    orc_vec3 from = v3(cfg->from[0], cfg->from[1], cfg->from[2]);
    orc_vec3 to = v3(cfg->to[0], cfg->to[1], cfg->to[2]);
    orc_vec3 up = v3(cfg->up[0], cfg->up[1], cfg->up[2]);
    orc_view_transform(&from, &to, &up, &out->transform);
    return 0;
}

/* loader.rs:398-465 */
static int build_material(const orc_material_cfg *mc, orc_material *out) {
    if (mc->ambient < 0.0 || mc->diffuse < 0.0 || mc->specular < 0.0 || mc->shininess < 0.0) { /* materials.rs:94-96 */
        set_err("Material values must be positive");
        return -1;
    }
    orc_material_default(out);
    out->color = col(mc->color[0], mc->color[1], mc->color[2]);
    out->ambient = mc->ambient; out->diffuse = mc->diffuse; out->specular = mc->specular;
    out->shininess = mc->shininess; out->reflectivity = mc->reflectivity; out->transparency = mc->transparency;
    out->refractive_index = mc->refractive_index;
    if (mc->pattern_kind != ORC_PAT_NONE) {
        out->pattern.kind = mc->pattern_kind;
        out->pattern.a = col(mc->color_a[0], mc->color_a[1], mc->color_a[2]);
        out->pattern.b = col(mc->color_b[0], mc->color_b[1], mc->color_b[2]);
        orc_build_transform(&mc->pattern_transform, 1, &out->pattern.transform);
    }
    return 0;
}

/* loader.rs:255-370: returns the index after this object's subtree, or -1 */
static int build_object(orc_world *w, const orc_object_cfg *cfgs, int n, int idx, int parent,
                        const orc_material *parent_material) {
    if (idx >= n) { set_err("object list ends inside a group"); return -1; }
    const orc_object_cfg *cfg = &cfgs[idx];
    orc_mat4 transformation;
    orc_build_transform(&cfg->transform, cfg->has_transform, &transformation);
    orc_material mat;
    if (cfg->has_material) { if (build_material(&cfg->material, &mat)) return -1; }
    else if (parent_material) mat = *parent_material;
    else orc_material_default(&mat);
    double minimum = cfg->has_minimum ? cfg->minimum : -INFINITY;
    double maximum = cfg->has_maximum ? cfg->maximum : INFINITY;
    int closed = cfg->has_closed ? cfg->closed : 0;
    int id = orc_world_add(w, parent, cfg->kind, &transformation, &mat, minimum, maximum, closed);
    if (id < 0) return -1;
    int next = idx + 1;
    if (cfg->kind == ORC_GROUP)
        for (int k = 0; k < cfg->n_children; ++k) {
            next = build_object(w, cfgs, n, next, id, &mat);
            if (next < 0) return -1;
        }
    return next;
}

/* loader.rs:214-247 */
orc_world *orc_build_world(const orc_object_cfg *objects, int n_objects, const double lp[3], const double li[3]) {
    orc_world *w = orc_world_new();
    if (!w) return NULL;
    orc_vec3 p = v3(lp[0], lp[1], lp[2]);
    orc_color i = col(li[0], li[1], li[2]);
    orc_world_set_light(w, &p, &i);
    int idx = 0;
    while (idx < n_objects) {
        idx = build_object(w, objects, n_objects, idx, -1, NULL);
        if (idx < 0) { orc_world_free(w); return NULL; }
    }
    if (orc_world_finalize(w)) { orc_world_free(w); return NULL; }
    return w;
}
