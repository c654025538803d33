// Host-side f64 math of the render path: Matrix4 / Point / Vector / Color with the reference's operation order
// (core/matrices.rs, core/tuples.rs, core/color.rs, core/floats.rs of the reference checkout).
//
// Everything the device consumes that is a pure function of the scene (inverses, transposes, products, bounds)
// is computed here once, with the same algorithms the reference re-runs per ray, so the device sees the very
// matrices the reference would.  Compile without FMA contraction (-ffp-contract=off).
#pragma once

#include <cmath>
#include <cstring>
#include <optional>

namespace rt {

constexpr double EPSILON = 1e-5;  // core/floats.rs:2

inline bool float_equal(double a, double b) { return std::fabs(a - b) < EPSILON; }  // core/floats.rs:4-6

struct Vec3 {  // Point (w = 1) or Vector (w = 0): core/tuples.rs:14-19,136-141
    double x = 0.0, y = 0.0, z = 0.0;
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // tuples.rs:59-61
inline double magnitude(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // tuples.rs:35-37
inline Vec3 normalize(Vec3 a) { return a / magnitude(a); }                                // tuples.rs:40-42
inline Vec3 cross(Vec3 a, Vec3 b) {                                                       // tuples.rs:45-51
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Matrix4 {  // core/matrices.rs:7-9, row-major
    double d[4][4];

    static Matrix4 zero() {
        Matrix4 m;
        std::memset(m.d, 0, sizeof m.d);
        return m;
    }
    static Matrix4 identity() {  // matrices.rs:17-26
        Matrix4 m = zero();
        m.d[0][0] = m.d[1][1] = m.d[2][2] = m.d[3][3] = 1.0;
        return m;
    }
    static Matrix4 from_array(const double a[16]) {
        Matrix4 m;
        std::memcpy(m.d, a, sizeof m.d);
        return m;
    }
    void to_array(double a[16]) const { std::memcpy(a, d, sizeof d); }

    static Matrix4 translate(double x, double y, double z) {  // matrices.rs:28-37
        Matrix4 m = identity();
        m.d[0][3] = x;
        m.d[1][3] = y;
        m.d[2][3] = z;
        return m;
    }
    static Matrix4 scale(double x, double y, double z) {  // matrices.rs:39-48
        Matrix4 m = identity();
        m.d[0][0] = x;
        m.d[1][1] = y;
        m.d[2][2] = z;
        return m;
    }
    static Matrix4 rotate_x(double r) {  // matrices.rs:50-59
        Matrix4 m = identity();
        m.d[1][1] = std::cos(r);
        m.d[1][2] = -std::sin(r);
        m.d[2][1] = std::sin(r);
        m.d[2][2] = std::cos(r);
        return m;
    }
    static Matrix4 rotate_y(double r) {  // matrices.rs:61-70
        Matrix4 m = identity();
        m.d[0][0] = std::cos(r);
        m.d[0][2] = std::sin(r);
        m.d[2][0] = -std::sin(r);
        m.d[2][2] = std::cos(r);
        return m;
    }
    static Matrix4 rotate_z(double r) {  // matrices.rs:72-81
        Matrix4 m = identity();
        m.d[0][0] = std::cos(r);
        m.d[0][1] = -std::sin(r);
        m.d[1][0] = std::sin(r);
        m.d[1][1] = std::cos(r);
        return m;
    }

    Matrix4 transpose() const {  // matrices.rs:137-145
        Matrix4 m;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) m.d[c][r] = d[r][c];
        return m;
    }

    // matrices.rs:160-162 with Matrix3::determinant (259-265) and Matrix2::determinant (310-312) expanded in place:
    // every level is a first-row cofactor expansion accumulated from 0.0, left to right.
    double minor(int row, int col) const {
        double s[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) s[i][j] = d[i + (i >= row ? 1 : 0)][j + (j >= col ? 1 : 0)];
        double result = 0.0;
        for (int c = 0; c < 3; ++c) {
            double t[2][2];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) t[i][j] = s[i + 1][j + (j >= c ? 1 : 0)];
            double m2 = t[0][0] * t[1][1] - t[0][1] * t[1][0];
            double cof = (c % 2 == 0) ? m2 : -m2;
            result += s[0][c] * cof;
        }
        return result;
    }
    double cofactor(int row, int col) const {  // matrices.rs:151-158
        double m = minor(row, col);
        return ((row + col) % 2 == 0) ? m : -m;
    }
    double determinant() const {  // matrices.rs:94-100
        double result = 0.0;
        for (int c = 0; c < 4; ++c) result += d[0][c] * cofactor(0, c);
        return result;
    }
    std::optional<Matrix4> inverse() const {  // matrices.rs:116-134
        double det = determinant();
        if (float_equal(det, 0.0)) return std::nullopt;
        Matrix4 m;
        double inv_det = 1.0 / det;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) m.d[c][r] = cofactor(r, c) * inv_det;
        return m;
    }
};

inline Matrix4 operator*(const Matrix4 &a, const Matrix4 &b) {  // matrices.rs:178-192
    Matrix4 m;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double sum = 0.0;
            for (int i = 0; i < 4; ++i) sum += a.d[r][i] * b.d[i][c];
            m.d[r][c] = sum;
        }
    return m;
}

// matrices.rs:195-221: each output component is row[0]*x + row[1]*y + row[2]*z + row[3]*w
inline Vec3 mul_point(const Matrix4 &m, Vec3 p) {
    Vec3 o;
    o.x = m.d[0][0] * p.x + m.d[0][1] * p.y + m.d[0][2] * p.z + m.d[0][3] * 1.0;
    o.y = m.d[1][0] * p.x + m.d[1][1] * p.y + m.d[1][2] * p.z + m.d[1][3] * 1.0;
    o.z = m.d[2][0] * p.x + m.d[2][1] * p.y + m.d[2][2] * p.z + m.d[2][3] * 1.0;
    return o;
}
inline Vec3 mul_vector(const Matrix4 &m, Vec3 v) {
    Vec3 o;
    o.x = m.d[0][0] * v.x + m.d[0][1] * v.y + m.d[0][2] * v.z + m.d[0][3] * 0.0;
    o.y = m.d[1][0] * v.x + m.d[1][1] * v.y + m.d[1][2] * v.z + m.d[1][3] * 0.0;
    o.z = m.d[2][0] * v.x + m.d[2][1] * v.y + m.d[2][2] * v.z + m.d[2][3] * 0.0;
    return o;
}

// scene/transformations.rs:4-17
inline Matrix4 view_transform(Vec3 from, Vec3 to, Vec3 up) {
    Vec3 forward = normalize(to - from);
    Vec3 up_normalized = normalize(up);
    Vec3 left = cross(forward, up_normalized);
    Vec3 true_up = cross(left, forward);
    Matrix4 o = Matrix4::identity();
    o.d[0][0] = left.x; o.d[0][1] = left.y; o.d[0][2] = left.z;
    o.d[1][0] = true_up.x; o.d[1][1] = true_up.y; o.d[1][2] = true_up.z;
    o.d[2][0] = -forward.x; o.d[2][1] = -forward.y; o.d[2][2] = -forward.z;
    return o * Matrix4::translate(-from.x, -from.y, -from.z);
}

struct Bounds {  // geometry/bounds.rs:7-10
    Vec3 min, max;

    static Bounds empty() {  // bounds.rs:20-25
        return {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    }
    static Bounds infinite() {  // bounds.rs:28-33
        return {{-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
    }
    void add_point(Vec3 p) {  // bounds.rs:36-44; f64::min/max ignore NaN like fmin/fmax
        min.x = std::fmin(min.x, p.x); min.y = std::fmin(min.y, p.y); min.z = std::fmin(min.z, p.z);
        max.x = std::fmax(max.x, p.x); max.y = std::fmax(max.y, p.y); max.z = std::fmax(max.z, p.z);
    }
    Bounds merge(const Bounds &o) const {  // bounds.rs:47-52
        Bounds r = *this;
        r.add_point(o.min);
        r.add_point(o.max);
        return r;
    }
    Bounds transform(const Matrix4 &m) const {  // bounds.rs:57-89
        bool inf_min = std::isinf(min.x) || std::isinf(min.y) || std::isinf(min.z);
        bool inf_max = std::isinf(max.x) || std::isinf(max.y) || std::isinf(max.z);
        if (inf_min || inf_max) return infinite();
        const Vec3 corners[8] = {min,
                                 {min.x, min.y, max.z}, {min.x, max.y, min.z}, {min.x, max.y, max.z},
                                 {max.x, min.y, min.z}, {max.x, min.y, max.z}, {max.x, max.y, min.z},
                                 max};
        Bounds r = empty();
        for (const Vec3 &c : corners) r.add_point(mul_point(m, c));
        return r;
    }
};

}  // namespace rt
