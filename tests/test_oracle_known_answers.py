"""Pins the CPU oracle with the known answers of the reference's own inline unit tests (tolerance 1e-5 =
core/floats.rs:2 where the reference compares through float_equal, exact where it uses assert_eq on f64).
Each test names the reference test module it transcribes."""
import math

import pytest

EPS = 1e-5
S2 = math.sqrt(2.0)
S3 = math.sqrt(3.0)


def close(a, b, eps=EPS):
    return all(abs(x - y) < eps for x, y in zip(a, b))


def mat_rows(orc, rows):
    return orc.Mat4.from_rows(rows)


# ---------------------------------------------------------------- core/matrices.rs tests
def test_matrix_determinants_and_cofactors(orc):
    L = orc.lib()
    m = mat_rows(orc, [[-2, -8, 3, 5], [-3, 1, 7, 3], [1, 2, -9, 6], [-6, 7, 7, -9]])  # matrices.rs:517-530
    assert L.orc_mat4_cofactor(m, 0, 0) == 690.0
    assert L.orc_mat4_cofactor(m, 0, 1) == 447.0
    assert L.orc_mat4_cofactor(m, 0, 2) == 210.0
    assert L.orc_mat4_cofactor(m, 0, 3) == 51.0
    assert L.orc_mat4_determinant(m) == -4071.0
    inv = mat_rows(orc, [[6, 4, 4, 4], [5, 5, 7, 6], [4, -9, 3, -7], [9, 1, 7, -6]])  # matrices.rs:573-582
    assert L.orc_mat4_determinant(inv) == -2120.0
    sing = mat_rows(orc, [[-4, 2, -2, -3], [9, 6, 2, 6], [0, -5, 1, -5], [0, 0, 0, 0]])  # matrices.rs:585-594
    assert L.orc_mat4_determinant(sing) == 0.0
    assert orc.inverse(sing) is None


def test_matrix_inverse_known_answers(orc):
    L = orc.lib()
    m = mat_rows(orc, [[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]])  # matrices.rs:597-645
    assert L.orc_mat4_determinant(m) == 532.0
    assert L.orc_mat4_cofactor(m, 2, 3) == -160.0
    assert L.orc_mat4_cofactor(m, 3, 2) == 105.0
    inv = orc.inverse(m)
    assert inv.m[3][2] == -160.0 / 532.0
    assert inv.m[2][3] == 105.0 / 532.0
    expected = [[0.21804511278195488, 0.45112781954887216, 0.24060150375939848, -0.045112781954887216],
                [-0.8082706766917294, -1.4567669172932332, -0.44360902255639095, 0.5206766917293233],
                [-0.07894736842105263, -0.2236842105263158, -0.05263157894736842, 0.19736842105263158],
                [-0.5225563909774437, -0.8139097744360902, -0.3007518796992481, 0.30639097744360905]]
    assert all(close(a, b) for a, b in zip(inv.rows(), expected))
    m2 = mat_rows(orc, [[8, -5, 9, 2], [7, 5, 6, 1], [-6, 0, 9, 6], [-3, 0, -9, -4]])  # matrices.rs:648-684
    e2 = [[-0.15384615384615385, -0.15384615384615385, -0.28205128205128205, -0.5384615384615384],
          [-0.07692307692307693, 0.12307692307692308, 0.02564102564102564, 0.03076923076923077],
          [0.358974358974359, 0.358974358974359, 0.4358974358974359, 0.9230769230769231],
          [-0.6923076923076923, -0.6923076923076923, -0.7692307692307693, -1.9230769230769231]]
    assert all(close(a, b) for a, b in zip(orc.inverse(m2).rows(), e2))
    m3 = mat_rows(orc, [[9, 3, 0, 9], [-5, -2, -6, -3], [-4, 9, 6, 4], [-7, 6, 6, 2]])  # matrices.rs:687-723
    e3 = [[-0.040740740740740744, -0.07777777777777778, 0.14444444444444443, -0.2222222222222222],
          [-0.07777777777777778, 0.03333333333333333, 0.36666666666666664, -0.3333333333333333],
          [-0.029012345679012345, -0.14629629629629629, -0.10925925925925926, 0.12962962962962962],
          [0.17777777777777778, 0.06666666666666667, -0.26666666666666666, 0.3333333333333333]]
    assert all(close(a, b) for a, b in zip(orc.inverse(m3).rows(), e3))


def test_matrix_product_and_inverse_round_trip(orc):
    a = mat_rows(orc, [[3, -9, 7, 3], [3, -8, 2, -9], [-4, 4, 4, 1], [-6, 5, -1, 1]])  # matrices.rs:726-744
    b = mat_rows(orc, [[8, 2, 2, 2], [3, -1, 7, 0], [7, 0, 5, 4], [6, -2, 0, 5]])
    c = orc.mul(a, b)
    back = orc.mul(c, orc.inverse(b))
    assert orc.lib().orc_mat4_equal(a, back)
    t = orc.Mat4()
    m = mat_rows(orc, [[0, 9, 3, 0], [9, 8, 0, 8], [1, 8, 5, 3], [0, 0, 5, 8]])  # matrices.rs:481-497
    orc.lib().orc_mat4_transpose(m, t)
    assert t.rows() == [[0, 9, 1, 0], [9, 8, 8, 0], [3, 0, 5, 5], [0, 8, 3, 8]]


def _mul_point(orc, m, p):
    out = orc.Vec3()
    orc.lib().orc_mat4_mul_point(m, orc.vec(*p), out)
    return out.tuple()


def _mul_vector(orc, m, v):
    out = orc.Vec3()
    orc.lib().orc_mat4_mul_vector(m, orc.vec(*v), out)
    return out.tuple()


def test_transformation_matrices(orc):
    # matrices.rs:747-946 (the book's chapter 4 scenarios)
    assert _mul_point(orc, orc.translate(5, -3, 2), (-3, 4, 5)) == (2, 1, 7)
    assert _mul_point(orc, orc.inverse(orc.translate(5, -3, 2)), (-3, 4, 5)) == (-8, 7, 3)
    assert _mul_vector(orc, orc.translate(5, -3, 2), (-3, 4, 5)) == (-3, 4, 5)
    assert _mul_point(orc, orc.scale(2, 3, 4), (-4, 6, 8)) == (-8, 18, 32)
    assert close(_mul_vector(orc, orc.inverse(orc.scale(2, 3, 4)), (-4, 6, 8)), (-2, 2, 2))
    assert _mul_point(orc, orc.scale(-1, 1, 1), (2, 3, 4)) == (-2, 3, 4)
    assert close(_mul_point(orc, orc.rotate_x(math.pi / 4), (0, 1, 0)), (0, S2 / 2, S2 / 2))
    assert close(_mul_point(orc, orc.rotate_x(math.pi / 2), (0, 1, 0)), (0, 0, 1))
    assert close(_mul_point(orc, orc.inverse(orc.rotate_x(math.pi / 4)), (0, 1, 0)), (0, S2 / 2, -S2 / 2))
    assert close(_mul_point(orc, orc.rotate_y(math.pi / 4), (0, 0, 1)), (S2 / 2, 0, S2 / 2))
    assert close(_mul_point(orc, orc.rotate_y(math.pi / 2), (0, 0, 1)), (1, 0, 0))
    assert close(_mul_point(orc, orc.rotate_z(math.pi / 4), (0, 1, 0)), (-S2 / 2, S2 / 2, 0))
    assert close(_mul_point(orc, orc.rotate_z(math.pi / 2), (0, 1, 0)), (-1, 0, 0))
    assert _mul_point(orc, orc.shear(1, 0, 0, 0, 0, 0), (2, 3, 4)) == (5, 3, 4)
    assert _mul_point(orc, orc.shear(0, 1, 0, 0, 0, 0), (2, 3, 4)) == (6, 3, 4)
    assert _mul_point(orc, orc.shear(0, 0, 1, 0, 0, 0), (2, 3, 4)) == (2, 5, 4)
    assert _mul_point(orc, orc.shear(0, 0, 0, 1, 0, 0), (2, 3, 4)) == (2, 7, 4)
    assert _mul_point(orc, orc.shear(0, 0, 0, 0, 1, 0), (2, 3, 4)) == (2, 3, 6)
    assert _mul_point(orc, orc.shear(0, 0, 0, 0, 0, 1), (2, 3, 4)) == (2, 3, 7)
    chained = orc.mul(orc.mul(orc.translate(10, 5, 7), orc.scale(5, 5, 5)), orc.rotate_x(math.pi / 2))
    assert close(_mul_point(orc, chained, (1, 0, 1)), (15, 0, 7))


# ---------------------------------------------------------------- core/tuples.rs tests
def test_vector_operations(orc):
    L = orc.lib()
    assert L.orc_vec_magnitude(orc.vec(1, 2, 3)) == math.sqrt(14.0)
    out = orc.Vec3()
    L.orc_vec_normalize(orc.vec(1, 2, 3), out)
    assert close(out.tuple(), (0.26726, 0.53452, 0.80178))
    assert L.orc_vec_dot(orc.vec(1, 2, 3), orc.vec(2, 3, 4)) == 20.0
    L.orc_vec_cross(orc.vec(1, 2, 3), orc.vec(2, 3, 4), out)
    assert out.tuple() == (-1, 2, -1)
    L.orc_vec_reflect(orc.vec(1, -1, 0), orc.vec(0, 1, 0), out)
    assert out.tuple() == (1, 1, 0)
    L.orc_vec_reflect(orc.vec(0, -1, 0), orc.vec(S2 / 2, S2 / 2, 0), out)
    assert close(out.tuple(), (1, 0, 0))


# ---------------------------------------------------------------- scene/transformations.rs tests
def test_view_transform(orc):
    def vt(f, t, u):
        m = orc.Mat4()
        orc.lib().orc_view_transform(orc.vec(*f), orc.vec(*t), orc.vec(*u), m)
        return m

    L = orc.lib()
    assert L.orc_mat4_equal(vt((0, 0, 0), (0, 0, -1), (0, 1, 0)), orc.identity())
    assert L.orc_mat4_equal(vt((0, 0, 0), (0, 0, 1), (0, 1, 0)), orc.scale(-1, 1, -1))
    assert L.orc_mat4_equal(vt((0, 0, 8), (0, 0, 0), (0, 1, 0)), orc.translate(0, 0, -8))
    e = [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
         [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]]
    assert L.orc_mat4_equal(vt((1, 3, 2), (4, -2, 8), (1, 1, 0)), mat_rows(orc, e))


# ---------------------------------------------------------------- rendering/rays.rs tests
def test_ray_position_and_transform(orc):
    L = orc.lib()
    r = orc.ray((2, 3, 4), (1, 0, 0))
    out = orc.Vec3()
    for t, e in ((0, (2, 3, 4)), (1, (3, 3, 4)), (-1, (1, 3, 4)), (2.5, (4.5, 3, 4))):
        L.orc_ray_position(r, t, out)
        assert out.tuple() == e
    r2 = orc.Ray()
    L.orc_ray_transform(orc.ray((1, 2, 3), (0, 1, 0)), orc.translate(3, 4, 5), r2)
    assert r2.origin.tuple() == (4, 6, 8) and r2.direction.tuple() == (0, 1, 0)
    L.orc_ray_transform(orc.ray((1, 2, 3), (0, 1, 0)), orc.scale(2, 3, 4), r2)
    assert r2.origin.tuple() == (2, 6, 12) and r2.direction.tuple() == (0, 3, 0)


# ---------------------------------------------------------------- geometry/*.rs tests
def _xs(orc, kind, origin, direction, minimum=-math.inf, maximum=math.inf, closed=False, normalize=False):
    d = direction
    if normalize:
        n = math.sqrt(sum(c * c for c in d))
        d = tuple(c / n for c in d)
    ts = (orc.C.c_double * 4)()
    n = orc.lib().orc_local_intersect(kind, minimum, maximum, int(closed), orc.ray(origin, d), ts)
    return [ts[i] for i in range(n)]


def _normal(orc, kind, p, minimum=-math.inf, maximum=math.inf):
    out = orc.Vec3()
    orc.lib().orc_local_normal_at(kind, minimum, maximum, orc.vec(*p), out)
    return out.tuple()


def test_sphere(orc):  # sphere.rs:101-198
    assert _xs(orc, orc.SPHERE, (0, 0, -5), (0, 0, 1)) == [4.0, 6.0]
    assert _xs(orc, orc.SPHERE, (0, 1, -5), (0, 0, 1)) == [5.0, 5.0]
    assert _xs(orc, orc.SPHERE, (0, 2, -0.5), (0, 0, 1)) == []
    assert _xs(orc, orc.SPHERE, (0, 0, 0), (0, 0, 1)) == [-1.0, 1.0]
    assert _xs(orc, orc.SPHERE, (0, 0, 5), (0, 0, 1)) == [-6.0, -4.0]
    assert _normal(orc, orc.SPHERE, (1, 0, 0)) == (1, 0, 0)
    assert _normal(orc, orc.SPHERE, (0, 1, 0)) == (0, 1, 0)
    assert _normal(orc, orc.SPHERE, (0, 0, 1)) == (0, 0, 1)
    assert close(_normal(orc, orc.SPHERE, (S3 / 3, S3 / 3, S3 / 3)), (S3 / 3, S3 / 3, S3 / 3))


def test_plane(orc):  # planes.rs:84-126
    assert _normal(orc, orc.PLANE, (0, 0, 0)) == (0, 1, 0)
    assert _normal(orc, orc.PLANE, (10, 0, -10)) == (0, 1, 0)
    assert _xs(orc, orc.PLANE, (0, 10, 0), (0, 0, 1)) == []
    assert _xs(orc, orc.PLANE, (0, 0, 0), (0, 0, 1)) == []
    assert _xs(orc, orc.PLANE, (0, 1, 0), (0, -1, 0)) == [1.0]
    assert _xs(orc, orc.PLANE, (0, -1, 0), (0, 1, 0)) == [1.0]


def test_cube(orc):  # cubes.rs:121-279
    hits = [((5, 0.5, 0), (-1, 0, 0), 4, 6), ((-5, 0.5, 0), (1, 0, 0), 4, 6), ((0.5, 5, 0), (0, -1, 0), 4, 6),
            ((0.5, -5, 0), (0, 1, 0), 4, 6), ((0.5, 0, 5), (0, 0, -1), 4, 6), ((0.5, 0, -5), (0, 0, 1), 4, 6),
            ((0, 0.5, 0), (0, 0, 1), -1, 1)]
    for o, d, t1, t2 in hits:
        assert _xs(orc, orc.CUBE, o, d) == [t1, t2]
    misses = [((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8108, 0.2673, 0.5345)),
              ((0, 0, -2), (0.5345, 0.8018, 0.2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)),
              ((2, 2, 0), (-1, 0, 0))]
    for o, d in misses:
        assert _xs(orc, orc.CUBE, o, d) == []
    normals = [((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, 0.9), (-1, 0, 0)), ((-0.4, 1, -0.1), (0, 1, 0)),
               ((0.3, -1, -0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
               ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
    for p, n in normals:
        assert _normal(orc, orc.CUBE, p) == n


def test_cylinder(orc):  # cylinders.rs:165-483
    for o, d in (((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))):
        assert _xs(orc, orc.CYLINDER, o, d, normalize=True) == []
    for o, d, t0, t1 in (((1, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (0, 0, 1), 4, 6),
                         ((0.5, 0, -5), (0.1, 1, 1), 6.80798, 7.08872)):
        xs = _xs(orc, orc.CYLINDER, o, d, normalize=True)
        assert len(xs) == 2 and close(xs, (t0, t1))
    trunc = [((0, 1.5, 0), (0.1, 1, 0), 0), ((0, 3, -5), (0, 0, 1), 0), ((0, 0, -5), (0, 0, 1), 0),
             ((0, 2, -5), (0, 0, 1), 0), ((0, 1, -5), (0, 0, 1), 0), ((0, 1.5, -2), (0, 0, 1), 2)]
    for o, d, count in trunc:
        assert len(_xs(orc, orc.CYLINDER, o, d, 1.0, 2.0, False, normalize=True)) == count
    caps = [((0, 3, 0), (0, -1, 0)), ((0, 3, -2), (0, -1, 2)), ((0, 4, -2), (0, -1, 1)), ((0, 0, -2), (0, 1, 2)),
            ((0, -1, -2), (0, 1, 1))]
    for o, d in caps:
        assert len(_xs(orc, orc.CYLINDER, o, d, 1.0, 2.0, True, normalize=True)) == 2
    for p, n in (((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, -2, 1), (0, 0, 1)), ((-1, 1, 0), (-1, 0, 0))):
        assert _normal(orc, orc.CYLINDER, p) == n
    for p, n in (((0, 1, 0), (0, -1, 0)), ((0.5, 1, 0), (0, -1, 0)), ((0, 1, 0.5), (0, -1, 0)),
                 ((0, 2, 0), (0, 1, 0)), ((0.5, 2, 0), (0, 1, 0)), ((0, 2, 0.5), (0, 1, 0))):
        assert _normal(orc, orc.CYLINDER, p, 1.0, 2.0) == n


def test_cone(orc):  # cones.rs:175-329
    for o, d, t0, t1 in (((0, 0, -5), (0, 0, 1), 5, 5), ((0, 0, -5), (1, 1, 1), 8.66025, 8.66025),
                         ((1, 1, -5), (-0.5, -1, 1), 4.55006, 49.44994)):
        xs = _xs(orc, orc.CONE, o, d, normalize=True)
        assert len(xs) == 2 and close(xs, (t0, t1))
    xs = _xs(orc, orc.CONE, (0, 0, -1), (0, 1, 1), normalize=True)
    assert len(xs) == 1 and abs(xs[0] - 0.35355) < EPS
    for o, d, count in (((0, 0, -5), (0, 1, 0), 0), ((0, 0, -0.25), (0, 1, 1), 2), ((0, 0, -0.25), (0, 1, 0), 4)):
        assert len(_xs(orc, orc.CONE, o, d, -0.5, 0.5, True, normalize=True)) == count
    assert _normal(orc, orc.CONE, (0, 0, 0)) == (0, 0, 0)
    assert close(_normal(orc, orc.CONE, (1, 1, 1)), (1, -S2, 1))
    assert close(_normal(orc, orc.CONE, (-1, -1, 0)), (-1, 1, 0))


def _bounds(orc, lo, hi):
    b = orc.Bounds()
    b.min, b.max = orc.vec(*lo), orc.vec(*hi)
    return b


def test_bounds(orc):  # bounds.rs:137-341
    L = orc.lib()
    b = orc.Bounds()
    L.orc_bounds_empty(b)
    assert b.min.x == math.inf and b.max.x == -math.inf
    L.orc_bounds_add_point(b, orc.vec(-5, 2, 0))
    L.orc_bounds_add_point(b, orc.vec(7, 0, -3))
    assert b.min.tuple() == (-5, 0, -3) and b.max.tuple() == (7, 2, 0)
    m = orc.Bounds()
    L.orc_bounds_merge(_bounds(orc, (-5, -2, 0), (7, 4, 4)), _bounds(orc, (8, -7, -2), (14, 2, 8)), m)
    assert m.min.tuple() == (-5, -7, -2) and m.max.tuple() == (14, 4, 8)
    unit = _bounds(orc, (-1, -1, -1), (1, 1, 1))
    assert L.orc_bounds_intersects(unit, orc.ray((5, 0.5, 0), (-1, 0, 0)))
    assert L.orc_bounds_intersects(unit, orc.ray((-5, 0.5, 0), (1, 0, 0)))
    assert not L.orc_bounds_intersects(unit, orc.ray((0, 5, 0), (1, 0, 0)))
    box = _bounds(orc, (5, -2, 0), (11, 4, 7))
    assert L.orc_bounds_intersects(box, orc.ray((15, 1, 2), (-1, 0, 0)))
    assert L.orc_bounds_intersects(box, orc.ray((15, -1, 2), (-1, 0, 0)))
    assert not L.orc_bounds_intersects(box, orc.ray((15, -3, 2), (-1, 0, 0)))
    t = orc.Bounds()
    L.orc_bounds_transform(unit, orc.mul(orc.rotate_x(math.pi / 4), orc.rotate_y(math.pi / 4)), t)
    assert t.min.x <= -1.0 and t.max.x >= 1.0
    L.orc_bounds_transform(unit, orc.translate(5, 3, 2), t)
    assert t.min.tuple() == (4, 2, 1) and t.max.tuple() == (6, 4, 3)
    L.orc_bounds_transform(unit, orc.scale(2, 3, 4), t)
    assert t.min.tuple() == (-2, -3, -4) and t.max.tuple() == (2, 3, 4)
    for kind in (orc.SPHERE, orc.CUBE):
        L.orc_shape_bounds(kind, -math.inf, math.inf, t)
        assert t.min.tuple() == (-1, -1, -1) and t.max.tuple() == (1, 1, 1)
    L.orc_shape_bounds(orc.CYLINDER, -math.inf, math.inf, t)
    assert (t.min.x, t.max.x, t.min.z, t.max.z) == (-1, 1, -1, 1) and t.min.y == -math.inf and t.max.y == math.inf
    L.orc_shape_bounds(orc.CYLINDER, -5.0, 3.0, t)
    assert t.min.tuple() == (-1, -5, -1) and t.max.tuple() == (1, 3, 1)
    L.orc_shape_bounds(orc.CONE, -math.inf, math.inf, t)
    assert t.min.tuple() == (-math.inf,) * 3 and t.max.tuple() == (math.inf,) * 3
    L.orc_shape_bounds(orc.CONE, -5.0, 3.0, t)
    assert t.min.tuple() == (-5, -5, -5) and t.max.tuple() == (5, 3, 5)
    L.orc_shape_bounds(orc.PLANE, 0, 0, t)
    assert t.min.tuple() == (-math.inf, 0, -math.inf) and t.max.tuple() == (math.inf, 0, math.inf)
    # an infinite box stays infinite under any transform (bounds.rs:57-69)
    L.orc_bounds_transform(t, orc.rotate_x(0.3), m)
    assert m.min.tuple() == (-math.inf,) * 3 and m.max.tuple() == (math.inf,) * 3


# ---------------------------------------------------------------- geometry/groups.rs tests
def test_group_intersections(orc):
    w = orc.World()
    g = w.add(orc.GROUP)
    assert w.intersect(orc.ray((0, 0, 0), (0, 0, 1))) == []  # groups.rs:233-241
    w = orc.World()
    g = w.add(orc.GROUP)
    s1 = w.add(orc.SPHERE, parent=g)
    s2 = w.add(orc.SPHERE, orc.translate(0, 0, -3), parent=g)
    w.add(orc.SPHERE, orc.translate(5, 0, 0), parent=g)
    xs = w.intersect(orc.ray((0, 0, -5), (0, 0, 1)))  # groups.rs:244-273
    assert [t for t, _ in xs] == [1.0, 3.0, 4.0, 6.0]
    assert [i for _, i in xs] == [s2, s2, s1, s1]
    w = orc.World()
    g = w.add(orc.GROUP, orc.scale(2, 2, 2))
    w.add(orc.SPHERE, orc.translate(5, 0, 0), parent=g)
    assert len(w.intersect(orc.ray((10, 0, -10), (0, 0, 1)))) == 2  # groups.rs:276-293


def _world_to_object(orc, w, oid, p):
    return _mul_point(orc, orc.inverse(w.world_transform(oid)), p)


def test_group_nested_transforms_and_normals(orc):
    w = orc.World()
    g1 = w.add(orc.GROUP, orc.rotate_y(math.pi / 2))
    g2 = w.add(orc.GROUP, orc.scale(2, 2, 2), parent=g1)
    s = w.add(orc.SPHERE, orc.translate(5, 0, 0), parent=g2)
    assert close(_world_to_object(orc, w, s, (-2, 0, -10)), (0, 0, -1))  # groups.rs:296-317
    w = orc.World()
    g1 = w.add(orc.GROUP, orc.rotate_y(math.pi / 2))
    g2 = w.add(orc.GROUP, orc.scale(1, 2, 3), parent=g1)
    s = w.add(orc.SPHERE, orc.translate(5, 0, 0), parent=g2)
    n = w.normal_at(s, (1.7321, 1.1547, -5.5774))  # groups.rs:351-376
    assert close(n, (0.2857, 0.4286, -0.8571), 1e-4)
    # translations accumulate down the hierarchy (groups.rs:636-692)
    w = orc.World()
    g1 = w.add(orc.GROUP, orc.translate(10, 0, 0))
    g2 = w.add(orc.GROUP, orc.translate(1, 0, 0), parent=g1)
    s = w.add(orc.SPHERE, orc.translate(0.5, 0, 0), parent=g2)
    assert close(_world_to_object(orc, w, s, (11.5, 0, 0)), (0, 0, 0))


def test_group_bounds(orc):
    w = orc.World()
    g = w.add(orc.GROUP)
    b = w.group_bounds(g)  # groups.rs:379-393
    assert b.min.tuple() == (math.inf,) * 3 and b.max.tuple() == (-math.inf,) * 3
    w.add(orc.SPHERE, parent=g)
    w.add(orc.SPHERE, orc.translate(2, 0, 0), parent=g)
    b = w.group_bounds(g)  # groups.rs:396-424
    assert b.min.tuple() == (-1, -1, -1) and b.max.tuple() == (3, 1, 1)
    w = orc.World()
    g = w.add(orc.GROUP, orc.scale(2, 2, 2))
    w.add(orc.SPHERE, parent=g)
    b = w.group_bounds(g)  # groups.rs:427-449: bounds live in the group's own space
    assert b.min.tuple() == (-1, -1, -1) and b.max.tuple() == (1, 1, 1)
    w = orc.World()
    g1 = w.add(orc.GROUP)
    g2 = w.add(orc.GROUP, orc.translate(2, 0, 0), parent=g1)
    w.add(orc.SPHERE, parent=g2)
    b = w.group_bounds(g1)  # groups.rs:489-520
    assert b.min.tuple() == (1, -1, -1) and b.max.tuple() == (3, 1, 1)
    # culling (groups.rs:452-486)
    w = orc.World()
    g = w.add(orc.GROUP)
    w.add(orc.SPHERE, orc.translate(0, 0, -5), parent=g)
    assert w.intersect(orc.ray((10, 10, 10), (1, 0, 0))) == []
    c = orc.Counters()
    w.intersect(orc.ray((10, 10, 10), (1, 0, 0)), c)
    assert c.aabb_tests == 1 and sum(c.prim_tests) == 0
    w = orc.World()
    g = w.add(orc.GROUP)
    s = w.add(orc.SPHERE, parent=g)
    xs = w.intersect(orc.ray((0, 0, -5), (0, 0, 1)))
    assert len(xs) == 2 and xs[0][1] == s  # groups.rs:523-553: hits reference the child
    assert w.normal_at(s, (0, 0, -1)) == (0, 0, -1)


# ---------------------------------------------------------------- rendering/intersections.rs tests
def test_prepare_computations(orc):
    w = orc.World()
    s = w.add(orc.SPHERE)
    c = w.prepare_computations(s, 4.0, orc.ray((0, 0, -5), (0, 0, 1)))  # intersections.rs:226-244
    assert c.time == 4.0 and c.object == s and not c.inside
    assert c.point.tuple() == (0, 0, -1) and c.eye_vector.tuple() == (0, 0, -1)
    c = w.prepare_computations(s, 1.0, orc.ray((0, 0, 0), (0, 0, 1)))  # intersections.rs:247-256
    assert c.point.tuple() == (0, 0, 1) and c.inside and c.normal_vector.tuple() == (0, 0, -1)
    w = orc.World()
    s = w.add(orc.SPHERE, orc.translate(0, 0, 1), orc.material(transparency=1.0, refractive_index=1.5))
    c = w.prepare_computations(s, 5.0, orc.ray((0, 0, -5), (0, 0, 1)))  # intersections.rs:259-279
    assert c.over_point.z < -EPS / 2
    assert c.under_point.z > EPS / 2 and c.point.z < c.under_point.z
    w = orc.World()
    p = w.add(orc.PLANE)
    c = w.prepare_computations(p, S2, orc.ray((0, 1, -1), (0, -S2 / 2, S2 / 2)))  # intersections.rs:282-294
    assert close(c.reflect_vector.tuple(), (0, S2 / 2, S2 / 2))


def test_n1_n2(orc):
    w = orc.World()
    glass = lambda ri: orc.material(transparency=1.0, refractive_index=ri)  # noqa: E731
    a = w.add(orc.SPHERE, orc.scale(2, 2, 2), glass(1.5))
    b = w.add(orc.SPHERE, orc.translate(0, 0, -0.25), glass(2.0))
    c = w.add(orc.SPHERE, orc.translate(0, 0, 0.25), glass(2.5))
    r = orc.ray((0, 0, -4), (0, 0, 1))
    xs = [(2.0, a), (2.75, b), (3.25, c), (4.75, b), (5.25, c), (6.0, a)]
    expected = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    for (t, oid), (n1, n2) in zip(xs, expected):  # intersections.rs:297-343 (full-list form, not on the render path)
        comps = w.prepare_computations(oid, t, r, xs)
        assert (comps.n1, comps.n2) == (n1, n2)
    # the form the renderer uses hands in the one-element list [self] (intersections.rs:41-43):
    # n1 is always 1.0 and n2 always the hit object's own refractive index, entering or leaving
    for (t, oid), ri in zip(xs, (1.5, 2.0, 2.5, 2.0, 2.5, 1.5)):
        comps = w.prepare_computations(oid, t, r)
        assert (comps.n1, comps.n2) == (1.0, ri)


def test_schlick(orc):
    w = orc.World()
    s = w.add(orc.SPHERE, mat=orc.material(transparency=1.0, refractive_index=1.5))
    r = orc.ray((0, 0, S2 / 2), (0, 1, 0))
    xs = [(-S2 / 2, s), (S2 / 2, s)]
    assert orc.lib().orc_schlick(w.prepare_computations(s, S2 / 2, r, xs)) == 1.0  # intersections.rs:346-360
    r = orc.ray((0, 0, 0), (0, 1, 0))
    xs = [(-1.0, s), (1.0, s)]
    assert abs(orc.lib().orc_schlick(w.prepare_computations(s, 1.0, r, xs)) - 0.04) < EPS  # :363-374
    r = orc.ray((0, 0.99, -2), (0, 0, 1))
    xs = [(1.8589, s)]
    assert abs(orc.lib().orc_schlick(w.prepare_computations(s, 1.8589, r, xs)) - 0.48873) < EPS  # :377-385


# ---------------------------------------------------------------- scene/materials.rs, scene/patterns.rs tests
def test_lighting(orc):
    w = orc.World()
    s = w.add(orc.SPHERE)
    m = orc.default_material()
    L = lambda light, eye, shadow=False, p=(0, 0, 0), mat=m: w.lighting(s, mat, light, (1, 1, 1), p, eye, (0, 0, -1), shadow)  # noqa: E731
    assert close(L((0, 0, -10), (0, 0, -1)), (1.9, 1.9, 1.9))  # materials.rs:212-227
    assert close(L((0, 0, -10), (0, S2 / 2, -S2 / 2)), (1.0, 1.0, 1.0))  # :230-245
    assert close(L((0, 10, -10), (0, 0, -1)), (0.7364, 0.7364, 0.7364))  # :248-263
    assert close(L((0, 10, -10), (0, -S2 / 2, -S2 / 2)), (1.6364, 1.6364, 1.6364))  # :266-281
    assert close(L((0, 0, 10), (0, 0, -1)), (0.1, 0.1, 0.1))  # :284-299
    assert close(L((0, 0, -10), (0, 0, -1), True), (0.1, 0.1, 0.1))  # :302-317
    pm = orc.material(ambient=1.0, diffuse=0.0, specular=0.0,
                      pattern=orc.pattern(orc.PAT_STRIPE, (1, 1, 1), (0, 0, 0)))  # :320-349
    assert L((0, 0, -10), (0, 0, -1), False, (0.9, 0, 0), pm) == (1, 1, 1)
    assert L((0, 0, -10), (0, 0, -1), False, (1.1, 0, 0), pm) == (0, 0, 0)


def _pat(orc, p, point):
    out = orc.Color()
    orc.lib().orc_pattern_color_at(p, orc.vec(*point), out)
    return out.tuple()


def test_patterns(orc):
    W, B = (1, 1, 1), (0, 0, 0)
    st = orc.pattern(orc.PAT_STRIPE, W, B)  # patterns.rs:203-227
    for p in ((0, 0, 0), (0, 1, 0), (0, 2, 0), (0, 0, 1), (0, 0, 2), (0.9, 0, 0), (-1.1, 0, 0)):
        assert _pat(orc, st, p) == W
    for p in ((1, 0, 0), (-0.1, 0, 0), (-1, 0, 0)):
        assert _pat(orc, st, p) == B
    gr = orc.pattern(orc.PAT_GRADIENT, W, B)  # patterns.rs:264-280
    for x, v in ((0, 1), (0.25, 0.75), (0.5, 0.5), (0.75, 0.25), (1.0, 0.0)):
        assert close(_pat(orc, gr, (x, 0, 0)), (v, v, v))
    ring = orc.pattern(orc.PAT_RING, W, B)  # patterns.rs:283-292
    assert _pat(orc, ring, (0, 0, 0)) == W
    for p in ((1, 0, 0), (0, 0, 1), (0.708, 0, 0.708)):
        assert _pat(orc, ring, p) == B
    ch = orc.pattern(orc.PAT_CHECKERS, W, B)  # patterns.rs:295-316
    for p in ((0, 0, 0), (0.99, 0, 0), (0, 0.99, 0), (0, 0, 0.99)):
        assert _pat(orc, ch, p) == W
    for p in ((1.01, 0, 0), (0, 1.01, 0), (0, 0, 1.01)):
        assert _pat(orc, ch, p) == B


def test_pattern_transforms(orc):
    def at(obj_t, pat_t, p, kind=None):
        w = orc.World()
        s = w.add(orc.SPHERE, obj_t)
        pat = orc.pattern(kind if kind is not None else orc.PAT_TEST, (1, 1, 1), (0, 0, 0), pat_t)
        return w.pattern_color_at_object(s, pat, p)

    assert close(at(orc.scale(2, 2, 2), None, (2, 3, 4)), (1, 1.5, 2))  # patterns.rs:175-181
    assert close(at(None, orc.scale(2, 2, 2), (2, 3, 4)), (1, 1.5, 2))  # :184-190
    assert close(at(orc.scale(2, 2, 2), orc.translate(0.5, 1, 1.5), (2.5, 3, 3.5)), (0.75, 0.5, 0.25))  # :193-200
    assert at(orc.scale(2, 2, 2), None, (1.5, 0, 0), orc.PAT_STRIPE) == (1, 1, 1)  # :230-238
    assert at(None, orc.scale(2, 2, 2), (1.5, 0, 0), orc.PAT_STRIPE) == (1, 1, 1)  # :241-249
    assert at(orc.scale(2, 2, 2), orc.translate(0.5, 0, 0), (2.5, 0, 0), orc.PAT_STRIPE) == (1, 1, 1)  # :252-261


# ---------------------------------------------------------------- rendering/world.rs tests
def _floor_world(orc, **mat):
    w = orc.World.default()
    f = w.add(orc.PLANE, orc.translate(0, -1, 0), orc.material(**mat))
    return w, f


def test_world_intersect_and_shading(orc):
    w = orc.World.default()
    r = orc.ray((0, 0, -5), (0, 0, 1))
    assert [t for t, _ in w.intersect(r)] == [4.0, 4.5, 5.5, 6.0]  # world.rs:265-274
    c = w.prepare_computations(0, 4.0, r)
    assert close(w.shade_hit(c), (0.38066, 0.47583, 0.2855))  # world.rs:277-288
    w2 = orc.World.default()
    w2.set_light((0, 0.25, 0), (1, 1, 1))
    c = w2.prepare_computations(1, 0.5, orc.ray((0, 0, 0), (0, 0, 1)))
    assert close(w2.shade_hit(c), (0.90498, 0.90498, 0.90498))  # world.rs:291-305
    assert w.color_at(orc.ray((0, 0, -5), (0, 1, 0))) == (0, 0, 0)  # world.rs:308-313
    assert close(w.color_at(r), (0.38066, 0.47583, 0.2855))  # world.rs:316-321
    w3 = orc.World()
    w3.set_light((-10, 10, -10), (1, 1, 1))
    w3.add(orc.SPHERE, None, orc.material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0))
    w3.add(orc.SPHERE, orc.scale(0.5, 0.5, 0.5), orc.material(ambient=1.0))
    assert close(w3.color_at(orc.ray((0, 0, 0.75), (0, 0, -1))), (1, 1, 1))  # world.rs:324-346


def test_world_shadows(orc):
    w = orc.World.default()
    assert not w.is_shadowed((0, 10, 0))  # world.rs:349-374
    assert w.is_shadowed((10, -10, 10))
    assert not w.is_shadowed((-20, 20, -20))
    assert not w.is_shadowed((-2, 2, -2))
    w = orc.World()
    w.set_light((0, 0, -10), (1, 1, 1))
    w.add(orc.SPHERE)
    s2 = w.add(orc.SPHERE, orc.translate(0, 0, 10))
    c = w.prepare_computations(s2, 4.0, orc.ray((0, 0, 5), (0, 0, 1)))
    assert close(w.shade_hit(c), (0.1, 0.1, 0.1))  # world.rs:377-393


def test_world_reflection(orc):
    w = orc.World.default()
    p = w.add(orc.PLANE)
    c = w.prepare_computations(p, 1.0, orc.ray((0, 0, 0), (0, 0, 1)))
    assert w.reflected_color(c) == (0, 0, 0)  # world.rs:396-404
    w, f = _floor_world(orc, reflectivity=0.5)
    r = orc.ray((0, 0, -3), (0, -S2 / 2, S2 / 2))
    c = w.prepare_computations(f, S2, r)
    assert close(w.reflected_color(c), (0.19033, 0.23792, 0.14275))  # world.rs:407-437
    assert close(w.shade_hit(c), (0.87676, 0.92434, 0.82917))  # world.rs:440-470
    assert w.reflected_color(c, 0) == (0, 0, 0)  # world.rs:502-524
    m = orc.World()
    m.set_light((0, 0, 0), (1, 1, 1))
    m.add(orc.PLANE, orc.translate(0, -1, 0), orc.material(reflectivity=1.0))
    m.add(orc.PLANE, orc.translate(0, 1, 0), orc.material(reflectivity=1.0))
    cnt = orc.Counters()
    m.color_at(orc.ray((0, 0, 0), (0, 1, 0)), counters=cnt)  # world.rs:474-499: terminates
    assert cnt.reflect == 5  # one shared budget of 5 (world.rs:11-12)


def test_world_refraction(orc):
    w = orc.World.default()
    r = orc.ray((0, 0, -5), (0, 0, 1))
    c = w.prepare_computations(0, 4.0, r)
    assert w.refracted_color(c) == (0, 0, 0)  # world.rs:527-536
    w, f = _floor_world(orc, transparency=1.0, refractive_index=1.5)
    c = w.prepare_computations(f, S2, orc.ray((0, 0, -3), (0, -S2 / 2, S2 / 2)))
    assert w.refracted_color(c, 0) == (0, 0, 0)  # world.rs:539-561
    # total internal reflection (world.rs:564-589), book-style n1/n2 from the full list
    t = orc.World()
    t.set_light((-10, 10, -10), (1, 1, 1))
    s = t.add(orc.SPHERE, None, orc.material(transparency=1.0, refractive_index=1.5))
    r = orc.ray((0, 0, -S2 / 2), (0, 1, 0))
    xs = [(-S2 / 2, s), (S2 / 2, s)]
    assert t.refracted_color(t.prepare_computations(s, S2 / 2, r, xs)) == (0, 0, 0)
    # refracted ray through a test-patterned sphere (world.rs:592-648)
    q = orc.World()
    q.set_light((-10, 10, -10), (1, 1, 1))
    a = q.add(orc.SPHERE, None, orc.material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2, ambient=1.0,
                                             pattern=orc.pattern(orc.PAT_TEST)))
    b = q.add(orc.SPHERE, orc.scale(0.5, 0.5, 0.5), orc.material(transparency=1.0, refractive_index=1.5))
    r = orc.ray((0, 0, 0.1), (0, 1, 0))
    xs = [(-0.9899, a), (-0.4899, b), (0.4899, b), (0.9899, a)]
    assert close(q.refracted_color(q.prepare_computations(b, 0.4899, r, xs)), (0, 0.99888, 0.04722))


def test_world_transparent_and_schlick_blend(orc):
    w, f = _floor_world(orc, transparency=0.5, refractive_index=1.5)
    w.add(orc.SPHERE, orc.translate(0, -3.5, -0.5), orc.material(color=(1, 0, 0), ambient=0.5))
    r = orc.ray((0, 0, -3), (0, -S2 / 2, S2 / 2))
    c = w.prepare_computations(f, S2, r, [(S2, f)])
    assert close(w.shade_hit(c), (0.93642, 0.68642, 0.68642))  # world.rs:651-695
    w, f = _floor_world(orc, reflectivity=0.5, transparency=0.5, refractive_index=1.5)
    w.add(orc.SPHERE, orc.translate(0, -3.5, -0.5), orc.material(color=(1, 0, 0), ambient=0.5))
    c = w.prepare_computations(f, S2, r, [(S2, f)])
    assert close(w.shade_hit(c), (0.93391, 0.69643, 0.69243))  # world.rs:698-744


# ---------------------------------------------------------------- rendering/camera.rs tests
def test_camera(orc):
    assert abs(orc.camera(200, 125, math.pi / 2).pixel_size - 0.01) < EPS  # camera.rs:94-103
    assert abs(orc.camera(125, 200, math.pi / 2).pixel_size - 0.01) < EPS
    cam = orc.camera(201, 101, math.pi / 2)
    r = orc.ray_for_pixel(cam, 100, 50)  # camera.rs:106-111
    assert r.origin.tuple() == (0, 0, 0) and close(r.direction.tuple(), (0, 0, -1))
    r = orc.ray_for_pixel(cam, 0, 0)  # camera.rs:114-122
    assert close(r.direction.tuple(), (0.6651864261194508, 0.3325932130597254, -0.6685123582500481))
    cam.transform = orc.mul(orc.rotate_y(math.pi / 4), orc.translate(0, -2, 5))
    r = orc.ray_for_pixel(cam, 100, 50)  # camera.rs:125-134
    assert close(r.origin.tuple(), (0, 2, -5)) and close(r.direction.tuple(), (S2 / 2, 0, -S2 / 2))
    m = orc.Mat4()
    orc.lib().orc_view_transform(orc.vec(0, 0, -5), orc.vec(0, 0, 0), orc.vec(0, 1, 0), m)
    img = orc.render(orc.camera(11, 11, math.pi / 2, m), orc.World.default())  # camera.rs:137-147
    assert close(img[5, 5], (0.38066, 0.47583, 0.2855))
    # square frames take the `else` branch of the aspect test (camera.rs:22-26)
    sq = orc.camera(100, 100, math.pi / 2)
    assert sq.half_width == sq.half_height


# ---------------------------------------------------------------- rendering/canvas.rs tests
def test_canvas_ppm(orc):
    import numpy as np

    img = np.zeros((3, 5, 3))
    ppm = orc.to_ppm(img).decode()
    lines = ppm.split("\n")
    assert lines[:3] == ["P3", "5 3", "255"]  # canvas.rs:165-172
    img[0, 0] = (1.5, 0, 0)
    img[1, 2] = (0, 0.5, 0)
    img[2, 4] = (-0.5, 0, 1)
    lines = orc.to_ppm(img).decode().split("\n")  # canvas.rs:175-188
    assert lines[3] == "255 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
    assert lines[4] == "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0"
    assert lines[5] == "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"
    img = np.zeros((2, 10, 3))
    img[:, :] = (1.0, 0.8, 0.6)
    ppm = orc.to_ppm(img).decode()
    lines = ppm.split("\n")  # canvas.rs:191-213
    assert lines[3] == "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204"
    assert lines[4] == "153 255 204 153 255 204 153 255 204 153 255 204 153"
    assert lines[5] == lines[3] and lines[6] == lines[4]
    assert ppm.endswith("\n\n")  # canvas.rs:216-220 and the trailing blank line of canvas.rs:88
    assert [orc.lib().orc_quantize(v) for v in (0.0, 0.5, 1.0, 1.5, -0.5, float("nan"), 0.00196)] == \
        [0, 128, 255, 255, 0, 0, 0]


# ---------------------------------------------------------------- mode equivalence
@pytest.mark.parametrize("scene", ["chapter_11", "chapter_13", "chapter_14"])
def test_faithful_mode_equals_cached_mode(orc, scene):
    """Per-ray inverses + per-ray Group::bounds() + intersection list with stable sort (what the reference executes)
    give bit-identical frames and counters to the cached inverses + running-minimum form."""
    import numpy as np

    from conftest import scene_path

    cam, w = orc.load_scene(scene_path(scene), 40, 24)
    c0, c1 = orc.Counters(), orc.Counters()
    a = orc.render(cam, w, counters=c0)
    w.set_mode(1)
    b = orc.render(cam, w, counters=c1)
    assert np.array_equal(a, b)
    assert c0.as_dict() == c1.as_dict()
