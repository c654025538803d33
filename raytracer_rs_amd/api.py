"""Host-side mirror of the reference's render API (`SceneFile`, `Camera`, `World`, `Canvas`, `Object`, `Material`,
`Pattern`, `Matrix4`, `Ray`) over the C ABI.  Same names, argument meaning and error behaviour as the Rust items
they stand in for (cited per class); all rendering work happens in the HIP library.

    scene  = SceneFile.from_file("scenes/chapter_11.toml")      # scenes/loader.rs:190
    camera = scene.build_camera()                                # scenes/loader.rs:196
    world  = scene.build_world()                                 # scenes/loader.rs:214
    canvas = camera.render(world)                                # rendering/camera.rs:56   (GPU)
    canvas.to_ppm("out.ppm")                                     # rendering/canvas.rs:116
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from ._lib import RtError

EPSILON = 1e-5  # core/floats.rs:2


# --------------------------------------------------------------------------------------------- core/matrices.rs
class Matrix4:
    """core/matrices.rs:7-9.  Row-major 4x4 of f64; products accumulate left to right from 0.0 (matrices.rs:178-192)."""

    __slots__ = ("data",)

    def __init__(self, data: Sequence[Sequence[float]]):
        self.data = [[float(v) for v in row] for row in data]

    @staticmethod
    def identity() -> "Matrix4":
        return Matrix4([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])

    @staticmethod
    def translate(x: float, y: float, z: float) -> "Matrix4":
        return Matrix4([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]])

    @staticmethod
    def scale(x: float, y: float, z: float) -> "Matrix4":
        return Matrix4([[x, 0, 0, 0], [0, y, 0, 0], [0, 0, z, 0], [0, 0, 0, 1]])

    @staticmethod
    def rotate_x(r: float) -> "Matrix4":
        return Matrix4([[1, 0, 0, 0], [0, math.cos(r), -math.sin(r), 0], [0, math.sin(r), math.cos(r), 0], [0, 0, 0, 1]])

    @staticmethod
    def rotate_y(r: float) -> "Matrix4":
        return Matrix4([[math.cos(r), 0, math.sin(r), 0], [0, 1, 0, 0], [-math.sin(r), 0, math.cos(r), 0], [0, 0, 0, 1]])

    @staticmethod
    def rotate_z(r: float) -> "Matrix4":
        return Matrix4([[math.cos(r), -math.sin(r), 0, 0], [math.sin(r), math.cos(r), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])

    @staticmethod
    def shear(xy, xz, yx, yz, zx, zy) -> "Matrix4":
        return Matrix4([[1, xy, xz, 0], [yx, 1, yz, 0], [zx, zy, 1, 0], [0, 0, 0, 1]])

    def __mul__(self, other: "Matrix4") -> "Matrix4":
        a, b = self.data, other.data
        out = []
        for r in range(4):
            row = []
            for c in range(4):
                s = 0.0
                for i in range(4):
                    s += a[r][i] * b[i][c]
                row.append(s)
            out.append(row)
        return Matrix4(out)

    def flat(self) -> List[float]:
        return [v for row in self.data for v in row]

    @staticmethod
    def from_flat(vals: Sequence[float]) -> "Matrix4":
        return Matrix4([[vals[r * 4 + c] for c in range(4)] for r in range(4)])

    def __eq__(self, other) -> bool:  # matrices.rs:223-234
        return all(abs(self.data[r][c] - other.data[r][c]) < EPSILON for r in range(4) for c in range(4))

    def __repr__(self) -> str:
        return f"Matrix4({self.data})"


def view_transform(from_: Sequence[float], to: Sequence[float], up: Sequence[float]) -> Matrix4:
    """scene/transformations.rs:4-17 (evaluated by the library's host code)."""
    out = (C.c_double * 16)()
    L.check(L.lib().rt_view_transform((C.c_double * 3)(*from_), (C.c_double * 3)(*to), (C.c_double * 3)(*up), out))
    return Matrix4.from_flat(list(out))


@dataclass
class Ray:
    """rendering/rays.rs:5-8"""
    origin: Tuple[float, float, float]
    direction: Tuple[float, float, float]


# --------------------------------------------------------------------------------------------- scene/*
@dataclass
class Pattern:
    """scene/patterns.rs:7-32"""
    kind: int
    color_a: Tuple[float, float, float]
    color_b: Tuple[float, float, float]
    transform: Matrix4 = field(default_factory=Matrix4.identity)

    @staticmethod
    def stripe(a, b) -> "Pattern":
        return Pattern(L.PATTERN_STRIPE, tuple(a), tuple(b))

    @staticmethod
    def gradient(a, b) -> "Pattern":
        return Pattern(L.PATTERN_GRADIENT, tuple(a), tuple(b))

    @staticmethod
    def ring(a, b) -> "Pattern":
        return Pattern(L.PATTERN_RING, tuple(a), tuple(b))

    @staticmethod
    def checkers(a, b) -> "Pattern":
        return Pattern(L.PATTERN_CHECKERS, tuple(a), tuple(b))

    def set_transform(self, m: Matrix4) -> None:
        self.transform = m


@dataclass
class Material:
    """scene/materials.rs:9-19 with the builder defaults of materials.rs:35-45"""
    color: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    ambient: float = 0.1
    diffuse: float = 0.9
    specular: float = 0.9
    shininess: float = 200.0
    reflectivity: float = 0.0
    transparency: float = 0.0
    refractive_index: float = 1.0
    pattern: Optional[Pattern] = None

    @staticmethod
    def build(**kw) -> "Material":
        """MaterialBuilder::build (materials.rs:93-108): negative ambient/diffuse/specular/shininess are refused."""
        m = Material(**kw)
        if m.ambient < 0.0 or m.diffuse < 0.0 or m.specular < 0.0 or m.shininess < 0.0:
            raise RtError(-3, "Material values must be positive")
        return m

    def _desc(self) -> L.MaterialDesc:
        d = L.MaterialDesc()
        d.color[:] = self.color
        d.ambient, d.diffuse, d.specular, d.shininess = self.ambient, self.diffuse, self.specular, self.shininess
        d.reflectivity, d.transparency, d.refractive_index = self.reflectivity, self.transparency, self.refractive_index
        d.pattern.kind = L.PATTERN_NONE
        d.pattern.transform[:] = Matrix4.identity().flat()
        if self.pattern is not None:
            d.pattern.kind = self.pattern.kind
            d.pattern.transform[:] = self.pattern.transform.flat()
            d.pattern.color_a[:] = self.pattern.color_a
            d.pattern.color_b[:] = self.pattern.color_b
        return d

    @staticmethod
    def _from_desc(d: L.MaterialDesc) -> "Material":
        pat = None
        if d.pattern.kind != L.PATTERN_NONE:
            pat = Pattern(d.pattern.kind, tuple(d.pattern.color_a), tuple(d.pattern.color_b),
                          Matrix4.from_flat(list(d.pattern.transform)))
        return Material(tuple(d.color), d.ambient, d.diffuse, d.specular, d.shininess, d.reflectivity, d.transparency,
                        d.refractive_index, pat)


@dataclass
class PointLight:
    """scene/lights.rs:5-17"""
    position: Tuple[float, float, float]
    intensity: Tuple[float, float, float]


class Object:
    """rendering/objects.rs:34-41: one of sphere / plane / cube / cylinder / cone / group."""

    def __init__(self, kind: int, transformation: Optional[Matrix4] = None, material: Optional[Material] = None,
                 minimum: float = -math.inf, maximum: float = math.inf, closed: bool = False):
        self.kind = kind
        self.transformation = transformation if transformation is not None else Matrix4.identity()
        self.world_transformation = self.transformation
        self.material = material if material is not None else Material()
        self.minimum, self.maximum, self.closed = minimum, maximum, closed
        self.children: List["Object"] = []

    # objects.rs:43-72
    @staticmethod
    def sphere(**kw) -> "Object":
        return Object(L.SPHERE, **kw)

    @staticmethod
    def glass_sphere() -> "Object":  # geometry/sphere.rs:36-46
        return Object(L.SPHERE, material=Material(transparency=1.0, refractive_index=1.5))

    @staticmethod
    def plane(**kw) -> "Object":
        return Object(L.PLANE, **kw)

    @staticmethod
    def cube(**kw) -> "Object":
        return Object(L.CUBE, **kw)

    @staticmethod
    def cylinder(**kw) -> "Object":
        return Object(L.CYLINDER, **kw)

    @staticmethod
    def cone(**kw) -> "Object":
        return Object(L.CONE, **kw)

    @staticmethod
    def group(**kw) -> "Object":
        return Object(L.GROUP, **kw)

    def set_transform(self, m: Matrix4) -> None:
        """Transformable::set_transform (objects.rs:223-236); world transforms are re-derived when the world is sent
        to the device."""
        self.transformation = m

    def set_material(self, m: Material) -> None:
        self.material = m

    def add_child(self, child: "Object") -> None:
        """Group::add_child (geometry/groups.rs:45-58)"""
        if self.kind != L.GROUP:
            raise RtError(-1, "add_child called on non-group object")
        self.children.append(child)


def _assign_world_transforms(objects: Sequence[Object]) -> None:
    """Cached world transformations as SceneFile::build_world leaves them (scenes/loader.rs:229-241,
    geometry/groups.rs:85-105): a top-level object keeps its own; children of a top-level group get
    (identity * group) * child; below that parent_world * local."""

    def below(obj: Object, world: Matrix4) -> None:
        obj.world_transformation = world
        for ch in obj.children:
            below(ch, world * ch.transformation)

    for obj in objects:
        obj.world_transformation = obj.transformation
        if obj.kind == L.GROUP:
            pg = Matrix4.identity() * obj.transformation
            for ch in obj.children:
                below(ch, pg * ch.transformation)


def _flatten_objects(objects: Sequence[Object], out: List[L.ObjectDesc], index: dict) -> None:
    for obj in objects:
        d = L.ObjectDesc()
        d.kind = obj.kind
        d.n_children = len(obj.children) if obj.kind == L.GROUP else 0
        d.closed = int(bool(obj.closed))
        d.minimum, d.maximum = obj.minimum, obj.maximum
        d.transformation[:] = obj.transformation.flat()
        d.world_transformation[:] = obj.world_transformation.flat()
        d.material = obj.material._desc()
        index[id(obj)] = len(out)
        out.append(d)
        if obj.kind == L.GROUP:
            _flatten_objects(obj.children, out, index)


def _set_depth(opts: "L.RenderOpts", max_depth: Optional[int]) -> None:
    """None = the reference's DEFAULT_MAX_*_DEPTH (world.rs:11-12); a number is the budget exactly, 0 included."""
    if max_depth is not None:
        opts.max_depth = int(max_depth)
        opts.flags |= L.RENDER_EXACT_DEPTH


class Scene:
    """A World resident on one GPU (`rt_scene`)."""

    def __init__(self, world_desc: L.WorldDesc, device: int = -1, keepalive=None):
        self._h = C.c_void_p()
        self._keep = keepalive
        L.check(L.lib().rt_scene_create(C.byref(world_desc), device, C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            L.lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> L.SceneInfo:
        i = L.SceneInfo()
        L.check(L.lib().rt_scene_get_info(self._h, C.byref(i)))
        return i

    def render(self, camera: "Camera", rows: Optional[Tuple[int, int]] = None, max_depth: Optional[int] = None,
               stats: bool = False, out: Optional[np.ndarray] = None,
               bands: Optional[Tuple[int, int, int]] = None):
        """rt_render into a host array [vsize, hsize, 3]; returns (array, Stats).  `rows` = (begin, end) restricts
        the call to a row range, `bands` = (band_rows, first, step) to every step-th band of band_rows rows."""
        cd = camera._desc()
        if out is None:
            out = np.zeros((camera.vsize, camera.hsize, 3), dtype=np.float64)
        _check_frame(out, camera)
        opts = L.RenderOpts()
        if bands is not None:
            opts.band_rows, opts.band_first, opts.band_step = bands
        if rows is not None:
            opts.row_begin, opts.row_end = rows
        _set_depth(opts, max_depth)
        opts.collect_stats = int(stats)
        st = L.Stats()
        L.check(L.lib().rt_render(self._h, C.byref(cd), C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_double)),
                                  C.byref(st)))
        return out, st

    def render_times(self, back: int = 0) -> Tuple[float, float]:
        """rt_scene_render_times: (tile-ordering ms, render-kernel ms) of the launch made with timed=True `back`
        launches ago; synchronise the launch's stream first."""
        a, b = C.c_double(), C.c_double()
        L.check(L.lib().rt_scene_render_times(self._h, back, C.byref(a), C.byref(b)))
        return a.value, b.value

    def render_device(self, camera: "Camera", device_ptr: int, rows: Optional[Tuple[int, int]] = None,
                      max_depth: Optional[int] = None, stream: int = 0, stats: bool = False, timed: bool = False,
                      quantized: bool = False, in_place: bool = False) -> Optional[L.Stats]:
        """rt_render_device: asynchronous launch on `stream` into device memory at `device_ptr`
        ((row_end-row_begin)*hsize*3 doubles; with `in_place` the whole frame, the rows landing at their own place)."""
        cd = camera._desc()
        opts = L.RenderOpts()
        if rows is not None:
            opts.row_begin, opts.row_end = rows
        opts.collect_stats = int(stats)
        opts.stream = C.c_void_p(stream)
        opts.flags = (L.RENDER_TIMED if timed else 0) | (L.RENDER_QUANTIZED if quantized else 0) | (L.RENDER_IN_PLACE if in_place else 0)
        _set_depth(opts, max_depth)
        st = L.Stats()
        L.check(L.lib().rt_render_device(self._h, C.byref(cd), C.byref(opts), C.c_void_p(device_ptr), C.byref(st)))
        return st if stats else None

    def render_device_bands(self, camera: "Camera", device_ptr: int, band_rows: int, band_first: int, band_step: int,
                            max_depth: Optional[int] = None, stream: int = 0, stats: bool = False, timed: bool = False,
                            quantized: bool = False, in_place: bool = False) -> Optional[L.Stats]:
        """rt_render_device restricted to bands band_first, band_first+band_step, ... of band_rows rows each; the
        bands are written one after another at `device_ptr` (with `in_place`: at their own rows of the whole frame there)."""
        cd = camera._desc()
        opts = L.RenderOpts()
        opts.band_rows, opts.band_first, opts.band_step = band_rows, band_first, band_step
        opts.collect_stats = int(stats)
        opts.stream = C.c_void_p(stream)
        opts.flags = (L.RENDER_TIMED if timed else 0) | (L.RENDER_QUANTIZED if quantized else 0) | (L.RENDER_IN_PLACE if in_place else 0)
        _set_depth(opts, max_depth)
        st = L.Stats()
        L.check(L.lib().rt_render_device(self._h, C.byref(cd), C.byref(opts), C.c_void_p(device_ptr), C.byref(st)))
        return st if stats else None

    def render_device_noalloc(self, camera: "Camera", device_ptr: int, rows: Optional[Tuple[int, int]] = None,
                              bands: Optional[Tuple[int, int, int]] = None, max_depth: Optional[int] = None, stream: int = 0,
                              quantized: bool = False) -> None:
        """rt_render_device_noalloc: rt_render_device without any allocation (its tile-ordering scratch is a static
        device array); at most NOALLOC_MAX_TILES 8x8-pixel tiles per call."""
        cd = camera._desc()
        opts = L.RenderOpts()
        if rows is not None:
            opts.row_begin, opts.row_end = rows
        if bands is not None:
            opts.band_rows, opts.band_first, opts.band_step = bands
        opts.stream = C.c_void_p(stream)
        opts.flags = L.RENDER_QUANTIZED if quantized else 0
        _set_depth(opts, max_depth)
        L.check(L.lib().rt_render_device_noalloc(self._h, C.byref(cd), C.byref(opts), C.c_void_p(device_ptr)))

    def color_at(self, rays: np.ndarray, remaining: int = L.DEFAULT_DEPTH, stats: bool = False):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 3), dtype=np.float64)
        st = L.Stats()
        L.check(L.lib().rt_color_at(self._h, rays.ctypes.data_as(C.POINTER(C.c_double)), rays.shape[0], remaining,
                                    out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st) if stats else None))
        return (out, st) if stats else out

    def intersect(self, rays: np.ndarray):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        t = np.zeros(rays.shape[0], dtype=np.float64)
        obj = np.zeros(rays.shape[0], dtype=np.int32)
        L.check(L.lib().rt_intersect(self._h, rays.ctypes.data_as(C.POINTER(C.c_double)), rays.shape[0],
                                     t.ctypes.data_as(C.POINTER(C.c_double)), obj.ctypes.data_as(C.POINTER(C.c_int32))))
        return t, obj

    def intersect_all(self, rays: np.ndarray, capacity: int = 16):
        """rt_intersect_all: per ray the sorted list [(t, object), ...] of World::intersect.  A list longer than
        `capacity` makes the call again with the capacity the library reports."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        while True:
            count = np.zeros(n, dtype=np.uint32)
            t = np.zeros((n, capacity), dtype=np.float64)
            obj = np.zeros((n, capacity), dtype=np.int32)
            rc = L.lib().rt_intersect_all(self._h, rays.ctypes.data_as(C.POINTER(C.c_double)), n, capacity,
                                          count.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          t.ctypes.data_as(C.POINTER(C.c_double)),
                                          obj.ctypes.data_as(C.POINTER(C.c_int32)))
            if rc == -9:  # RT_ERR_CAPACITY
                capacity = int(count.max())
                continue
            L.check(rc)
            return [[(float(t[i, k]), int(obj[i, k])) for k in range(int(count[i]))] for i in range(n)]

    def prepare_computations(self, rays: np.ndarray, t: Sequence[float], objects: Sequence[int],
                             xs: Optional[Sequence[Sequence[Tuple[float, int]]]] = None):
        """rt_prepare_computations -> ctypes array of Computations; xs[i] is hit i's intersection list or None."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        tt = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
        oo = np.ascontiguousarray(objects, dtype=np.int32).reshape(-1)
        out = (L.Computations * max(1, n))()
        begin = xs_t = xs_o = None
        if xs is not None:
            lists = [list(x) if x is not None else [(float(tt[i]), int(oo[i]))] for i, x in enumerate(xs)]
            begin = np.zeros(n + 1, dtype=np.uint32)
            begin[1:] = np.cumsum([len(x) for x in lists])
            xs_t = np.array([e[0] for x in lists for e in x] or [0.0], dtype=np.float64)
            xs_o = np.array([e[1] for x in lists for e in x] or [0], dtype=np.int32)
        L.check(L.lib().rt_prepare_computations(
            self._h, rays.ctypes.data_as(C.POINTER(C.c_double)), tt.ctypes.data_as(C.POINTER(C.c_double)),
            oo.ctypes.data_as(C.POINTER(C.c_int32)), n,
            begin.ctypes.data_as(C.POINTER(C.c_uint32)) if begin is not None else None,
            xs_t.ctypes.data_as(C.POINTER(C.c_double)) if xs_t is not None else None,
            xs_o.ctypes.data_as(C.POINTER(C.c_int32)) if xs_o is not None else None, out))
        return out

    def _part(self, fn, comps, n: int, remaining: int) -> np.ndarray:
        out = np.zeros((n, 3), dtype=np.float64)
        L.check(fn(self._h, comps, n, remaining, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def shade_hit(self, comps, n: int, remaining: int = L.DEFAULT_DEPTH) -> np.ndarray:
        return self._part(L.lib().rt_shade_hit, comps, n, remaining)

    def reflected_color(self, comps, n: int, remaining: int = L.DEFAULT_DEPTH) -> np.ndarray:
        return self._part(L.lib().rt_reflected_color, comps, n, remaining)

    def refracted_color(self, comps, n: int, remaining: int = L.DEFAULT_DEPTH) -> np.ndarray:
        return self._part(L.lib().rt_refracted_color, comps, n, remaining)

    def is_shadowed(self, points: np.ndarray) -> np.ndarray:
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(pts.shape[0], dtype=np.uint8)
        L.check(L.lib().rt_is_shadowed(self._h, pts.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[0],
                                       out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    def normal_at(self, objects: Sequence[int], points: np.ndarray) -> np.ndarray:
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        obj = np.ascontiguousarray(objects, dtype=np.int32).reshape(-1)
        out = np.zeros((pts.shape[0], 3), dtype=np.float64)
        L.check(L.lib().rt_normal_at(self._h, obj.ctypes.data_as(C.POINTER(C.c_int32)),
                                     pts.ctypes.data_as(C.POINTER(C.c_double)), pts.shape[0],
                                     out.ctypes.data_as(C.POINTER(C.c_double))))
        return out


class MultiScene:
    """A World resident on several GPUs of one node (`rt_multi`): bands of rows handed out dynamically, the frame
    assembled on devices[0] (render_device) or in host memory (render)."""

    def __init__(self, world_desc: L.WorldDesc, devices: Sequence[int], keepalive=None):
        self._h = C.c_void_p()
        self._keep = keepalive
        self.devices = list(devices)
        arr = (C.c_int * len(self.devices))(*self.devices)
        L.check(L.lib().rt_multi_create(C.byref(world_desc), arr, len(self.devices), C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            L.lib().rt_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, band_rows: int, max_depth: Optional[int], stats: bool, quantized: bool, units: int = 0) -> L.RenderOpts:
        opts = L.RenderOpts()
        opts.band_rows = band_rows  # height of a stripe (0: the library's 16)
        opts.band_step = units      # number of interleaved units (0: two per worker)
        opts.collect_stats = int(stats)
        opts.flags = L.RENDER_QUANTIZED if quantized else 0
        _set_depth(opts, max_depth)
        return opts

    def render(self, camera: "Camera", band_rows: int = 0, max_depth: Optional[int] = None, stats: bool = False,
               out: Optional[np.ndarray] = None, units: int = 0):
        """rt_multi_render into a host array [vsize, hsize, 3]; returns (array, Stats)."""
        cd = camera._desc()
        if out is None:
            out = np.zeros((camera.vsize, camera.hsize, 3), dtype=np.float64)
        _check_frame(out, camera)
        opts = self._opts(band_rows, max_depth, stats, False, units)
        st = L.Stats()
        L.check(L.lib().rt_multi_render(self._h, C.byref(cd), C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.byref(st)))
        return out, st

    def render_device(self, camera: "Camera", device_ptr: int, band_rows: int = 0, max_depth: Optional[int] = None,
                      stats: bool = False, quantized: bool = False, units: int = 0) -> L.Stats:
        """rt_multi_render_device: the whole frame into memory of devices[0] at `device_ptr`; returns when complete."""
        cd = camera._desc()
        opts = self._opts(band_rows, max_depth, stats, quantized, units)
        st = L.Stats()
        L.check(L.lib().rt_multi_render_device(self._h, C.byref(cd), C.byref(opts), C.c_void_p(device_ptr), C.byref(st)))
        return st

    def units_taken(self) -> List[int]:
        arr = (C.c_uint32 * len(self.devices))()
        L.check(L.lib().rt_multi_units_taken(self._h, arr, len(self.devices)))
        return list(arr)

    def worker_ms(self) -> List[float]:
        """When each worker's last kernel of the last call had finished, in ms since the call began."""
        arr = (C.c_double * len(self.devices))()
        L.check(L.lib().rt_multi_worker_ms(self._h, arr, len(self.devices)))
        return list(arr)

    def info(self) -> dict:
        """rt_multi_get_info: workers, distinct GPUs, how bands are collected, the band plan of the last call."""
        mi = L.MultiInfo()
        L.check(L.lib().rt_multi_get_info(self._h, C.byref(mi)))
        return {"n_workers": mi.n_workers, "n_devices": mi.n_devices, "gather": L.GATHER_NAMES.get(mi.gather, str(mi.gather)),
                "rccl_ranks": mi.rccl_ranks, "band_rows": mi.band_rows, "n_units": mi.n_units, "note": mi.note.decode(errors="replace")}


def _check_frame(out: np.ndarray, camera: "Camera") -> None:
    """A caller-provided frame must be what the C side writes: float64, C-contiguous, [vsize, hsize, 3]."""
    if out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"] or out.shape != (camera.vsize, camera.hsize, 3):
        raise ValueError(f"out must be a C-contiguous float64 array of shape ({camera.vsize}, {camera.hsize}, 3)")


class World:
    """rendering/world.rs:14-17: `objects` + `light_source`.  Queries run on the GPU through a cached `Scene`;
    call `invalidate()` after mutating objects of a world that has already been used."""

    def __init__(self):
        self.objects: List[Object] = []
        self.light_source: Optional[PointLight] = None
        self._scene: Optional[Scene] = None
        self._loaded = None  # (WorldDesc, owner) when built by the C++ loader
        self._index: dict = {}

    @staticmethod
    def new() -> "World":
        return World()

    @staticmethod
    def default() -> "World":
        """rendering/world.rs:201-232"""
        w = World()
        w.light_source = PointLight((-10.0, 10.0, -10.0), (1.0, 1.0, 1.0))
        w.objects.append(Object.sphere(material=Material(color=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)))
        w.objects.append(Object.sphere(transformation=Matrix4.scale(0.5, 0.5, 0.5)))
        return w

    def invalidate(self) -> None:
        if self._scene is not None:
            self._scene.close()
        self._scene = None

    def world_desc(self) -> Tuple[L.WorldDesc, object]:
        if self._loaded is not None:
            return self._loaded
        _assign_world_transforms(self.objects)
        flat: List[L.ObjectDesc] = []
        self._index = {}
        _flatten_objects(self.objects, flat, self._index)
        arr = (L.ObjectDesc * max(1, len(flat)))(*flat)
        wd = L.WorldDesc()
        wd.objects = C.cast(arr, C.POINTER(L.ObjectDesc))
        wd.n_objects = len(flat)
        wd.has_light = int(self.light_source is not None)
        if self.light_source is not None:
            wd.light_position[:] = self.light_source.position
            wd.light_intensity[:] = self.light_source.intensity
        return wd, arr

    def object_index(self, obj: Object) -> int:
        """Index of `obj` in the depth-first object list the device reports hits against."""
        self.scene()
        return self._index[id(obj)]

    def multi_scene(self, devices: Sequence[int]) -> MultiScene:
        """The world on several GPUs (a new rt_multi every call: the caller keeps it)."""
        wd, keep = self.world_desc()
        return MultiScene(wd, devices, keepalive=keep)

    def scene(self, device: int = -1) -> Scene:
        if self._scene is None:
            wd, keep = self.world_desc()
            self._scene = Scene(wd, device, keepalive=keep)
        return self._scene

    def release_scene(self) -> None:
        """Destroys the GPU copy of the world; the next scene() makes a new one (and reads the environment again)."""
        if self._scene is not None:
            self._scene.close()
            self._scene = None

    # rendering/world.rs:27-36 + intersections.rs:37-39: the hit of one ray, or None
    def hit(self, ray: Ray) -> Optional[Tuple[float, int]]:
        t, obj = self.scene().intersect(np.array([list(ray.origin) + list(ray.direction)]))
        return None if obj[0] < 0 else (float(t[0]), int(obj[0]))

    def intersect(self, ray: Ray) -> List[Tuple[float, int]]:
        """World::intersect (world.rs:27-36): every intersection as (t, object index), sorted by t."""
        return self.scene().intersect_all(np.array([list(ray.origin) + list(ray.direction)]))[0]

    def prepare_computations(self, t: float, obj: int, ray: Ray,
                             xs: Optional[Sequence[Tuple[float, int]]] = None) -> L.Computations:
        """Intersection::prepare_computations / _for_intersections (intersections.rs:41-118) of the hit (t, obj)."""
        r = np.array([list(ray.origin) + list(ray.direction)])
        return self.scene().prepare_computations(r, [t], [obj], None if xs is None else [xs])[0]

    def _one(self, fn, comps: L.Computations, remaining: int) -> Tuple[float, float, float]:
        arr = (L.Computations * 1)(comps)
        return tuple(float(v) for v in fn(arr, 1, remaining)[0])

    def shade_hit(self, comps: L.Computations, remaining: int = L.DEFAULT_DEPTH) -> Tuple[float, float, float]:
        """World::shade_hit (world.rs:80-111)"""
        return self._one(self.scene().shade_hit, comps, remaining)

    def reflected_color(self, comps: L.Computations, remaining: int = L.DEFAULT_DEPTH) -> Tuple[float, float, float]:
        """World::reflected_color (world.rs:151-168)"""
        return self._one(self.scene().reflected_color, comps, remaining)

    def refracted_color(self, comps: L.Computations, remaining: int = L.DEFAULT_DEPTH) -> Tuple[float, float, float]:
        """World::refracted_color (world.rs:170-198)"""
        return self._one(self.scene().refracted_color, comps, remaining)

    def color_at(self, ray: Ray, remaining: int = L.DEFAULT_DEPTH) -> Tuple[float, float, float]:
        """World::color_at (world.rs:113-126)"""
        c = self.scene().color_at(np.array([list(ray.origin) + list(ray.direction)]), remaining)
        return tuple(float(v) for v in c[0])

    def is_shadowed(self, point: Sequence[float]) -> bool:
        """World::is_shadowed (world.rs:128-149)"""
        return bool(self.scene().is_shadowed(np.array([list(point)]))[0])


def quantize_device(rgb_device_ptr: int, n: int, out_device_ptr: int, stream: int = 0) -> None:
    """rt_canvas_quantize_device: canvas.rs:123-130 over n f64 values in device memory -> n bytes in device memory."""
    L.check(L.lib().rt_canvas_quantize_device(C.c_void_p(rgb_device_ptr), n, C.c_void_p(out_device_ptr), C.c_void_p(stream)))


class Canvas:
    """rendering/canvas.rs:7-11: width, height, pixels (row-major, 3 f64 per pixel)."""

    def __init__(self, width: int, height: int, pixels: Optional[np.ndarray] = None):
        self.width, self.height = int(width), int(height)
        self.pixels = pixels if pixels is not None else np.zeros((self.height, self.width, 3), dtype=np.float64)

    def write_pixel(self, x: int, y: int, color: Sequence[float]) -> None:
        if x >= self.width or y >= self.height or x < 0 or y < 0:  # canvas.rs:27-29: silently ignored
            return
        self.pixels[y, x, :] = color

    def pixel_at(self, x: int, y: int) -> Tuple[float, float, float]:
        if not (0 <= x < self.width and 0 <= y < self.height):
            raise IndexError("pixel out of bounds")  # canvas.rs:34-37 panics on the index
        return tuple(float(v) for v in self.pixels[y, x])

    def _ptr(self):
        self.pixels = np.ascontiguousarray(self.pixels, dtype=np.float64)
        return self.pixels.ctypes.data_as(C.POINTER(C.c_double))

    def to_ppm_string(self) -> bytes:
        """Canvas::to_ppm_string (canvas.rs:44-90)"""
        n = L.lib().rt_canvas_to_ppm(self._ptr(), self.width, self.height, None, 0)
        buf = C.create_string_buffer(n)
        L.lib().rt_canvas_to_ppm(self._ptr(), self.width, self.height, buf, n)
        return buf.raw[:n]

    def to_ppm(self, path: str) -> None:
        """Canvas::to_ppm (canvas.rs:116-120)"""
        L.check(L.lib().rt_canvas_write_ppm(self._ptr(), self.width, self.height, path.encode()))

    def to_jpeg_bytes(self) -> bytes:
        """The file Canvas::to_jpeg writes (canvas.rs:92-114: quality 100 through the `image` crate's encoder), in memory."""
        # one call when the file fits a buffer of the frame's raw size (it nearly always does); the call returns the
        # length needed either way, and a second call follows only if it did not fit
        cap = self.width * self.height * 3 + 4096
        buf = C.create_string_buffer(cap)
        n = L.lib().rt_canvas_to_jpeg(self._ptr(), self.width, self.height, buf, cap)
        if n > cap:
            buf = C.create_string_buffer(n)
            L.lib().rt_canvas_to_jpeg(self._ptr(), self.width, self.height, buf, n)
        return buf.raw[:n]

    def to_jpeg(self, path: str) -> None:
        """Canvas::to_jpeg (canvas.rs:109-114)"""
        L.check(L.lib().rt_canvas_write_jpeg(self._ptr(), self.width, self.height, path.encode()))

    def to_rgb8(self) -> np.ndarray:
        """The 8-bit image of canvas.rs:123-130 as uint8 [height, width, 3]."""
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        L.check(L.lib().rt_canvas_quantize(self._ptr(), self.pixels.size, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out


class Camera:
    """rendering/camera.rs:8-16"""

    def __init__(self, hsize: int, vsize: int, field_of_view: float):
        d = L.CameraDesc()
        L.check(L.lib().rt_camera_new(hsize, vsize, field_of_view, C.byref(d)))
        self.hsize, self.vsize, self.field_of_view = int(hsize), int(vsize), float(field_of_view)
        self.transform = Matrix4.identity()
        self.pixel_size, self.half_width, self.half_height = d.pixel_size, d.half_width, d.half_height

    @staticmethod
    def new(hsize: int, vsize: int, field_of_view: float) -> "Camera":
        return Camera(hsize, vsize, field_of_view)

    @staticmethod
    def _from_desc(d: L.CameraDesc) -> "Camera":
        c = Camera.__new__(Camera)
        c.hsize, c.vsize, c.field_of_view = d.hsize, d.vsize, d.field_of_view
        c.transform = Matrix4.from_flat(list(d.transform))
        c.pixel_size, c.half_width, c.half_height = d.pixel_size, d.half_width, d.half_height
        return c

    def _desc(self) -> L.CameraDesc:
        d = L.CameraDesc()
        d.hsize, d.vsize, d.field_of_view = self.hsize, self.vsize, self.field_of_view
        d.transform[:] = self.transform.flat()
        d.pixel_size, d.half_width, d.half_height = self.pixel_size, self.half_width, self.half_height
        return d

    def ray_for_pixel(self, px: int, py: int) -> Ray:
        """Camera::ray_for_pixel (camera.rs:39-54), evaluated by the device code."""
        return self.rays_for_pixels([px], [py])[0]

    def rays_for_pixels(self, px: Sequence[int], py: Sequence[int]) -> List[Ray]:
        n = len(px)
        out = np.zeros((n, 6), dtype=np.float64)
        d = self._desc()
        L.check(L.lib().rt_ray_for_pixel(C.byref(d), (C.c_uint32 * n)(*px), (C.c_uint32 * n)(*py), n,
                                         out.ctypes.data_as(C.POINTER(C.c_double)), -1))
        return [Ray(tuple(r[:3]), tuple(r[3:])) for r in out.tolist()]

    def render(self, world: World) -> Canvas:
        """Camera::render (camera.rs:56-67)"""
        px, _ = world.scene().render(self)
        return Canvas(self.hsize, self.vsize, px)


class SceneFile:
    """scenes/loader.rs:21-26, parsed by the library's own TOML reader."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle

    def __del__(self):
        try:
            if self._h and self._h.value:
                L.lib().rt_scene_file_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def from_file(path: str) -> "SceneFile":
        """SceneFile::from_file (loader.rs:190-194); raises RtError (RT_ERR_IO / RT_ERR_PARSE) where the reference
        returns Err."""
        h = C.c_void_p()
        L.check(L.lib().rt_scene_file_load(path.encode(), C.byref(h)))
        return SceneFile(h)

    @staticmethod
    def from_str(text: str) -> "SceneFile":
        h = C.c_void_p()
        b = text.encode()
        L.check(L.lib().rt_scene_file_parse(b, len(b), C.byref(h)))
        return SceneFile(h)

    def build_camera(self, width: Optional[int] = None, height: Optional[int] = None) -> Camera:
        """SceneFile::build_camera (loader.rs:196-212).  width/height, when given, replace the file's two integers
        (the reference reads them from the TOML only)."""
        d = L.CameraDesc()
        L.check(L.lib().rt_scene_file_build_camera(self._h, C.byref(d)))
        cam = Camera._from_desc(d)
        if width is not None or height is not None:
            resized = Camera(width if width is not None else cam.hsize, height if height is not None else cam.vsize,
                             cam.field_of_view)
            resized.transform = cam.transform
            cam = resized
        return cam

    def build_world(self) -> World:
        """SceneFile::build_world (loader.rs:214-247)"""
        wd = L.WorldDesc()
        L.check(L.lib().rt_scene_file_build_world(self._h, C.byref(wd)))
        w = World()
        w._loaded = (wd, self)
        w.light_source = PointLight(tuple(wd.light_position), tuple(wd.light_intensity))
        w.objects = _objects_from_desc(wd)
        return w


def _objects_from_desc(wd: L.WorldDesc) -> List[Object]:
    def take(i: int) -> Tuple[Object, int]:
        d = wd.objects[i]
        o = Object(d.kind, Matrix4.from_flat(list(d.transformation)), Material._from_desc(d.material), d.minimum,
                   d.maximum, bool(d.closed))
        o.world_transformation = Matrix4.from_flat(list(d.world_transformation))
        i += 1
        for _ in range(d.n_children):
            ch, i = take(i)
            o.children.append(ch)
        return o, i

    out, i = [], 0
    while i < wd.n_objects:
        o, i = take(i)
        out.append(o)
    return out
