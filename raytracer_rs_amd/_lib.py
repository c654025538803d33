"""ctypes view of the C ABI in include/raytracer_rs_amd.h (raytracer_rs_amd/librt_amd.so).

The library is the product: there is no Python or CPU rendering fallback.  If the shared object is missing, or no HIP
device is present, the calls fail loudly (`RtError`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB points a developer run at another build of the same library (A/B timing of kernel variants).
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

RT_OK = 0
ERR_NAMES = {-1: "RT_ERR_INVALID_ARGUMENT", -2: "RT_ERR_NOT_INVERTIBLE", -3: "RT_ERR_MATERIAL", -4: "RT_ERR_HIP",
             -5: "RT_ERR_IO", -6: "RT_ERR_PARSE", -7: "RT_ERR_UNSUPPORTED", -8: "RT_ERR_NO_DEVICE", -9: "RT_ERR_CAPACITY"}
SPHERE, PLANE, CUBE, CYLINDER, CONE, GROUP = range(6)
PATTERN_NONE, PATTERN_STRIPE, PATTERN_GRADIENT, PATTERN_RING, PATTERN_CHECKERS = -1, 0, 1, 2, 3
MAX_GROUP_NESTING = 16
DEFAULT_DEPTH = 5
RENDER_TIMED = 1
RENDER_QUANTIZED = 2
RENDER_EXACT_DEPTH = 4
RENDER_IN_PLACE = 8
MAX_DEPTH = 8
NOALLOC_MAX_TILES = 524288


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class PatternDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("transform", C.c_double * 16),
                ("color_a", C.c_double * 3), ("color_b", C.c_double * 3)]


class MaterialDesc(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double),
                ("specular", C.c_double), ("shininess", C.c_double), ("reflectivity", C.c_double),
                ("transparency", C.c_double), ("refractive_index", C.c_double), ("pattern", PatternDesc)]


class ObjectDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_children", C.c_int32), ("closed", C.c_int32), ("_pad", C.c_int32),
                ("minimum", C.c_double), ("maximum", C.c_double), ("transformation", C.c_double * 16),
                ("world_transformation", C.c_double * 16), ("material", MaterialDesc)]


class WorldDesc(C.Structure):
    _fields_ = [("objects", C.POINTER(ObjectDesc)), ("n_objects", C.c_uint32), ("has_light", C.c_int32),
                ("light_position", C.c_double * 3), ("light_intensity", C.c_double * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("field_of_view", C.c_double),
                ("transform", C.c_double * 16), ("pixel_size", C.c_double), ("half_width", C.c_double),
                ("half_height", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary", C.c_uint64), ("shadow", C.c_uint64), ("reflect", C.c_uint64),
                ("refract", C.c_uint64), ("prim_tests", C.c_uint64 * 5), ("prim_hits", C.c_uint64 * 5),
                ("aabb_tests", C.c_uint64), ("ray_transforms", C.c_uint64), ("shades", C.c_uint64),
                ("lit", C.c_uint64), ("specular", C.c_uint64), ("patterns", C.c_uint64),
                ("refract_setups", C.c_uint64), ("schlick", C.c_uint64), ("kernel_ms", C.c_double),
                ("gather_ms", C.c_double)]

    COUNT_FIELDS = ("rays", "primary", "shadow", "reflect", "refract", "prim_tests", "prim_hits", "aabb_tests",
                    "ray_transforms", "shades", "lit", "specular", "patterns", "refract_setups", "schlick")

    def as_dict(self) -> dict:
        d = {}
        for name in self.COUNT_FIELDS:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d


class Computations(C.Structure):
    """struct Computations (rendering/intersections.rs:12-24); `object` indexes the world's depth-first object list."""
    _fields_ = [("time", C.c_double), ("object", C.c_int32), ("inside", C.c_int32), ("point", C.c_double * 3),
                ("eye_vector", C.c_double * 3), ("normal_vector", C.c_double * 3), ("reflect_vector", C.c_double * 3),
                ("over_point", C.c_double * 3), ("under_point", C.c_double * 3), ("n1", C.c_double), ("n2", C.c_double)]


class RenderOpts(C.Structure):
    _fields_ = [("row_begin", C.c_uint32), ("row_end", C.c_uint32), ("max_depth", C.c_int32),
                ("collect_stats", C.c_int32), ("stream", C.c_void_p), ("band_rows", C.c_uint32),
                ("band_first", C.c_uint32), ("band_step", C.c_uint32), ("flags", C.c_uint32)]


class MultiInfo(C.Structure):
    _fields_ = [("n_workers", C.c_uint32), ("n_devices", C.c_uint32), ("gather", C.c_int32), ("rccl_ranks", C.c_uint32),
                ("band_rows", C.c_uint32), ("n_units", C.c_uint32), ("note", C.c_char * 160)]


GATHER_NAMES = {0: "none", 1: "rccl", 2: "copy"}


class SceneInfo(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("n_leaves", C.c_uint32), ("n_groups", C.c_uint32),
                ("max_nesting", C.c_uint32), ("device_bytes", C.c_uint64), ("device", C.c_int32), ("kernel_build", C.c_int32)]


# every symbol include/raytracer_rs_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_int, []),
    "rt_device_count": (C.c_int, [_P(C.c_int)]),
    "rt_scene_create": (C.c_int, [_P(WorldDesc), C.c_int, _P(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_get_info": (C.c_int, [C.c_void_p, _P(SceneInfo)]),
    "rt_render": (C.c_int, [C.c_void_p, _P(CameraDesc), _P(RenderOpts), _P(C.c_double), _P(Stats)]),
    "rt_render_device": (C.c_int, [C.c_void_p, _P(CameraDesc), _P(RenderOpts), C.c_void_p, _P(Stats)]),
    "rt_render_device_noalloc": (C.c_int, [C.c_void_p, _P(CameraDesc), _P(RenderOpts), C.c_void_p]),
    "rt_scene_render_times": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_double), _P(C.c_double)]),
    "rt_multi_create": (C.c_int, [_P(WorldDesc), _P(C.c_int), C.c_uint32, _P(C.c_void_p)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_multi_render_device": (C.c_int, [C.c_void_p, _P(CameraDesc), _P(RenderOpts), C.c_void_p, _P(Stats)]),
    "rt_multi_render": (C.c_int, [C.c_void_p, _P(CameraDesc), _P(RenderOpts), _P(C.c_double), _P(Stats)]),
    "rt_multi_units_taken": (C.c_int, [C.c_void_p, _P(C.c_uint32), C.c_uint32]),
    "rt_multi_worker_ms": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32]),
    "rt_multi_get_info": (C.c_int, [C.c_void_p, _P(MultiInfo)]),
    "rt_color_at": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32, C.c_int32, _P(C.c_double), _P(Stats)]),
    "rt_intersect": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32, _P(C.c_double), _P(C.c_int32)]),
    "rt_intersect_all": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32, C.c_uint32, _P(C.c_uint32), _P(C.c_double),
                                   _P(C.c_int32)]),
    "rt_prepare_computations": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_double), _P(C.c_int32), C.c_uint32,
                                          _P(C.c_uint32), _P(C.c_double), _P(C.c_int32), _P(Computations)]),
    "rt_shade_hit": (C.c_int, [C.c_void_p, _P(Computations), C.c_uint32, C.c_int32, _P(C.c_double)]),
    "rt_reflected_color": (C.c_int, [C.c_void_p, _P(Computations), C.c_uint32, C.c_int32, _P(C.c_double)]),
    "rt_refracted_color": (C.c_int, [C.c_void_p, _P(Computations), C.c_uint32, C.c_int32, _P(C.c_double)]),
    "rt_is_shadowed": (C.c_int, [C.c_void_p, _P(C.c_double), C.c_uint32, _P(C.c_uint8)]),
    "rt_normal_at": (C.c_int, [C.c_void_p, _P(C.c_int32), _P(C.c_double), C.c_uint32, _P(C.c_double)]),
    "rt_ray_for_pixel": (C.c_int, [_P(CameraDesc), _P(C.c_uint32), _P(C.c_uint32), C.c_uint32, _P(C.c_double), C.c_int]),
    "rt_camera_new": (C.c_int, [C.c_uint32, C.c_uint32, C.c_double, _P(CameraDesc)]),
    "rt_view_transform": (C.c_int, [_P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "rt_canvas_to_ppm": (C.c_size_t, [_P(C.c_double), C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "rt_canvas_write_ppm": (C.c_int, [_P(C.c_double), C.c_uint32, C.c_uint32, C.c_char_p]),
    "rt_canvas_to_jpeg": (C.c_size_t, [_P(C.c_double), C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "rt_canvas_write_jpeg": (C.c_int, [_P(C.c_double), C.c_uint32, C.c_uint32, C.c_char_p]),
    "rt_canvas_quantize": (C.c_int, [_P(C.c_double), C.c_size_t, _P(C.c_uint8)]),
    "rt_canvas_quantize_device": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rt_scene_file_load": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "rt_scene_file_parse": (C.c_int, [C.c_char_p, C.c_size_t, _P(C.c_void_p)]),
    "rt_scene_file_free": (None, [C.c_void_p]),
    "rt_scene_file_build_camera": (C.c_int, [C.c_void_p, _P(CameraDesc)]),
    "rt_scene_file_build_world": (C.c_int, [C.c_void_p, _P(WorldDesc)]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Loads librt_amd.so (built by raytracer_rs_amd/csrc/Makefile).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(-4, f"{LIB_PATH} is missing: build it with `make -C raytracer_rs_amd/csrc` "
                              f"(or __graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int) -> None:
    if code != RT_OK:
        msg = lib().rt_last_error()
        raise RtError(code, msg.decode() if msg else "")
