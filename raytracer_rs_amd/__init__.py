"""MI355X (gfx950) renderer for the per-pixel path of raytracer-rs: HIP kernels behind a C ABI
(include/raytracer_rs_amd.h, raytracer_rs_amd/librt_amd.so) plus this host-side mirror of the reference's
`SceneFile / Camera / World / Canvas` API."""
from ._lib import (CONE, CUBE, CYLINDER, DEFAULT_DEPTH, GROUP, MAX_DEPTH, MAX_GROUP_NESTING, PATTERN_CHECKERS,
                   PATTERN_GRADIENT, PATTERN_NONE, PATTERN_RING, PATTERN_STRIPE, PLANE, SPHERE, RtError, Stats)
from .api import (Camera, Canvas, Material, Matrix4, MultiScene, Object, Pattern, PointLight, Ray, Scene, SceneFile, World,
                  quantize_device, view_transform)

__all__ = ["Camera", "Canvas", "Material", "Matrix4", "Object", "Pattern", "PointLight", "Ray", "Scene", "SceneFile",
           "World", "view_transform", "quantize_device", "RtError", "Stats", "SPHERE", "PLANE", "CUBE", "CYLINDER", "CONE", "GROUP",
           "PATTERN_NONE", "PATTERN_STRIPE", "PATTERN_GRADIENT", "PATTERN_RING", "PATTERN_CHECKERS",
           "DEFAULT_DEPTH", "MAX_DEPTH", "MAX_GROUP_NESTING"]
