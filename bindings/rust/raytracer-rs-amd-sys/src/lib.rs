//! Raw declarations of `include/raytracer_rs_amd.h` (ABI version 3).  Layouts are `#[repr(C)]` mirrors of the C
//! structs, field for field; every function returns `RT_OK` (0) or a negative `RT_ERR_*` code and leaves a message for
//! `rt_last_error()`.  Not compiled in the repository's build image (no Rust toolchain there).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: c_int = 4;
pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID_ARGUMENT: c_int = -1;
pub const RT_ERR_NOT_INVERTIBLE: c_int = -2;
pub const RT_ERR_MATERIAL: c_int = -3;
pub const RT_ERR_HIP: c_int = -4;
pub const RT_ERR_IO: c_int = -5;
pub const RT_ERR_PARSE: c_int = -6;
pub const RT_ERR_UNSUPPORTED: c_int = -7;
pub const RT_ERR_NO_DEVICE: c_int = -8;
pub const RT_ERR_CAPACITY: c_int = -9;

pub const RT_SPHERE: i32 = 0;
pub const RT_PLANE: i32 = 1;
pub const RT_CUBE: i32 = 2;
pub const RT_CYLINDER: i32 = 3;
pub const RT_CONE: i32 = 4;
pub const RT_GROUP: i32 = 5;
pub const RT_PATTERN_NONE: i32 = -1;
pub const RT_PATTERN_STRIPE: i32 = 0;
pub const RT_PATTERN_GRADIENT: i32 = 1;
pub const RT_PATTERN_RING: i32 = 2;
pub const RT_PATTERN_CHECKERS: i32 = 3;

pub const RT_MAX_GROUP_NESTING: u32 = 16;
pub const RT_DEFAULT_DEPTH: i32 = 5;
pub const RT_MAX_DEPTH: i32 = 8;
pub const RT_RENDER_TIMED: u32 = 1;
pub const RT_RENDER_QUANTIZED: u32 = 2;
pub const RT_RENDER_EXACT_DEPTH: u32 = 4;
pub const RT_NOALLOC_MAX_TILES: u32 = 524_288;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_pattern_desc {
    pub kind: i32,
    pub _pad: i32,
    pub transform: [f64; 16],
    pub color_a: [f64; 3],
    pub color_b: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_material_desc {
    pub color: [f64; 3],
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflectivity: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pattern: rt_pattern_desc,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_object_desc {
    pub kind: i32,
    pub n_children: i32,
    pub closed: i32,
    pub _pad: i32,
    pub minimum: f64,
    pub maximum: f64,
    pub transformation: [f64; 16],
    pub world_transformation: [f64; 16],
    pub material: rt_material_desc,
}

#[repr(C)]
pub struct rt_world_desc {
    pub objects: *const rt_object_desc,
    pub n_objects: u32,
    pub has_light: i32,
    pub light_position: [f64; 3],
    pub light_intensity: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_camera_desc {
    pub hsize: u32,
    pub vsize: u32,
    pub field_of_view: f64,
    pub transform: [f64; 16],
    pub pixel_size: f64,
    pub half_width: f64,
    pub half_height: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub rays: u64,
    pub primary: u64,
    pub shadow: u64,
    pub reflect: u64,
    pub refract: u64,
    pub prim_tests: [u64; 5],
    pub prim_hits: [u64; 5],
    pub aabb_tests: u64,
    pub ray_transforms: u64,
    pub shades: u64,
    pub lit: u64,
    pub specular: u64,
    pub patterns: u64,
    pub refract_setups: u64,
    pub schlick: u64,
    pub kernel_ms: f64,
    pub gather_ms: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_render_opts {
    pub row_begin: u32,
    pub row_end: u32,
    pub max_depth: i32,
    pub collect_stats: i32,
    pub stream: *mut c_void,
    pub band_rows: u32,
    pub band_first: u32,
    pub band_step: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_scene_info {
    pub n_objects: u32,
    pub n_leaves: u32,
    pub n_groups: u32,
    pub max_nesting: u32,
    pub device_bytes: u64,
    pub device: i32,
    pub kernel_build: i32,
}

pub const RT_GATHER_NONE: i32 = 0;
pub const RT_GATHER_RCCL: i32 = 1;
pub const RT_GATHER_COPY: i32 = 2;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_multi_info {
    pub n_workers: u32,
    pub n_devices: u32,
    pub gather: i32,
    pub rccl_ranks: u32,
    pub band_rows: u32,
    pub n_units: u32,
    pub note: [c_char; 160],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_computations {
    pub time: f64,
    pub object: i32,
    pub inside: i32,
    pub point: [f64; 3],
    pub eye_vector: [f64; 3],
    pub normal_vector: [f64; 3],
    pub reflect_vector: [f64; 3],
    pub over_point: [f64; 3],
    pub under_point: [f64; 3],
    pub n1: f64,
    pub n2: f64,
}

#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rt_scene_file {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rt_multi {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_abi_version() -> c_int;
    pub fn rt_device_count(count: *mut c_int) -> c_int;

    pub fn rt_scene_create(world: *const rt_world_desc, device: c_int, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_get_info(scene: *const rt_scene, out: *mut rt_scene_info) -> c_int;

    pub fn rt_render(scene: *const rt_scene, camera: *const rt_camera_desc, opts: *const rt_render_opts, rgb: *mut f64, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_device(scene: *const rt_scene, camera: *const rt_camera_desc, opts: *const rt_render_opts, rgb_device: *mut f64, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_device_noalloc(scene: *const rt_scene, camera: *const rt_camera_desc, opts: *const rt_render_opts, rgb_device: *mut f64) -> c_int;
    pub fn rt_scene_render_times(scene: *const rt_scene, back: u32, ordering_ms: *mut f64, render_ms: *mut f64) -> c_int;

    pub fn rt_multi_create(world: *const rt_world_desc, devices: *const c_int, n_devices: u32, out: *mut *mut rt_multi) -> c_int;
    pub fn rt_multi_destroy(multi: *mut rt_multi);
    pub fn rt_multi_render_device(multi: *mut rt_multi, camera: *const rt_camera_desc, opts: *const rt_render_opts, frame_device: *mut c_void, stats: *mut rt_stats) -> c_int;
    pub fn rt_multi_render(multi: *mut rt_multi, camera: *const rt_camera_desc, opts: *const rt_render_opts, rgb: *mut f64, stats: *mut rt_stats) -> c_int;
    pub fn rt_multi_units_taken(multi: *const rt_multi, per_worker: *mut u32, n: u32) -> c_int;
    pub fn rt_multi_worker_ms(multi: *const rt_multi, per_worker: *mut f64, n: u32) -> c_int;
    pub fn rt_multi_get_info(multi: *const rt_multi, out: *mut rt_multi_info) -> c_int;

    pub fn rt_color_at(scene: *const rt_scene, rays: *const f64, n: u32, remaining: i32, colors: *mut f64, stats: *mut rt_stats) -> c_int;
    pub fn rt_intersect(scene: *const rt_scene, rays: *const f64, n: u32, t: *mut f64, object: *mut i32) -> c_int;
    pub fn rt_intersect_all(scene: *const rt_scene, rays: *const f64, n: u32, capacity: u32, count: *mut u32, t: *mut f64, object: *mut i32) -> c_int;
    pub fn rt_prepare_computations(scene: *const rt_scene, rays: *const f64, t: *const f64, object: *const i32, n: u32, xs_begin: *const u32, xs_t: *const f64, xs_object: *const i32, out: *mut rt_computations) -> c_int;
    pub fn rt_shade_hit(scene: *const rt_scene, comps: *const rt_computations, n: u32, remaining: i32, colors: *mut f64) -> c_int;
    pub fn rt_reflected_color(scene: *const rt_scene, comps: *const rt_computations, n: u32, remaining: i32, colors: *mut f64) -> c_int;
    pub fn rt_refracted_color(scene: *const rt_scene, comps: *const rt_computations, n: u32, remaining: i32, colors: *mut f64) -> c_int;
    pub fn rt_is_shadowed(scene: *const rt_scene, points: *const f64, n: u32, shadowed: *mut u8) -> c_int;
    pub fn rt_normal_at(scene: *const rt_scene, object: *const i32, points: *const f64, n: u32, normals: *mut f64) -> c_int;
    pub fn rt_ray_for_pixel(camera: *const rt_camera_desc, px: *const u32, py: *const u32, n: u32, rays: *mut f64, device: c_int) -> c_int;

    pub fn rt_camera_new(hsize: u32, vsize: u32, field_of_view: f64, out: *mut rt_camera_desc) -> c_int;
    pub fn rt_view_transform(from: *const f64, to: *const f64, up: *const f64, out: *mut f64) -> c_int;

    pub fn rt_canvas_to_ppm(rgb: *const f64, width: u32, height: u32, out: *mut c_char, cap: usize) -> usize;
    pub fn rt_canvas_write_ppm(rgb: *const f64, width: u32, height: u32, path: *const c_char) -> c_int;
    pub fn rt_canvas_to_jpeg(rgb: *const f64, width: u32, height: u32, out: *mut u8, cap: usize) -> usize;
    pub fn rt_canvas_write_jpeg(rgb: *const f64, width: u32, height: u32, path: *const c_char) -> c_int;
    pub fn rt_canvas_quantize(rgb: *const f64, n: usize, out: *mut u8) -> c_int;
    pub fn rt_canvas_quantize_device(rgb_device: *const f64, n: usize, out_device: *mut u8, stream: *mut c_void) -> c_int;

    pub fn rt_scene_file_load(path: *const c_char, out: *mut *mut rt_scene_file) -> c_int;
    pub fn rt_scene_file_parse(text: *const c_char, len: usize, out: *mut *mut rt_scene_file) -> c_int;
    pub fn rt_scene_file_free(file: *mut rt_scene_file);
    pub fn rt_scene_file_build_camera(file: *const rt_scene_file, out: *mut rt_camera_desc) -> c_int;
    pub fn rt_scene_file_build_world(file: *mut rt_scene_file, out: *mut rt_world_desc) -> c_int;
}
