//! `Camera::render(&World) -> Canvas` (rendering/camera.rs:56-67 of raytracer-rs) on an MI355X.
//!
//! Needs two one-line changes in the reference crate: `#[repr(C)]` on `Color` (core/color.rs:4-9), so that
//! `Canvas.pixels: Vec<Color>` is the `double[h][w][3]` the library fills, and `pub` access to `Matrix4.data`
//! (core/matrices.rs:7-9, a private field) for `row_major`.  The reference's package is called `raytracer`
//! (Cargo.toml:2).  Not compiled in the repository's build image (no Rust toolchain); every item used below was
//! checked by hand against the reference's declarations: `Camera` fields (rendering/camera.rs:8-16, u32 sizes),
//! `Canvas::new(u32, u32)` (rendering/canvas.rs:17), `Group::children() -> &[Object]` (geometry/groups.rs:66; the
//! field itself is private), `Cylinder` / `Cone` `minimum, maximum, closed` (geometry/cylinders.rs:15-23),
//! `HasMaterial::material()`, `Transformable::transformation()` (rendering/objects.rs:23, 29),
//! `Object::world_transformation()` (objects.rs:241), `World { objects, light_source: Option<PointLight> }`
//! (rendering/world.rs:14-17; `PointLight` is `Copy`, scene/lights.rs:4), `Material.pattern: Option<Pattern>`,
//! `Pattern { transform, kind }` and the struct variants of `PatternKind` (scene/patterns.rs:7-28).
use std::ffi::CStr;

use raytracer::core::matrices::Matrix4;
use raytracer::rendering::camera::Camera;
use raytracer::rendering::canvas::Canvas;
use raytracer::rendering::objects::{HasMaterial, Object, Transformable};
use raytracer::rendering::world::World;
use raytracer::scene::materials::Material;
use raytracer::scene::patterns::PatternKind;
use raytracer_rs_amd_sys as sys;

fn row_major(m: &Matrix4) -> [f64; 16] {
    let mut out = [0.0; 16];
    for r in 0..4 {
        for c in 0..4 {
            out[r * 4 + c] = m.data[r][c];
        }
    }
    out
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::rt_last_error()) }.to_string_lossy().into_owned()
}

fn material_desc(m: &Material) -> sys::rt_material_desc {
    let mut pattern = sys::rt_pattern_desc { kind: sys::RT_PATTERN_NONE, _pad: 0, transform: row_major(&Matrix4::identity()),
                                             color_a: [0.0; 3], color_b: [0.0; 3] };
    if let Some(p) = &m.pattern {
        let (kind, a, b) = match &p.kind {
            PatternKind::Stripe { color_a, color_b } => (sys::RT_PATTERN_STRIPE, color_a, color_b),
            PatternKind::Gradient { color_a, color_b } => (sys::RT_PATTERN_GRADIENT, color_a, color_b),
            PatternKind::Ring { color_a, color_b } => (sys::RT_PATTERN_RING, color_a, color_b),
            PatternKind::Checkers { color_a, color_b } => (sys::RT_PATTERN_CHECKERS, color_a, color_b),
            #[allow(unreachable_patterns)]
            _ => panic!("test-only pattern kinds have no device counterpart"),
        };
        pattern = sys::rt_pattern_desc { kind, _pad: 0, transform: row_major(&p.transform),
                                         color_a: [a.red, a.green, a.blue], color_b: [b.red, b.green, b.blue] };
    }
    sys::rt_material_desc { color: [m.color.red, m.color.green, m.color.blue], ambient: m.ambient, diffuse: m.diffuse,
                            specular: m.specular, shininess: m.shininess, reflectivity: m.reflectivity,
                            transparency: m.transparency, refractive_index: m.refractive_index, pattern }
}

/// The world's objects in the reference's own traversal order (depth-first, world.rs:29-32 and 66-68).
fn flatten(obj: &Object, out: &mut Vec<sys::rt_object_desc>) {
    let kind = match obj {
        Object::Sphere(_) => sys::RT_SPHERE, Object::Plane(_) => sys::RT_PLANE, Object::Cube(_) => sys::RT_CUBE,
        Object::Cylinder(_) => sys::RT_CYLINDER, Object::Cone(_) => sys::RT_CONE, Object::Group(_) => sys::RT_GROUP,
    };
    let mut d = sys::rt_object_desc { kind, n_children: 0, closed: 0, _pad: 0, minimum: f64::NEG_INFINITY, maximum: f64::INFINITY,
                                      transformation: row_major(&obj.transformation()),
                                      world_transformation: row_major(&obj.world_transformation()),
                                      material: material_desc(&obj.material()) };
    match obj {
        Object::Cylinder(c) => { d.minimum = c.minimum; d.maximum = c.maximum; d.closed = c.closed as i32; }
        Object::Cone(c) => { d.minimum = c.minimum; d.maximum = c.maximum; d.closed = c.closed as i32; }
        Object::Group(g) => { d.n_children = g.children().len() as i32; }
        _ => {}
    }
    out.push(d);
    if let Object::Group(g) = obj {
        for child in g.children() {
            flatten(child, out);
        }
    }
}

/// A `World` resident on one GPU; keep it while rendering several frames of the same world.
pub struct GpuWorld {
    scene: *mut sys::rt_scene,
}

impl GpuWorld {
    pub fn new(world: &World, device: i32) -> GpuWorld {
        let mut objects = Vec::new();
        for o in &world.objects {
            flatten(o, &mut objects);
        }
        let light = world.light_source;
        let desc = sys::rt_world_desc {
            objects: objects.as_ptr(), n_objects: objects.len() as u32, has_light: light.is_some() as i32,
            light_position: light.map_or([0.0; 3], |l| [l.position.x, l.position.y, l.position.z]),
            light_intensity: light.map_or([0.0; 3], |l| [l.intensity.red, l.intensity.green, l.intensity.blue]),
        };
        let mut scene = std::ptr::null_mut();
        let rc = unsafe { sys::rt_scene_create(&desc, device, &mut scene) };
        // the reference panics (`expect("... should be invertible")`) where the library reports an error
        assert!(rc == sys::RT_OK, "{}", last_error());
        GpuWorld { scene }
    }

    /// Same result as `Camera::render(&self, world)` on the CPU (camera.rs:56-67).
    pub fn render(&self, camera: &Camera) -> Canvas {
        let cam = sys::rt_camera_desc { hsize: camera.hsize, vsize: camera.vsize, field_of_view: camera.field_of_view,
                                        transform: row_major(&camera.transform), pixel_size: camera.pixel_size,
                                        half_width: camera.half_width, half_height: camera.half_height };
        let mut canvas = Canvas::new(camera.hsize, camera.vsize);
        let rc = unsafe { sys::rt_render(self.scene, &cam, std::ptr::null(), canvas.pixels.as_mut_ptr() as *mut f64, std::ptr::null_mut()) };
        assert!(rc == sys::RT_OK, "{}", last_error());
        canvas
    }
}

impl Drop for GpuWorld {
    fn drop(&mut self) {
        unsafe { sys::rt_scene_destroy(self.scene) };
    }
}

/// Drop-in for `camera.render(&world)`.
pub fn render(camera: &Camera, world: &World) -> Canvas {
    GpuWorld::new(world, 0).render(camera)
}
