// C++ mirror of the reference's scene loader (scenes/loader.rs): TOML schema, SceneFile::from_file,
// build_camera, build_world.  Produces the plain descriptions of include/raytracer_rs_amd.h.
#pragma once

#include <array>
#include <optional>
#include <string>
#include <vector>

#include "../../include/raytracer_rs_amd.h"

namespace rt {

struct TransformConfig {  // loader.rs:175-187
    std::optional<std::array<double, 3>> translate, scale;
    std::optional<double> rotate_x, rotate_y, rotate_z;  // degrees
};

struct PatternConfig {  // loader.rs:114-145
    int kind = RT_PATTERN_NONE;
    std::array<double, 3> color_a{}, color_b{};
    TransformConfig transform;  // #[serde(flatten)]: the transform keys live in the pattern table itself
};

struct MaterialConfig {  // loader.rs:93-112, defaults loader.rs:147-173
    std::array<double, 3> color{1.0, 1.0, 1.0};
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    double reflectivity = 0.0, transparency = 0.0, refractive_index = 1.0;
    std::optional<PatternConfig> pattern;
};

struct ObjectConfig {  // loader.rs:44-91
    int kind = RT_SPHERE;
    std::optional<TransformConfig> transform;
    std::optional<MaterialConfig> material;
    std::optional<double> minimum, maximum;
    std::optional<bool> closed;
    std::vector<ObjectConfig> children;
};

struct CameraConfig {  // loader.rs:28-36
    uint32_t width = 0, height = 0;
    double fov = 0.0;
    std::array<double, 3> from{}, to{}, up{};
};

struct LightConfig {  // loader.rs:38-42
    std::array<double, 3> position{}, intensity{};
};

struct SceneFile {  // loader.rs:21-26
    CameraConfig camera;
    LightConfig light;
    std::vector<ObjectConfig> objects;

    // SceneFile::from_file (loader.rs:190-194): RT_ERR_IO / RT_ERR_PARSE with a message in err
    static int from_file(const std::string &path, SceneFile &out, std::string &err);
    static int from_str(const std::string &text, SceneFile &out, std::string &err);
    // SceneFile::build_camera (loader.rs:196-212)
    void build_camera(rt_camera_desc &out) const;
    // SceneFile::build_world (loader.rs:214-247): objects in depth-first pre-order
    int build_world(std::vector<rt_object_desc> &objects, rt_world_desc &out, std::string &err) const;
};

// Camera::new (rendering/camera.rs:19-37)
void camera_new(uint32_t hsize, uint32_t vsize, double field_of_view, rt_camera_desc &out);

}  // namespace rt
