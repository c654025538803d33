// Scene loader: TOML -> SceneFile -> (Camera, World) descriptions, after scenes/loader.rs of the reference.
#include "rt_loader.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rt_math.hpp"
#include "rt_toml.hpp"

namespace rt {
namespace {

constexpr double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI

using toml::Value;

struct Reader {
    std::string &err;
    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }

    bool f64(const Value &tbl, const char *key, const std::string &where, bool required, std::optional<double> &out) {
        const Value *v = tbl.get(key);
        if (!v) return required ? fail("missing field `" + std::string(key) + "` in " + where) : true;
        if (!v->is_number())
            return fail("invalid type: " + std::string(v->type_name()) + ", expected f64 for `" + key + "` in " + where);
        out = v->as_double();
        return true;
    }
    bool f64_or(const Value &tbl, const char *key, const std::string &where, double &out) {
        std::optional<double> o;
        if (!f64(tbl, key, where, false, o)) return false;
        if (o) out = *o;
        return true;
    }
    bool u32(const Value &tbl, const char *key, const std::string &where, uint32_t &out) {
        const Value *v = tbl.get(key);
        if (!v) return fail("missing field `" + std::string(key) + "` in " + where);
        if (v->type != Value::INTEGER)
            return fail("invalid type: " + std::string(v->type_name()) + ", expected u32 for `" + key + "` in " + where);
        if (v->i < 0 || v->i > 4294967295LL) return fail("invalid value: out of range for u32 `" + std::string(key) + "` in " + where);
        out = (uint32_t)v->i;
        return true;
    }
    bool vec3(const Value &tbl, const char *key, const std::string &where, bool required,
              std::optional<std::array<double, 3>> &out) {
        const Value *v = tbl.get(key);
        if (!v) return required ? fail("missing field `" + std::string(key) + "` in " + where) : true;
        if (v->type != Value::ARRAY)
            return fail("invalid type: " + std::string(v->type_name()) + ", expected an array of length 3 for `" + key + "` in " + where);
        if (v->array.size() != 3)
            return fail("invalid length " + std::to_string(v->array.size()) + ", expected an array of length 3 for `" + key + "` in " + where);
        std::array<double, 3> a{};
        for (int i = 0; i < 3; ++i) {
            if (!v->array[i]->is_number())
                return fail("invalid type: " + std::string(v->array[i]->type_name()) + ", expected f64 in `" + key + "` of " + where);
            a[i] = v->array[i]->as_double();
        }
        out = a;
        return true;
    }
    bool vec3_req(const Value &tbl, const char *key, const std::string &where, std::array<double, 3> &out) {
        std::optional<std::array<double, 3>> o;
        if (!vec3(tbl, key, where, true, o)) return false;
        out = *o;
        return true;
    }

    bool transform(const Value &tbl, const std::string &where, TransformConfig &t) {
        return vec3(tbl, "translate", where, false, t.translate) && vec3(tbl, "scale", where, false, t.scale) &&
               f64(tbl, "rotate_x", where, false, t.rotate_x) && f64(tbl, "rotate_y", where, false, t.rotate_y) &&
               f64(tbl, "rotate_z", where, false, t.rotate_z);
    }

    bool pattern(const Value &tbl, const std::string &where, PatternConfig &p) {
        const Value *ty = tbl.get("type");
        if (!ty) return fail("missing field `type` in " + where);
        if (ty->type != Value::STRING) return fail("invalid type: expected a string for `type` in " + where);
        if (ty->s == "stripe") p.kind = RT_PATTERN_STRIPE;
        else if (ty->s == "gradient") p.kind = RT_PATTERN_GRADIENT;
        else if (ty->s == "ring") p.kind = RT_PATTERN_RING;
        else if (ty->s == "checkers") p.kind = RT_PATTERN_CHECKERS;
        else return fail("unknown variant `" + ty->s + "`, expected one of `stripe`, `gradient`, `ring`, `checkers` in " + where);
        return vec3_req(tbl, "color_a", where, p.color_a) && vec3_req(tbl, "color_b", where, p.color_b) &&
               transform(tbl, where, p.transform);
    }

    bool material(const Value &tbl, const std::string &where, MaterialConfig &m) {
        std::optional<std::array<double, 3>> c;
        if (!vec3(tbl, "color", where, false, c)) return false;
        if (c) m.color = *c;
        if (!(f64_or(tbl, "ambient", where, m.ambient) && f64_or(tbl, "diffuse", where, m.diffuse) &&
              f64_or(tbl, "specular", where, m.specular) && f64_or(tbl, "shininess", where, m.shininess) &&
              f64_or(tbl, "reflectivity", where, m.reflectivity) && f64_or(tbl, "transparency", where, m.transparency) &&
              f64_or(tbl, "refractive_index", where, m.refractive_index)))
            return false;
        if (const Value *p = tbl.get("pattern")) {
            if (p->type != Value::TABLE) return fail("invalid type: expected a table for `pattern` in " + where);
            PatternConfig pc;
            if (!pattern(*p, where + ".pattern", pc)) return false;
            m.pattern = pc;
        }
        return true;
    }

    // `depth`: nesting level of the object (0 = the world's own list).  Deeper than the renderer walks is refused
    // here: recursing through a file that nests `children` without end would overflow the stack.
    bool object(const Value &tbl, const std::string &where, ObjectConfig &o, int depth = 0) {
        if (depth > RT_MAX_GROUP_NESTING) return fail("groups are nested too deep (more than " + std::to_string(RT_MAX_GROUP_NESTING) + " levels) at " + where);
        if (tbl.type != Value::TABLE) return fail("invalid type: expected a table for " + where);
        const Value *ty = tbl.get("type");
        if (!ty) return fail("missing field `type` in " + where);
        if (ty->type != Value::STRING) return fail("invalid type: expected a string for `type` in " + where);
        if (ty->s == "sphere") o.kind = RT_SPHERE;
        else if (ty->s == "plane") o.kind = RT_PLANE;
        else if (ty->s == "cube") o.kind = RT_CUBE;
        else if (ty->s == "cylinder") o.kind = RT_CYLINDER;
        else if (ty->s == "cone") o.kind = RT_CONE;
        else if (ty->s == "group") o.kind = RT_GROUP;
        else return fail("unknown variant `" + ty->s + "`, expected one of `sphere`, `plane`, `cube`, `cylinder`, `cone`, `group` in " + where);
        if (const Value *t = tbl.get("transform")) {
            if (t->type != Value::TABLE) return fail("invalid type: expected a table for `transform` in " + where);
            TransformConfig tc;
            if (!transform(*t, where + ".transform", tc)) return false;
            o.transform = tc;
        }
        if (const Value *m = tbl.get("material")) {
            if (m->type != Value::TABLE) return fail("invalid type: expected a table for `material` in " + where);
            MaterialConfig mc;
            if (!material(*m, where + ".material", mc)) return false;
            o.material = mc;
        }
        if (o.kind == RT_CYLINDER || o.kind == RT_CONE) {
            if (!f64(tbl, "minimum", where, false, o.minimum) || !f64(tbl, "maximum", where, false, o.maximum)) return false;
            if (const Value *c = tbl.get("closed")) {
                if (c->type != Value::BOOLEAN) return fail("invalid type: expected a boolean for `closed` in " + where);
                o.closed = c->b;
            }
        }
        if (o.kind == RT_GROUP) {
            if (const Value *ch = tbl.get("children")) {
                if (ch->type != Value::ARRAY) return fail("invalid type: expected a sequence for `children` in " + where);
                o.children.resize(ch->array.size());
                for (size_t i = 0; i < ch->array.size(); ++i)
                    if (!object(*ch->array[i], where + ".children[" + std::to_string(i) + "]", o.children[i], depth + 1)) return false;
            }
        }
        return true;
    }
};

// build_transform (loader.rs:372-396): T * Rz * Ry * Rx * S, each factor only when its key is present
Matrix4 build_transform(const std::optional<TransformConfig> &config) {
    Matrix4 matrix = Matrix4::identity();
    if (config) {
        const TransformConfig &t = *config;
        if (t.translate) matrix = matrix * Matrix4::translate((*t.translate)[0], (*t.translate)[1], (*t.translate)[2]);
        if (t.rotate_z) matrix = matrix * Matrix4::rotate_z(*t.rotate_z * PI / 180.0);
        if (t.rotate_y) matrix = matrix * Matrix4::rotate_y(*t.rotate_y * PI / 180.0);
        if (t.rotate_x) matrix = matrix * Matrix4::rotate_x(*t.rotate_x * PI / 180.0);
        if (t.scale) matrix = matrix * Matrix4::scale((*t.scale)[0], (*t.scale)[1], (*t.scale)[2]);
    }
    return matrix;
}

void default_material(rt_material_desc &m) {  // scene/materials.rs:35-45
    std::memset(&m, 0, sizeof m);
    m.color[0] = m.color[1] = m.color[2] = 1.0;
    m.ambient = 0.1; m.diffuse = 0.9; m.specular = 0.9; m.shininess = 200.0;
    m.reflectivity = 0.0; m.transparency = 0.0; m.refractive_index = 1.0;
    m.pattern.kind = RT_PATTERN_NONE;
    Matrix4::identity().to_array(m.pattern.transform);
}

// build_material (loader.rs:398-422) + build_pattern (loader.rs:424-465)
bool build_material(const MaterialConfig &c, rt_material_desc &m, std::string &err) {
    if (c.ambient < 0.0 || c.diffuse < 0.0 || c.specular < 0.0 || c.shininess < 0.0) {  // scene/materials.rs:94-96
        err = "Material values must be positive";
        return false;
    }
    default_material(m);
    for (int i = 0; i < 3; ++i) m.color[i] = c.color[i];
    m.ambient = c.ambient; m.diffuse = c.diffuse; m.specular = c.specular; m.shininess = c.shininess;
    m.reflectivity = c.reflectivity; m.transparency = c.transparency; m.refractive_index = c.refractive_index;
    if (c.pattern) {
        m.pattern.kind = c.pattern->kind;
        for (int i = 0; i < 3; ++i) { m.pattern.color_a[i] = c.pattern->color_a[i]; m.pattern.color_b[i] = c.pattern->color_b[i]; }
        build_transform(std::optional<TransformConfig>(c.pattern->transform)).to_array(m.pattern.transform);
    }
    return true;
}

// build_object_with_material (loader.rs:255-356): appends the object and its subtree in pre-order
bool build_object(const ObjectConfig &cfg, const rt_material_desc *parent_material, std::vector<rt_object_desc> &out,
                  std::string &err) {
    rt_object_desc o;
    std::memset(&o, 0, sizeof o);
    o.kind = cfg.kind;
    build_transform(cfg.transform).to_array(o.transformation);
    Matrix4::identity().to_array(o.world_transformation);
    if (cfg.material) {  // build_material_with_inheritance (loader.rs:359-370)
        if (!build_material(*cfg.material, o.material, err)) return false;
    } else if (parent_material) {
        o.material = *parent_material;
    } else {
        default_material(o.material);
    }
    o.minimum = cfg.minimum.value_or(-INFINITY);  // loader.rs:309-311,328-330
    o.maximum = cfg.maximum.value_or(INFINITY);
    o.closed = cfg.closed.value_or(false) ? 1 : 0;
    o.n_children = cfg.kind == RT_GROUP ? (int32_t)cfg.children.size() : 0;
    size_t self = out.size();
    out.push_back(o);
    if (cfg.kind == RT_GROUP) {
        const rt_material_desc group_material = out[self].material;
        for (const ObjectConfig &child : cfg.children)
            if (!build_object(child, &group_material, out, err)) return false;
    }
    return true;
}

// Cached world transformations as build_world leaves them (loader.rs:229-241 with geometry/groups.rs:85-105 and
// rendering/objects.rs:253-290): a top-level object keeps its own transformation; the children of a top-level
// group get (identity * group) * child; below that every object gets parent_world * local.
size_t assign_world(std::vector<rt_object_desc> &objs, size_t idx, const Matrix4 &world) {
    world.to_array(objs[idx].world_transformation);
    size_t next = idx + 1;
    if (objs[idx].kind == RT_GROUP)
        for (int32_t k = 0; k < objs[idx].n_children; ++k) {
            Matrix4 child_world = world * Matrix4::from_array(objs[next].transformation);
            next = assign_world(objs, next, child_world);
        }
    return next;
}

}  // namespace

void camera_new(uint32_t hsize, uint32_t vsize, double field_of_view, rt_camera_desc &out) {  // camera.rs:19-37
    double half_view = std::tan(field_of_view / 2.0);
    double aspect = (double)hsize / (double)vsize;
    double half_width, half_height;
    if (aspect > 1.0) {
        half_width = half_view;
        half_height = half_view / aspect;
    } else {
        half_width = half_view * aspect;
        half_height = half_view;
    }
    out.hsize = hsize;
    out.vsize = vsize;
    out.field_of_view = field_of_view;
    Matrix4::identity().to_array(out.transform);
    out.pixel_size = (half_width * 2.0) / (double)hsize;
    out.half_width = half_width;
    out.half_height = half_height;
}

int SceneFile::from_file(const std::string &path, SceneFile &out, std::string &err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        err = "cannot read '" + path + "': " + std::strerror(errno);
        return RT_ERR_IO;
    }
    std::ostringstream ss;
    ss << f.rdbuf();
    return from_str(ss.str(), out, err);
}

int SceneFile::from_str(const std::string &text, SceneFile &out, std::string &err) {
    std::string perr;
    auto doc = toml::parse(text, perr);
    if (!doc) {
        err = "TOML parse error: " + perr;
        return RT_ERR_PARSE;
    }
    out = SceneFile{};
    err.clear();
    Reader r{err};
    const Value *cam = doc->get("camera");
    if (!cam) { err = "missing field `camera`"; return RT_ERR_PARSE; }
    if (cam->type != Value::TABLE) { err = "invalid type: expected a table for `camera`"; return RT_ERR_PARSE; }
    std::optional<double> fov;
    if (!(r.u32(*cam, "width", "camera", out.camera.width) && r.u32(*cam, "height", "camera", out.camera.height) &&
          r.f64(*cam, "fov", "camera", true, fov) && r.vec3_req(*cam, "from", "camera", out.camera.from) &&
          r.vec3_req(*cam, "to", "camera", out.camera.to) && r.vec3_req(*cam, "up", "camera", out.camera.up)))
        return RT_ERR_PARSE;
    out.camera.fov = *fov;
    const Value *light = doc->get("light");
    if (!light) { err = "missing field `light`"; return RT_ERR_PARSE; }
    if (light->type != Value::TABLE) { err = "invalid type: expected a table for `light`"; return RT_ERR_PARSE; }
    if (!(r.vec3_req(*light, "position", "light", out.light.position) &&
          r.vec3_req(*light, "intensity", "light", out.light.intensity)))
        return RT_ERR_PARSE;
    const Value *objs = doc->get("objects");
    if (!objs) { err = "missing field `objects`"; return RT_ERR_PARSE; }
    if (objs->type != Value::ARRAY) { err = "invalid type: expected a sequence for `objects`"; return RT_ERR_PARSE; }
    out.objects.resize(objs->array.size());
    for (size_t i = 0; i < objs->array.size(); ++i)
        if (!r.object(*objs->array[i], "objects[" + std::to_string(i) + "]", out.objects[i])) return RT_ERR_PARSE;
    return RT_OK;
}

void SceneFile::build_camera(rt_camera_desc &out) const {
    camera_new(camera.width, camera.height, camera.fov * PI / 180.0, out);
    view_transform(Vec3{camera.from[0], camera.from[1], camera.from[2]}, Vec3{camera.to[0], camera.to[1], camera.to[2]},
                   Vec3{camera.up[0], camera.up[1], camera.up[2]})
        .to_array(out.transform);
}

int SceneFile::build_world(std::vector<rt_object_desc> &objs, rt_world_desc &out, std::string &err) const {
    objs.clear();
    for (const ObjectConfig &cfg : objects)
        if (!build_object(cfg, nullptr, objs, err)) return RT_ERR_MATERIAL;
    size_t idx = 0;
    while (idx < objs.size()) {
        Matrix4 T = Matrix4::from_array(objs[idx].transformation);
        if (objs[idx].kind == RT_GROUP) {
            // obj.set_world_transform(T) then group.rebuild_children_transforms(identity)
            T.to_array(objs[idx].world_transformation);
            Matrix4 pg = Matrix4::identity() * T;
            size_t next = idx + 1;
            for (int32_t k = 0; k < objs[idx].n_children; ++k) {
                Matrix4 child_world = pg * Matrix4::from_array(objs[next].transformation);
                next = assign_world(objs, next, child_world);
            }
            idx = next;
        } else {
            idx = assign_world(objs, idx, T);
        }
    }
    std::memset(&out, 0, sizeof out);
    out.objects = objs.data();
    out.n_objects = (uint32_t)objs.size();
    out.has_light = 1;  // loader.rs:243-246 always sets a light
    for (int i = 0; i < 3; ++i) {
        out.light_position[i] = light.position[i];
        out.light_intensity[i] = light.intensity[i];
    }
    return RT_OK;
}

}  // namespace rt
