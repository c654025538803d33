"""Host-side checks that need no GPU: the C ABI library loads and exports every symbol the header declares; the C++
scene loader agrees with an independent TOML reader (tomli) + the oracle's loader; error behaviour; the PPM writer."""
import ctypes as C
import hashlib
import json
import math
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, scene_path

ALL_SCENES = ["chapter_5", "chapter_6", "chapter_7_and_8", "chapter_9", "chapter_10", "chapter_11", "chapter_12",
              "chapter_13", "chapter_14"]


def test_library_exports_every_declared_symbol(rt):
    from raytracer_rs_amd import _lib

    header = open(os.path.join(ROOT, "include", "raytracer_rs_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.rt_abi_version() == 4


def _c_declarations():
    """(functions: name -> number of parameters, structs: name -> field names in order) of the C header."""
    header = open(os.path.join(ROOT, "include", "raytracer_rs_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    funcs = {}
    for m in re.finditer(r"\b(rt_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", header):
        params = m.group(2).strip()
        funcs[m.group(1)] = 0 if params in ("", "void") else len(params.split(","))
    structs = {}
    for m in re.finditer(r"typedef struct (rt_[a-z_]+)\s*\{(.*?)\}\s*\1\s*;", header, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            names[0] = names[0].split()[-1]  # drop the type of the first declarator
            fields += [re.sub(r"\[.*?\]", "", n).strip().lstrip("*") for n in names]
        structs[m.group(1)] = fields
    return funcs, structs


def test_rust_sys_crate_declares_the_header(rt):
    """bindings/rust/raytracer-rs-amd-sys/src/lib.rs cannot be compiled here (no Rust toolchain): check it textually
    against the C header -- every rt_* function with the same number of parameters, every struct #[repr(C)] with the
    same fields in the same order."""
    funcs, structs = _c_declarations()
    assert len(funcs) >= 30 and "rt_render_device_noalloc" in funcs and "rt_multi_render" in funcs
    src = open(os.path.join(ROOT, "bindings", "rust", "raytracer-rs-amd-sys", "src", "lib.rs")).read()
    rust_funcs = {}
    for m in re.finditer(r"pub fn (rt_[a-z0-9_]+)\(([^)]*)\)", src):
        params = m.group(2).strip()
        rust_funcs[m.group(1)] = 0 if not params else len([p for p in params.split(",") if p.strip()])
    assert rust_funcs == funcs, {k: (funcs.get(k), rust_funcs.get(k)) for k in set(funcs) | set(rust_funcs) if funcs.get(k) != rust_funcs.get(k)}
    for name, fields in structs.items():
        m = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct " + name + r"\s*\{(.*?)\}", src, flags=re.S)
        assert m, f"{name} has no #[repr(C)] struct in lib.rs"
        rust_fields = re.findall(r"pub (\w+)\s*:", m.group(1))
        assert rust_fields == fields, (name, fields, rust_fields)
    assert f"RT_ABI_VERSION: c_int = {rt._lib.lib().rt_abi_version()}" in src


def test_struct_sizes_match_the_header(rt):
    """ctypes mirrors vs the C structs: compile a tiny C program against the header and compare sizeof."""
    import subprocess
    import tempfile

    from raytracer_rs_amd import _lib

    names = {"rt_pattern_desc": _lib.PatternDesc, "rt_material_desc": _lib.MaterialDesc,
             "rt_object_desc": _lib.ObjectDesc, "rt_world_desc": _lib.WorldDesc, "rt_camera_desc": _lib.CameraDesc,
             "rt_stats": _lib.Stats, "rt_render_opts": _lib.RenderOpts, "rt_scene_info": _lib.SceneInfo,
             "rt_multi_info": _lib.MultiInfo}
    src = '#include <stdio.h>\n#include "raytracer_rs_amd.h"\nint main(void){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "s.c"), "w") as f:
            f.write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "s"), os.path.join(d, "s.c")],
                       check=True)
        out = subprocess.run([os.path.join(d, "s")], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        n, size = line.split()
        assert C.sizeof(names[n]) == int(size), n


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_loader_matches_independent_reader(rt, orc, scene):
    """SceneFile::from_file / build_camera / build_world in C++ vs tomli + the oracle's restatement of loader.rs:
    every matrix and material value bit-identical."""
    sf = rt.SceneFile.from_file(scene_path(scene))
    cam, world = sf.build_camera(), sf.build_world()
    ocam, oworld = orc.load_scene(scene_path(scene))
    assert (cam.hsize, cam.vsize, cam.field_of_view) == (ocam.hsize, ocam.vsize, ocam.field_of_view)
    assert cam.transform.flat() == [v for row in ocam.transform.rows() for v in row]
    assert (cam.pixel_size, cam.half_width, cam.half_height) == (ocam.pixel_size, ocam.half_width, ocam.half_height)
    wd, _ = world.world_desc()
    assert wd.n_objects == oworld.object_count()
    for i in range(wd.n_objects):
        d = wd.objects[i]
        assert list(d.world_transformation) == [v for row in oworld.world_transform(i).rows() for v in row]
        m, om = d.material, oworld.object_material(i)
        assert tuple(m.color) == om.color.tuple()
        assert (m.ambient, m.diffuse, m.specular, m.shininess, m.reflectivity, m.transparency, m.refractive_index) == \
            (om.ambient, om.diffuse, om.specular, om.shininess, om.reflectivity, om.transparency, om.refractive_index)
        assert m.pattern.kind == om.pattern.kind
        if m.pattern.kind >= 0:
            assert list(m.pattern.transform) == [v for row in om.pattern.transform.rows() for v in row]
            assert tuple(m.pattern.color_a) == om.pattern.a.tuple() and tuple(m.pattern.color_b) == om.pattern.b.tuple()


def test_resolution_override_keeps_view_and_changes_frustum(rt, orc):
    sf = rt.SceneFile.from_file(scene_path("chapter_11"))
    native, big = sf.build_camera(), sf.build_camera(3840, 2160)
    assert native.transform.flat() == big.transform.flat()
    ocam, _ = orc.load_scene(scene_path("chapter_11"), 3840, 2160)
    assert (big.pixel_size, big.half_width, big.half_height) == (ocam.pixel_size, ocam.half_width, ocam.half_height)


MINIMAL = """
[camera]
width = 4
height = 3
fov = 60
from = [0, 0, -5]
to = [0, 0, 0]
up = [0, 1, 0]
[light]
position = [-10, 10, -10]
intensity = [1, 1, 1]
"""


def test_loader_toml_features(rt):
    text = MINIMAL + """
# array of tables, sub-tables attach to the last element; integers where floats are expected; keys in any order
[[objects]]
type = "group"
  [objects.transform]
  rotate_y = 45      # trailing comment
  translate = [1, 2.5e0, -3]
  [objects.material]
  color = [0.5, 0.25, 1]
  reflectivity = 0.5
  [[objects.children]]
  type = 'cylinder'
  minimum = -1
  maximum = 2
  closed = true
  [[objects.children]]
  type = "group"
    [[objects.children.children]]
    type = "cone"
    material = { ambient = 0.3, pattern = { type = "ring", color_a = [1, 0, 0], color_b = [0, 1, 0], scale = [2, 2, 2] } }
[[objects]]
type = "plane"
unknown_key = "ignored"
"""
    sf = rt.SceneFile.from_str(text)
    world = sf.build_world()
    objs = world.objects
    assert [o.kind for o in objs] == [rt.GROUP, rt.PLANE]
    g = objs[0]
    expected = rt.Matrix4.translate(1, 2.5, -3) * rt.Matrix4.rotate_y(45 * math.pi / 180.0)
    assert g.transformation.flat() == (rt.Matrix4.identity() * expected).flat()
    cyl, inner = g.children
    assert (cyl.kind, cyl.minimum, cyl.maximum, cyl.closed) == (rt.CYLINDER, -1.0, 2.0, True)
    assert cyl.material.color == (0.5, 0.25, 1.0) and cyl.material.reflectivity == 0.5  # inherited (loader.rs:359-370)
    cone = inner.children[0]
    assert cone.kind == rt.CONE and cone.minimum == -math.inf and cone.maximum == math.inf and not cone.closed
    # a [material] table that is present but partial takes DEFAULTS for the rest, not the parent's values
    assert cone.material.ambient == 0.3 and cone.material.color == (1.0, 1.0, 1.0) and cone.material.reflectivity == 0.0
    assert cone.material.pattern.kind == rt.PATTERN_RING
    assert cone.material.pattern.transform.flat() == (rt.Matrix4.identity() * rt.Matrix4.scale(2, 2, 2)).flat()
    # world transforms: group itself; children (I*G)*C; grandchildren W_parent*T  (loader.rs:229-241)
    assert cone.world_transformation.flat() == ((rt.Matrix4.identity() * g.transformation) * inner.transformation
                                                * cone.transformation).flat()
    assert objs[1].material == rt.Material()


@pytest.mark.parametrize("text,code,needle", [
    ("[camera]\nwidth = 4\n", -6, "missing field"),
    (MINIMAL, -6, "missing field `objects`"),
    (MINIMAL + '[[objects]]\ntype = "torus"\n', -6, "unknown variant `torus`"),
    (MINIMAL + "[[objects]]\n", -6, "missing field `type`"),
    (MINIMAL.replace("width = 4", "width = 4.5") + '[[objects]]\ntype="sphere"\n', -6, "expected u32"),
    (MINIMAL.replace("fov = 60", 'fov = "wide"') + '[[objects]]\ntype="sphere"\n', -6, "expected f64"),
    (MINIMAL + '[[objects]]\ntype="sphere"\n[objects.transform]\ntranslate=[1,2]\n', -6, "length 3"),
    (MINIMAL + '[[objects]]\ntype="sphere"\n[objects.material]\npattern = { type = "stripe" }\n', -6, "color_a"),
    (MINIMAL + '[[objects]]\ntype = "sphere"\ntype = "plane"\n', -6, "duplicate key"),
    (MINIMAL + '[[objects]]\ntype = "sphere\n', -6, "line"),
    (MINIMAL + "[[objects]]\ntype = 1979-05-27\n", -6, "dates"),
    # hostile nesting must come back as an error, not as a stack overflow
    (MINIMAL + '[[objects]]\ntype = "sphere"\nx = ' + "[" * 100000 + "]" * 100000 + "\n", -6, "nested too deep"),
    (MINIMAL + '[[objects]]\ntype = "group"\nchildren = [' + '{type = "group", children = [' * 40 + "]}" * 40 + "]\n", -6, "nested too deep"),
])
def test_loader_errors_are_results_not_aborts(rt, text, code, needle):
    """SceneFile::from_file returns Err (loader.rs:190-194): here an RtError with RT_ERR_PARSE."""
    with pytest.raises(rt.RtError) as e:
        rt.SceneFile.from_str(text)
    assert e.value.code == code and needle in e.value.message, e.value.message


def test_render_rejects_frames_that_are_not_the_canvas_layout(rt):
    """Scene.render(out=...) hands the array's memory to C: anything but C-contiguous float64 [vsize, hsize, 3] is
    refused before any call is made (no device needed to find that out)."""
    from raytracer_rs_amd import api

    cam = rt.Camera(8, 6, 1.0)
    for bad in (np.zeros((6, 8, 3), dtype=np.float32), np.zeros((8, 6, 3)), np.zeros((6, 8, 4))[:, :, :3], np.zeros((6, 8))):
        with pytest.raises(ValueError):
            api._check_frame(bad, cam)
    api._check_frame(np.zeros((6, 8, 3)), cam)


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    """The scene-file reader, the loader, the world flattener and the P3 writer, built with -fsanitize=address,undefined
    (g++; no HIP involved) and fed the nine scene files, three generated ones and some 3 600 damaged copies of them: every input comes back
    as a result, and the sanitizers stay silent.  (Its first run found a read through a dangling pointer in the
    integer parser.)"""
    import glob
    import subprocess

    src = [os.path.join(ROOT, "tests", "host_sanitize", "sanitize_host.cpp")] + [
        os.path.join(ROOT, "raytracer_rs_amd", "csrc", f) for f in ("rt_toml.cpp", "rt_loader.cpp", "rt_flatten.cpp", "rt_canvas.cpp")]
    exe = str(tmp_path / "sanitize_host")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-static-libasan", "-static-libubsan", "-ffp-contract=off", "-fno-builtin-sin", "-fno-builtin-cos", "-o", exe] + src, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0")
    # besides the reference's scenes: generated ones with long lists, nested groups and every leaf kind, whose order
    # trees and range-skip plans the driver checks record by record
    from raytracer_rs_amd import synthetic
    generated = []
    for name, text in [("hierarchy", synthetic.large_scene(cells=3, leaves_per_subgroup=30)), ("random_a", synthetic.random_scene(5, n_top=20)),
                       ("random_b", synthetic.random_scene(11, n_top=40))]:
        path = tmp_path / (name + ".toml")
        path.write_text(text)
        generated.append(str(path))
    run = subprocess.run([exe] + sorted(glob.glob(os.path.join(GOLDEN, "scenes", "*.toml"))) + generated, capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "sanitized host run" in run.stdout and "ERROR" not in run.stderr


def test_loader_io_error_and_negative_material(rt):
    with pytest.raises(rt.RtError) as e:
        rt.SceneFile.from_file("/nonexistent/scene.toml")
    assert e.value.code == -5
    sf = rt.SceneFile.from_str(MINIMAL + '[[objects]]\ntype="sphere"\n[objects.material]\nambient = -0.1\n')
    with pytest.raises(rt.RtError) as e:  # MaterialBuilder::build panics (materials.rs:94-96)
        sf.build_world()
    assert e.value.code == -3 and "Material values must be positive" in e.value.message


def test_scene_validation_precedes_device_use(rt):
    """rt_scene_create reports what the reference would panic on at render time -- before touching any device."""
    w = rt.World()
    w.objects.append(rt.Object.sphere(transformation=rt.Matrix4.scale(0.0, 1.0, 1.0)))
    with pytest.raises(rt.RtError) as e:
        w.scene()
    assert e.value.code == -2 and "shape transformation should be invertible" in e.value.message
    w = rt.World()
    pat = rt.Pattern.stripe((1, 1, 1), (0, 0, 0))
    pat.set_transform(rt.Matrix4.scale(1.0, 0.0, 1.0))
    w.objects.append(rt.Object.sphere(material=rt.Material(pattern=pat)))
    with pytest.raises(rt.RtError) as e:
        w.scene()
    assert e.value.code == -2 and "pattern transform" in e.value.message
    # nesting deeper than the kernels are built for
    w = rt.World()
    g = rt.Object.group()
    w.objects.append(g)
    for _ in range(rt.MAX_GROUP_NESTING):
        inner = rt.Object.group()
        g.add_child(inner)
        g = inner
    g.add_child(rt.Object.sphere())
    with pytest.raises(rt.RtError) as e:
        w.scene()
    assert e.value.code == -7
    # 1e-5 determinant threshold of Matrix4::inverse (matrices.rs:118): 0.0215^3 < 1e-5 <= 0.0216^3
    w = rt.World()
    w.objects.append(rt.Object.sphere(transformation=rt.Matrix4.scale(0.0215, 0.0215, 0.0215)))
    with pytest.raises(rt.RtError) as e:
        w.scene()
    assert e.value.code == -2


def test_camera_new_and_view_transform(rt, orc):
    for h, v, fov in ((200, 125, math.pi / 2), (125, 200, math.pi / 2), (100, 100, 1.0), (3840, 2160, math.pi / 3)):
        a, b = rt.Camera(h, v, fov), orc.camera(h, v, fov)
        assert (a.pixel_size, a.half_width, a.half_height) == (b.pixel_size, b.half_width, b.half_height)
    assert abs(rt.Camera(200, 125, math.pi / 2).pixel_size - 0.01) < 1e-5  # camera.rs:94-97
    m = rt.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0))  # transformations.rs:53-66
    e = rt.Matrix4([[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
                    [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]])
    assert m == e


def test_canvas_ppm_writer(rt, orc, tmp_path):
    c = rt.Canvas(5, 3)
    assert c.to_ppm_string().split(b"\n")[:3] == [b"P3", b"5 3", b"255"]  # canvas.rs:165-172
    c.write_pixel(0, 0, (1.5, 0, 0))
    c.write_pixel(2, 1, (0, 0.5, 0))
    c.write_pixel(4, 2, (-0.5, 0, 1))
    c.write_pixel(5, 3, (1, 1, 1))  # out of bounds: silently ignored (canvas.rs:27-29)
    lines = c.to_ppm_string().split(b"\n")  # canvas.rs:175-188
    assert lines[3] == b"255 0 0 0 0 0 0 0 0 0 0 0 0 0 0"
    assert lines[4] == b"0 0 0 0 0 0 0 128 0 0 0 0 0 0 0"
    assert lines[5] == b"0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"
    with pytest.raises(IndexError):
        c.pixel_at(5, 0)
    c = rt.Canvas(10, 2)
    c.pixels[:, :] = (1.0, 0.8, 0.6)
    text = c.to_ppm_string()
    lines = text.split(b"\n")  # canvas.rs:191-213
    assert lines[3] == b"255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204"
    assert lines[4] == b"153 255 204 153 255 204 153 255 204 153 255 204 153"
    assert text.endswith(b"\n\n")
    rng = np.random.default_rng(7)
    img = rng.uniform(-0.2, 1.3, size=(37, 53, 3))
    img[0, 0] = (float("nan"), float("inf"), -float("inf"))
    img[1, 1] = (0.5 / 255, 1.5 / 255, 2.5 / 255)  # ties round half away from zero
    c = rt.Canvas(53, 37, img)
    assert c.to_ppm_string() == orc.to_ppm(img)
    assert np.array_equal(c.to_rgb8(), orc.quantize(img))
    p = tmp_path / "out.ppm"
    c.to_ppm(str(p))
    assert p.read_bytes() == orc.to_ppm(img)


def test_ppm_writer_reproduces_reference_files(rt):
    """Byte-identical P3 text for the reference's own images (from the committed 8-bit payloads)."""
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        man = json.load(f)["ppm"]
    for name, info in man.items():
        px = np.load(os.path.join(GOLDEN, name + ".npz"))["pixels"]
        c = rt.Canvas(px.shape[1], px.shape[0], px.astype(np.float64) / 255.0)
        text = c.to_ppm_string()
        assert hashlib.sha256(text).hexdigest() == info["sha256_p3_text"], name


def test_loader_matches_on_generated_scenes(rt, orc):
    """Generated scenes (every leaf kind, patterns, groups nested two deep, 10 000 leaves): every local and world
    transformation bit-identical between the C++ loader and the independent reader."""
    import tomli

    from raytracer_rs_amd import synthetic

    texts = [synthetic.random_scene(seed) for seed in range(1, 9)] + [synthetic.large_scene(64, 36)]
    for text in texts:
        sf = rt.SceneFile.from_str(text)
        wd, _ = sf.build_world().world_desc()
        _, oworld = orc.scene_from_dict(tomli.loads(text))
        assert wd.n_objects == oworld.object_count()
        for i in range(wd.n_objects):
            assert list(wd.objects[i].world_transformation) == [v for row in oworld.world_transform(i).rows() for v in row]


def test_command_line_renderer_usage(rt):
    """No GPU needed: argument errors are reported before anything touches a device."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_rs_amd", "rt_render")
    assert os.path.exists(exe), "rt_render has not been built (make -C raytracer_rs_amd/csrc)"
    done = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert done.returncode == 2 and "usage: rt_render" in done.stderr
    done = subprocess.run([exe, "a.toml", "--width", "10"], capture_output=True, text=True, timeout=60)
    assert done.returncode == 2
