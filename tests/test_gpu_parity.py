"""Parity of the HIP renderer with the CPU oracle, through the C ABI (all tests need an MI355X).

The bar (stated once, used everywhere below):
  * f64 colours within TOL = 1e-6 of the oracle before quantisation;
  * 8-bit images within +-1 level on >= 99.9 % of pixels;
  * everything that is an index, a count or a decision is exact: hit object indices, ray/operation counters,
    shadow flags, and intersection distances t (same IEEE operations in the same order, no contraction).
The only arithmetic that is not bit-identical to the oracle is `powf` in the specular term (device libm vs glibc)
and the summation order of the ray tree (path weights instead of nested sums), both a few ulp; GUARD = 1e-12 is a
regression tripwire far below TOL that any real divergence (a flipped decision, a wrong epsilon) breaks.
"""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path

pytestmark = pytest.mark.gpu

TOL = 1e-6
GUARD = 1e-12
S2 = math.sqrt(2.0)
ALL_SCENES = ["chapter_5", "chapter_6", "chapter_7_and_8", "chapter_9", "chapter_10", "chapter_11", "chapter_12",
              "chapter_13", "chapter_14"]
PPM_PAIRS = [("chapter_5", "chapter_5_circle"), ("chapter_6", "chapter_6_sphere"), ("chapter_10", "chapter_10_patterns"),
             ("chapter_11", "chapter_11_reflections"), ("chapter_13", "chapter_13_rocket")]


def assert_frames_match(orc, gpu_img, ref_img):
    d = np.abs(gpu_img - ref_img)
    assert d.max() <= TOL, f"max |gpu - oracle| = {d.max()}"
    assert d.max() <= GUARD, f"max |gpu - oracle| = {d.max()} (within the 1e-6 bar but beyond the expected few ulp)"
    qg, qr = orc.quantize(gpu_img).astype(int), orc.quantize(ref_img).astype(int)
    off = np.abs(qg - qr).max(axis=2)
    assert (off <= 1).mean() >= 0.999
    # a value that lies on a rounding boundary of the 8-bit scale may fall either way with the last ulp
    assert (off > 0).sum() <= max(1, int(1e-4 * off.size))


def counters_equal(stats, counters):
    assert stats.as_dict() == counters.as_dict()


def by_levels(scene):
    """The world is rendered level by level of its ray trees (rt_scene_info.kernel_build 3: worlds with groups, unless
    RT_RENDER_LEVELS=0 when the scene was created) instead of by one launch."""
    return scene.info().kernel_build == 3


def assert_kernels_agree(scene, plain, counted):
    """The counting kernel -- always the single launch, it visits what the reference visits -- and the plain path render
    the same frame: bit for bit when the plain path is the single launch too; within 1e-13 when the world is rendered
    level by level, where a ray's term is added to its parent's sum (reflection, then refraction) instead of to a
    running sum in the order the rays are walked."""
    if by_levels(scene):
        assert np.abs(plain - counted).max() <= 1e-13, "counting and plain kernels disagree"
    else:
        assert np.array_equal(plain, counted), "counting and plain kernels disagree"


def load_both(rt, orc, scene, width=None, height=None):
    sf = rt.SceneFile.from_file(scene_path(scene))
    cam = sf.build_camera(width, height)
    world = sf.build_world()
    ocam, oworld = orc.load_scene(scene_path(scene), width, height)
    return cam, world, ocam, oworld


@pytest.mark.parametrize("scene", ALL_SCENES)
def test_scene_matches_oracle(gpu, rt, orc, scene):
    cam, world, ocam, oworld = load_both(rt, orc, scene)
    img, st = world.scene().render(cam, stats=True)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=16, counters=oc)
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    plain, _ = world.scene().render(cam)
    assert_frames_match(orc, plain, ref)
    assert_kernels_agree(world.scene(), plain, img)


@pytest.mark.parametrize("scene,image", PPM_PAIRS)
def test_gpu_reproduces_reference_ppm(gpu, rt, scene, image):
    """Straight against the reference's own renderings: Camera::render -> Canvas::to_ppm."""
    import hashlib
    import json

    sf = rt.SceneFile.from_file(scene_path(scene))
    canvas = sf.build_camera().render(sf.build_world())
    golden = np.load(os.path.join(GOLDEN, image + ".npz"))["pixels"].astype(int)
    off = np.abs(canvas.to_rgb8().astype(int) - golden).max(axis=2)
    assert (off <= 1).mean() >= 0.999
    assert off.max() == 0, f"{int((off > 0).sum())} pixels differ from the reference PPM"
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        sha = json.load(f)["ppm"][image]["sha256_p3_text"]
    assert hashlib.sha256(canvas.to_ppm_string()).hexdigest() == sha


def test_command_line_renderer_writes_the_reference_ppm(gpu, tmp_path):
    """rt_render stands in for the reference's main(): scene file in, byte-identical PPM out."""
    import hashlib
    import json
    import subprocess

    exe = os.path.join(os.path.dirname(GOLDEN), "..", "raytracer_rs_amd", "rt_render")
    out = tmp_path / "chapter_6_sphere.ppm"
    done = subprocess.run([exe, scene_path("chapter_6"), "-o", str(out), "--stats"], capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr
    assert "rays" in done.stdout
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        sha = json.load(f)["ppm"]["chapter_6_sphere"]["sha256_p3_text"]
    assert hashlib.sha256(out.read_bytes()).hexdigest() == sha
    bad = subprocess.run([exe, str(tmp_path / "missing.toml")], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "missing.toml" in bad.stderr


def test_command_line_renderer_reproduces_mains_set_of_images(gpu, tmp_path):
    """`rt_render --all-chapters` stands in for the reference's main() (main.rs:172-184): nine scene files in, nine PPMs
    and nine JPEGs under the reference's output names out; the five whose PPM the reference ships are byte-identical
    to it, and so are their JPEGs (Canvas::to_jpeg, canvas.rs:109-114); the other four JPEGs are held to the
    reference's JPEGs of those scenes pixel by pixel after decoding (both are quality-100 encodings of frames that
    agree to the last bit wherever they can be compared, so they agree almost everywhere exactly)."""
    import hashlib
    import json
    import subprocess

    exe = os.path.join(os.path.dirname(GOLDEN), "..", "raytracer_rs_amd", "rt_render")
    done = subprocess.run([exe, "--all-chapters", os.path.join(GOLDEN, "scenes"), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        manifest = json.load(f)
    ppm, jpg_of_ppm = manifest["ppm"], manifest["jpg_of_ppm"]
    names = ["chapter_5_circle", "chapter_6_sphere", "chapter_7_and_8_world", "chapter_9_plane_scene", "chapter_10_patterns",
             "chapter_11_reflections", "chapter_12_cubes", "chapter_13_rocket", "chapter_14_groups"]
    for name in names:
        data = (tmp_path / (name + ".ppm")).read_bytes()
        assert data.startswith(b"P3\n") and data.endswith(b"\n\n")
        if name in ppm:
            assert hashlib.sha256(data).hexdigest() == ppm[name]["sha256_p3_text"], name
        jpeg = (tmp_path / (name + ".jpg")).read_bytes()
        if name in jpg_of_ppm:
            assert hashlib.sha256(jpeg).hexdigest() == jpg_of_ppm[name]["sha256"], name
        else:  # the reference's own JPEG of this scene is the only rendering of it the reference ships
            with open(os.path.join(GOLDEN, name + ".jpg"), "rb") as f:
                theirs = f.read()
            assert hashlib.sha256(theirs).hexdigest() == manifest["jpg"][name]["sha256"]
            assert jpeg == theirs, f"{name}.jpg differs from the reference's file"
    assert sum(name in ppm for name in names) == 5


def test_two_ranks_render_one_frame(gpu):
    """The several-ranks path of bench.py with HIP kernels on every rank: two processes (sharing the one GPU of a test
    box, process group on gloo, so the bands travel through pinned host memory) render their bands of the chapter_11
    frame, rank 0 collects them and compares the frame with the one it renders alone (`--check`, exit status 3 on a
    difference).  With RCCL the same code runs one rank per GPU."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(GOLDEN))
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_WATCHDOG="100")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--check",
           "--no-cpu-baseline", "--width", "960", "--height", "540"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=env, cwd=root)
    assert done.returncode == 0, done.stderr[-2000:]
    assert "collected frame equals the single-GPU frame" in done.stderr
    import json
    line = json.loads(done.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["rays_per_frame"] > 960 * 540


def test_device_quantiser_matches_host(gpu, rt):
    """canvas.rs:123-130 on a frame in device memory: ties away from zero, saturation, NaN -> 0."""
    import ctypes

    from raytracer_rs_amd import _lib

    rng = np.random.default_rng(3)
    vals = np.concatenate([rng.uniform(-0.2, 1.3, size=100003), np.arange(0, 256) / 255.0, (np.arange(0, 256) + 0.5) / 255.0,
                           np.nextafter((np.arange(0, 256) + 0.5) / 255.0, -1.0),
                           [0.0, -0.0, 1.0, 2.0, -1.0, np.nan, np.inf, -np.inf, 1e300, -1e300, 0.5 / 255.0, 254.5 / 255.0]])
    expect = np.zeros(vals.size, dtype=np.uint8)
    lib = _lib.lib()
    _lib.check(lib.rt_canvas_quantize(vals.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), vals.size,
                                      expect.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself is linked against
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_in), vals.nbytes) == 0 and hip.hipMalloc(ctypes.byref(d_out), vals.size + 16) == 0
    try:
        assert hip.hipMemcpy(d_in, vals.ctypes.data_as(ctypes.c_void_p), vals.nbytes, 1) == 0  # host to device
        for offset in (0, 1, 5):  # the output need not be 8-byte aligned
            assert hip.hipMemset(d_out, 0, vals.size + 16) == 0
            _lib.check(lib.rt_canvas_quantize_device(d_in, vals.size, ctypes.c_void_p(d_out.value + offset), None))
            assert hip.hipDeviceSynchronize() == 0
            got = np.zeros(vals.size + 16, dtype=np.uint8)
            assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), d_out, got.size, 2) == 0  # device to host
            assert np.array_equal(got[offset:offset + vals.size], expect)
            assert not got[:offset].any() and not got[offset + vals.size:].any()
    finally:
        hip.hipFree(d_in)
        hip.hipFree(d_out)


# --------------------------------------------------------------------------- the reference's unit tests, on the GPU
def close(a, b, eps=1e-5):
    return all(abs(x - y) < eps for x, y in zip(a, b))


def test_world_known_answers(gpu, rt):
    w = rt.World.default()
    r = rt.Ray((0, 0, -5), (0, 0, 1))
    assert w.hit(r) == (4.0, 0)  # world.rs:265-274: first of 4, 4.5, 5.5, 6
    assert w.color_at(rt.Ray((0, 0, -5), (0, 1, 0))) == (0.0, 0.0, 0.0)  # world.rs:308-313
    assert close(w.color_at(r), (0.38066, 0.47583, 0.2855))  # world.rs:316-321
    inner = rt.World.default()
    inner.light_source = rt.PointLight((0, 0.25, 0), (1, 1, 1))
    # world.rs:291-305 through color_at: from inside the inner sphere
    assert close(inner.color_at(rt.Ray((0, 0, 0), (0, 0, 1))), (0.90498, 0.90498, 0.90498))
    amb = rt.World.default()
    amb.objects[0].material.ambient = 1.0
    amb.objects[1].material.ambient = 1.0
    assert close(amb.color_at(rt.Ray((0, 0, 0.75), (0, 0, -1))), amb.objects[1].material.color)  # world.rs:324-346
    assert not w.is_shadowed((0, 10, 0))  # world.rs:349-374
    assert w.is_shadowed((10, -10, 10))
    assert not w.is_shadowed((-20, 20, -20))
    assert not w.is_shadowed((-2, 2, -2))
    sh = rt.World()
    sh.light_source = rt.PointLight((0, 0, -10), (1, 1, 1))
    sh.objects = [rt.Object.sphere(), rt.Object.sphere(transformation=rt.Matrix4.translate(0, 0, 10))]
    assert close(sh.color_at(rt.Ray((0, 0, 5), (0, 0, 1))), (0.1, 0.1, 0.1))  # world.rs:377-393
    assert rt.World().color_at(r) == (0.0, 0.0, 0.0)  # empty world (world.rs:251-255)
    dark = rt.World.default()
    dark.light_source = None
    assert dark.color_at(r) == (0.0, 0.0, 0.0)  # no light: black (world.rs:86-89)


def test_reflection_refraction_known_answers(gpu, rt, orc):
    def floor_world(**mat):
        w = rt.World.default()
        w.objects.append(rt.Object.plane(transformation=rt.Matrix4.translate(0, -1, 0), material=rt.Material(**mat)))
        return w

    r = rt.Ray((0, 0, -3), (0, -S2 / 2, S2 / 2))
    assert close(floor_world(reflectivity=0.5).color_at(r), (0.87676, 0.92434, 0.82917))  # world.rs:440-470
    # depth 0: the reflective floor is lit but casts no further ray (world.rs:502-524)
    w = floor_world(reflectivity=0.5)
    c0, st = w.scene().color_at(np.array([[0, 0, -3, 0, -S2 / 2, S2 / 2]]), remaining=0, stats=True)
    assert st.reflect == 0 and st.rays == 2
    mirrors = rt.World()
    mirrors.light_source = rt.PointLight((0, 0, 0), (1, 1, 1))
    mirrors.objects = [rt.Object.plane(transformation=rt.Matrix4.translate(0, -1, 0), material=rt.Material(reflectivity=1.0)),
                       rt.Object.plane(transformation=rt.Matrix4.translate(0, 1, 0), material=rt.Material(reflectivity=1.0))]
    _, st = mirrors.scene().color_at(np.array([[0, 0, 0, 0, 1, 0]]), stats=True)  # world.rs:474-499: terminates
    assert st.reflect == 5
    ball = rt.Object.sphere(transformation=rt.Matrix4.translate(0, -3.5, -0.5), material=rt.Material(color=(1, 0, 0), ambient=0.5))
    w = floor_world(transparency=0.5, refractive_index=1.5)
    w.objects.append(ball)
    assert close(w.color_at(r), (0.93642, 0.68642, 0.68642))  # world.rs:651-695
    w = floor_world(reflectivity=0.5, transparency=0.5, refractive_index=1.5)
    w.objects.append(ball)
    assert close(w.color_at(r), (0.93391, 0.69643, 0.69243))  # world.rs:698-744


def test_camera_known_answers(gpu, rt):
    cam = rt.Camera(201, 101, math.pi / 2)
    r = cam.ray_for_pixel(100, 50)  # camera.rs:106-111
    assert r.origin == (0, 0, 0) and close(r.direction, (0, 0, -1))
    r = cam.ray_for_pixel(0, 0)  # camera.rs:114-122
    assert close(r.direction, (0.6651864261194508, 0.3325932130597254, -0.6685123582500481), 1e-15)
    cam.transform = rt.Matrix4.rotate_y(math.pi / 4) * rt.Matrix4.translate(0, -2, 5)
    r = cam.ray_for_pixel(100, 50)  # camera.rs:125-134
    assert close(r.origin, (0, 2, -5)) and close(r.direction, (S2 / 2, 0, -S2 / 2))
    cam = rt.Camera(11, 11, math.pi / 2)
    cam.transform = rt.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0))
    image = cam.render(rt.World.default())  # camera.rs:137-147
    assert close(image.pixel_at(5, 5), (0.38066, 0.47583, 0.2855))
    bad = rt.Camera(4, 4, 1.0)
    bad.transform = rt.Matrix4.scale(1, 0, 1)
    with pytest.raises(rt.RtError) as e:  # camera.rs:46-49 panics; here an error code
        bad.render(rt.World.default())
    assert e.value.code == -2 and "camera transform should be invertible" in e.value.message


def test_camera_rays_match_oracle_exactly(gpu, rt, orc):
    sf = rt.SceneFile.from_file(scene_path("chapter_14"))
    cam = sf.build_camera()
    ocam, _ = orc.load_scene(scene_path("chapter_14"))
    rng = np.random.default_rng(3)
    px = rng.integers(0, cam.hsize, 200).tolist() + [0, cam.hsize - 1]
    py = rng.integers(0, cam.vsize, 200).tolist() + [0, cam.vsize - 1]
    for ray, x, y in zip(cam.rays_for_pixels(px, py), px, py):
        o = orc.ray_for_pixel(ocam, x, y)
        assert ray.origin == o.origin.tuple() and ray.direction == o.direction.tuple()


def test_group_known_answers(gpu, rt):
    w = rt.World()
    g = rt.Object.group()
    s1, s2, s3 = rt.Object.sphere(), rt.Object.sphere(transformation=rt.Matrix4.translate(0, 0, -3)), \
        rt.Object.sphere(transformation=rt.Matrix4.translate(5, 0, 0))
    for s in (s1, s2, s3):
        g.add_child(s)
    w.objects = [g]
    t, obj = w.hit(rt.Ray((0, 0, -5), (0, 0, 1)))  # groups.rs:244-273: 1, 3 (s2), 4, 6 (s1)
    assert t == 1.0 and obj == w.object_index(s2)
    assert w.hit(rt.Ray((10, 10, 10), (1, 0, 0))) is None  # groups.rs:452-469
    w = rt.World()
    g = rt.Object.group(transformation=rt.Matrix4.scale(2, 2, 2))
    g.add_child(rt.Object.sphere(transformation=rt.Matrix4.translate(5, 0, 0)))
    w.objects = [g]
    assert w.hit(rt.Ray((10, 0, -10), (0, 0, 1))) is not None  # groups.rs:276-293
    w = rt.World()
    g1 = rt.Object.group(transformation=rt.Matrix4.rotate_y(math.pi / 2))
    g2 = rt.Object.group(transformation=rt.Matrix4.scale(1, 2, 3))
    s = rt.Object.sphere(transformation=rt.Matrix4.translate(5, 0, 0))
    g2.add_child(s)
    g1.add_child(g2)
    w.objects = [g1]
    n = w.scene().normal_at([w.object_index(s)], np.array([[1.7321, 1.1547, -5.5774]]))[0]  # groups.rs:351-376
    assert close(n, (0.2857, 0.4286, -0.8571), 1e-4)
    with pytest.raises(rt.RtError):  # groups.rs:125-127: groups have no normal
        w.scene().normal_at([w.object_index(g1)], np.array([[0.0, 0.0, 0.0]]))


# --------------------------------------------------------------------------- shapes: exact t and hit object
def _single_object_worlds(rt, orc):
    I = None
    cases = []
    spec = [
        (rt.SPHERE, orc.SPHERE, {}), (rt.PLANE, orc.PLANE, {}), (rt.CUBE, orc.CUBE, {}),
        (rt.CYLINDER, orc.CYLINDER, {}), (rt.CYLINDER, orc.CYLINDER, dict(minimum=1.0, maximum=2.0)),
        (rt.CYLINDER, orc.CYLINDER, dict(minimum=-0.5, maximum=0.75, closed=True)),
        (rt.CONE, orc.CONE, {}), (rt.CONE, orc.CONE, dict(minimum=-0.5, maximum=0.5, closed=True)),
        (rt.CONE, orc.CONE, dict(minimum=-1.0, maximum=0.0, closed=True)),
    ]
    transforms = [None, (("translate", (0.3, -0.2, 0.5)), ("rotate_x", 0.7), ("scale", (0.8, 1.3, 0.6)))]
    for kind, okind, kw in spec:
        for tr in transforms:
            m, om = rt.Matrix4.identity(), orc.identity()
            if tr:
                for name, arg in tr:
                    a = arg if isinstance(arg, tuple) else (arg,)
                    m = m * getattr(rt.Matrix4, name)(*a)
                    om = orc.mul(om, getattr(orc, name)(*a))
            w = rt.World()
            w.light_source = rt.PointLight((-10, 10, -10), (1, 1, 1))
            w.objects = [rt.Object(kind, m, None, kw.get("minimum", -math.inf), kw.get("maximum", math.inf), kw.get("closed", False))]
            ow = orc.World()
            ow.set_light((-10, 10, -10), (1, 1, 1))
            ow.add(okind, om, None, -1, kw.get("minimum", -math.inf), kw.get("maximum", math.inf), kw.get("closed", False))
            cases.append((w, ow))
    return cases


def test_intersections_are_bit_exact(gpu, rt, orc):
    rng = np.random.default_rng(11)
    n = 4000
    origins = rng.uniform(-3, 3, size=(n, 3))
    targets = rng.uniform(-1.2, 1.2, size=(n, 3))
    dirs = targets - origins
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    # axis-parallel, tangent and cap-grazing rays: the branches the thresholds guard
    special = np.array([[0, 0, -5, 0, 0, 1], [0, 1, -5, 0, 0, 1], [1, 0, -5, 0, 0, 1], [0, 0, -1, 0, S2 / 2, S2 / 2],
                        [0, 0, -0.25, 0, 1, 0], [0, 3, 0, 0, -1, 0], [0.5, 5, 0, 0, -1, 0], [2, 2, 0, -1, 0, 0],
                        [0, 1, 0, 1, 0, 0], [0, 1e-6, -2, 0, 0, 1], [0, 0.5, 0, 1, 1e-6, 0], [0, 0, 0, 0, 0, 1]], dtype=float)
    rays = np.concatenate([np.concatenate([origins, dirs], axis=1), special])
    for w, ow in _single_object_worlds(rt, orc):
        t, obj = w.scene().intersect(rays)
        for i, ray in enumerate(rays):
            hit = ow.hit(orc.ray(ray[:3], ray[3:]))
            if hit is None:
                assert obj[i] == -1 and math.isnan(t[i]), (i, ray)
            else:
                assert obj[i] == hit[1] and t[i] == hit[0], (i, ray, t[i], hit)


def test_normals_are_bit_exact(gpu, rt, orc):
    rng = np.random.default_rng(5)
    for w, ow in _single_object_worlds(rt, orc):
        pts = rng.uniform(-1.5, 1.5, size=(300, 3))
        got = w.scene().normal_at([0] * len(pts), pts)
        for p, g in zip(pts, got):
            e = ow.normal_at(0, p)
            assert all((a == b) or (math.isnan(a) and math.isnan(b)) for a, b in zip(g, e)), (p, g, e)


# --------------------------------------------------------------------------- generated scenes
# n_top > 12 gives the top-level list a tree of range-skip nodes on top of the per-member ones every scene with groups has
@pytest.mark.parametrize("seed,n_top", [(s, 6) for s in range(1, 13)] + [(101, 25), (102, 40), (103, 60), (104, 90)])
def test_random_scene_matches_oracle(gpu, rt, orc, seed, n_top):
    import tomli

    from raytracer_rs_amd import synthetic

    text = synthetic.random_scene(seed, 96, 64, n_top)
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    img, st = world.scene().render(cam, stats=True)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=8, counters=oc)
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    plain, _ = world.scene().render(cam)
    assert_frames_match(orc, plain, ref)
    assert_kernels_agree(world.scene(), plain, img)


def test_large_generated_scene_matches_oracle(gpu, rt, orc):
    """The 10 000-leaf, three-level hierarchy at a reduced frame size."""
    import tomli

    from raytracer_rs_amd import synthetic

    text = synthetic.large_scene(320, 180)
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    info = world.scene().info()
    assert (info.n_leaves, info.n_groups, info.max_nesting) == (10001, 500, 2)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    img, st = world.scene().render(cam, stats=True)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=16, counters=oc)
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    # the plain path steps over runs of nodes no ray of a wave can touch, and renders a hierarchy of this size level by
    # level; the counting kernel visits everything, in one launch
    assert info.kernel_build == 3
    plain, _ = world.scene().render(cam)
    assert_frames_match(orc, plain, ref)
    assert_kernels_agree(world.scene(), plain, img)


def _lone_rays_agree(orc, scene, oworld, rays):
    """Every ray alone in its wave (64 copies of it): nothing another lane needs keeps a skipped run alive."""
    t, obj = scene.intersect(np.repeat(rays, 64, axis=0))
    t, obj = t.reshape(-1, 64), obj.reshape(-1, 64)
    assert (obj == obj[:, :1]).all()
    for k, ray in enumerate(rays):
        hit = oworld.hit(orc.ray(ray[:3], ray[3:]))
        if hit is None:
            assert obj[k, 0] == -1, (k, ray, obj[k, 0], t[k, 0])
        else:
            assert (obj[k, 0], t[k, 0]) == (hit[1], hit[0]), (k, ray, (obj[k, 0], t[k, 0]), hit)


def _probe_rays(orc, ocam, seed, n_camera, n_random, box):
    rng = np.random.default_rng(seed)
    rays = []
    for _ in range(n_camera):
        r = orc.ray_for_pixel(ocam, int(rng.integers(0, ocam.hsize)), int(rng.integers(0, ocam.vsize)))
        rays.append(list(r.origin.tuple()) + list(r.direction.tuple()))
    lo, hi = np.array(box[0], dtype=float), np.array(box[1], dtype=float)
    o = rng.uniform(lo, hi, size=(n_random, 3))
    d = rng.normal(size=(n_random, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([np.array(rays), np.concatenate([o, d], axis=1)])


def test_range_skipping_never_loses_a_hit(gpu, rt, orc):
    """Closest hits of lone rays (camera rays and rays from random points in random directions) in the scenes whose
    node tables carry range-skip nodes: the 10 001-leaf hierarchy and chapter_14's hexagons."""
    import tomli

    from raytracer_rs_amd import synthetic

    text = synthetic.large_scene(320, 180)
    sf = rt.SceneFile.from_str(text)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    rays = _probe_rays(orc, ocam, 21, 300, 500, ((-21, 0.05, -21), (21, 3, 21)))
    _lone_rays_agree(orc, sf.build_world().scene(), oworld, rays)

    cam, world, ocam, oworld = load_both(rt, orc, "chapter_14", 200, 100)
    rays = _probe_rays(orc, ocam, 22, 300, 500, ((-3, -1, -3), (3, 3, 3)))
    _lone_rays_agree(orc, world.scene(), oworld, rays)


def test_range_skipping_on_rays_the_reference_treats_specially(gpu, rt, orc):
    """Rays picked to sit on the thresholds a range-skip box has to be conservative about, inside a group with a
    non-uniform scale and two rotations (so that box space, list space and world space all differ) and against a tiny
    object far away (where an error of the reciprocal is multiplied by the distance):
      * directions with |d_i| < EPSILON in a cube's space, which the reference calls parallel to the slab and then
        accepts points up to EPSILON*t outside the cube (cubes.rs:47-53), from origins whose distance is just inside
        and well outside the reach for which a box may still cull;
      * directions with dx^2 + dz^2 below EPSILON in a cylinder's space (cylinders.rs:62-66: caps only);
      * origins on or within 1e-7 of a surface, heading out: the hit at t = 0 counts, the one at t = -1e-7 does not;
      * rays grazing the silhouette of a sphere of radius 0.022 (the smallest scale the reference can invert) from 4500
        units away, with directions ten units long, so that the sphere is within the reach where boxes may cull.
    Each ray alone in its wave, exact agreement with the oracle on object and distance."""
    import tomli

    head = ["[camera]", "width = 64", "height = 48", "fov = 60.0", "from = [0.0, 2.0, -30.0]", "to = [0.0, 0.0, 0.0]",
            "up = [0.0, 1.0, 0.0]", "[light]", "position = [-10.0, 10.0, -10.0]", "intensity = [1.0, 1.0, 1.0]"]
    lines = list(head)
    lines += ["[[objects]]", 'type = "group"', "[objects.transform]", "translate = [1.0, -0.5, 2.0]", "scale = [3.0, 0.5, 1.5]",
              "rotate_y = 30.0", "rotate_z = 20.0"]
    kinds = []
    for k in range(18):
        kind = ("cube", "cylinder", "sphere")[k % 3]
        kinds.append(kind)
        lines += ["[[objects.children]]", f'type = "{kind}"']
        if kind == "cylinder":
            lines += ["minimum = -1.0", "maximum = 1.0", "closed = true"]
        lines += ["[objects.children.transform]", f"translate = [{2.5 * k - 20.0}, {0.3 * (k % 4)}, {1.5 * (k % 5) - 3.0}]"]
        if k % 2:
            lines += [f"rotate_x = {15.0 * k}"]
    # a dozen more top-level objects, so that the world's own list gets range-skip nodes too; the last one tiny and far
    for k in range(12):
        lines += ["[[objects]]", 'type = "cube"', "[objects.transform]", f"translate = [{-40.0 + 3.0 * k}, 8.0, 30.0]", "scale = [0.5, 0.5, 0.5]"]
    lines += ["[[objects]]", 'type = "sphere"', "[objects.transform]", "translate = [4500.0, 0.0, 0.0]", "scale = [0.022, 0.022, 0.022]"]
    text = "\n".join(lines) + "\n"
    sf = rt.SceneFile.from_str(text)
    _, oworld = orc.scene_from_dict(tomli.loads(text))
    scene = sf.build_world().scene()
    assert scene.info().n_groups == 1

    G = orc.mul(orc.mul(orc.mul(orc.translate(1.0, -0.5, 2.0), orc.rotate_z(math.radians(20.0))), orc.rotate_y(math.radians(30.0))),
                orc.scale(3.0, 0.5, 1.5)).numpy()

    def to_world(o, d):  # a ray given in the group's space
        o4, d4 = G @ np.array([*o, 1.0]), G @ np.array([*d, 0.0])
        return [*o4[:3], *d4[:3]]

    rays = []
    eps_dirs = [(1.0, 9e-6, 0.0), (1.0, 0.0, 9e-6), (1.0, 1.1e-5, -9e-6), (-1.0, 9.9e-6, 9.9e-6), (1.0, -5e-6, 2e-5)]
    for k in range(0, 18, 3):  # the cubes: child k sits at (2.5k - 20, 0.3(k%4), 1.5(k%5) - 3) in the group's space
        cx, cy, cz = 2.5 * k - 20.0, 0.3 * (k % 4), 1.5 * (k % 5) - 3.0
        for dist in (3.0, 120.0, 497.0, 503.0, 2500.0):
            for d in eps_dirs:
                for edge in (0.999995, 1.000004, -0.999995, 0.5):  # grazing a face just inside / just outside, and through the middle
                    o = (cx - math.copysign(dist, d[0]), cy + edge, cz + 0.25 * edge)
                    rays.append(to_world(o, d))
    for k in range(1, 18, 3):  # the cylinders (odd k are rotated about x): rays almost along their axes, in their own space
        cx, cy, cz = 2.5 * k - 20.0, 0.3 * (k % 4), 1.5 * (k % 5) - 3.0
        R = orc.rotate_x(math.radians(15.0 * k)).numpy() if k % 2 else np.eye(4)
        for tilt in (1e-3, 2e-3, 3.2e-3, 1e-4):
            for off in (0.0, 0.5, 0.99, 1.01):
                d = R @ np.array([tilt, 1.0, tilt, 0.0])
                o = R @ np.array([off, -30.0, 0.1, 1.0])
                rays.append(to_world((cx + o[0], cy + o[1], cz + o[2]), d[:3]))
    for k in range(2, 18, 3):  # the spheres: origins on the surface and a hair inside / outside, heading out
        cx, cy, cz = 2.5 * k - 20.0, 0.3 * (k % 4), 1.5 * (k % 5) - 3.0
        for r in (1.0, 1.0 - 1e-7, 1.0 + 1e-7, 1.0 - 1e-9):
            for n in ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.6, 0.0, 0.8)):
                rays.append(to_world((cx + r * n[0], cy + r * n[1], cz + r * n[2]), n))
    rng = np.random.default_rng(5)
    for _ in range(200):  # the far, tiny sphere: aim at points on and around its silhouette as seen from near the origin
        o = rng.uniform(-2.0, 2.0, size=3)
        a = rng.uniform(0.0, 2.0 * math.pi)
        r = 0.022 * (1.0 + rng.choice([-1e-3, -1e-5, 0.0, 1e-5, 1e-3, -0.5]))
        target = np.array([4500.0, r * math.cos(a), r * math.sin(a)])
        d = target - o
        rays.append([*o, *(d / np.linalg.norm(d) * rng.choice([1.0, 10.0, 10.0]))])
    rays = np.array(rays, dtype=np.float64)
    hits = sum(oworld.hit(orc.ray(r[:3], r[3:])) is not None for r in rays)
    assert hits >= len(rays) // 4, hits
    _lone_rays_agree(orc, scene, oworld, rays)


def test_sphere_balls_never_lose_a_hit(gpu, rt, orc):
    """In a world without groups a wave skips a sphere when none of its rays comes near the ball the host put round
    it.  The ball has to contain every point at which the reference can report a hit, rounding of its transformed ray
    and of its discriminant included.  Rays that pass a sphere's silhouette at 1 +- 1e-15 .. 1e-6 radii (built in the
    sphere's own space, so that the distance is what it says), from origins 1.5 to a million radii away, with
    directions of several lengths; rays that start on or a hair outside the surface and head away (the reference's
    larger root then is a difference of nearly equal numbers and may come out as 0 = a hit).  Spheres: at the origin;
    a thousand and a hundred thousand units from it; stretched 4:1 and 300:1 (the latter gets no ball); of radius
    0.022.  Each ray 64 times, so that its wave's decision is its own; the colour (materials are ambient only, one
    colour per sphere) must be the oracle's: a lost hit is black instead of a colour."""
    import tomli

    spheres = [((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), ((1000.5, -3.0, 7.0), (0.5, 0.5, 0.5)), ((5.0, 1.0, 0.0), (2.0, 0.5, 1.0)),
               ((40.0, 0.0, 0.0), (0.022, 0.022, 0.022)), ((0.0, 30.0, 0.0), (300.0, 1.0, 1.0)), ((1.0e5, 0.0, 0.0), (1.0, 1.0, 1.0))]
    lines = ["[camera]", "width = 64", "height = 48", "fov = 60.0", "from = [0.0, 2.0, -30.0]", "to = [0.0, 0.0, 0.0]",
             "up = [0.0, 1.0, 0.0]", "[light]", "position = [-10.0, 500.0, -10.0]", "intensity = [1.0, 1.0, 1.0]"]
    for k, (t, s) in enumerate(spheres):
        lines += ["[[objects]]", 'type = "sphere"', "[objects.transform]", f"translate = [{t[0]}, {t[1]}, {t[2]}]",
                  f"scale = [{s[0]}, {s[1]}, {s[2]}]", "[objects.material]", f"color = [{0.125 * (k + 1)}, 0.5, {1.0 - 0.125 * k}]",
                  "ambient = 1.0", "diffuse = 0.0", "specular = 0.0"]
    text = "\n".join(lines) + "\n"
    sf = rt.SceneFile.from_str(text)
    _, oworld = orc.scene_from_dict(tomli.loads(text))
    scene = sf.build_world().scene()
    assert scene.info().n_groups == 0

    rng = np.random.default_rng(11)
    rays = []
    for t, s in spheres:
        t, s = np.array(t), np.array(s)
        for dist in (1.5, 30.0, 1.0e3, 1.0e6):
            for off in (-1e-6, -1e-9, -1e-12, -1e-15, 0.0, 1e-15, 1e-12, 1e-9, 1e-6):
                for _ in range(3):
                    u = rng.normal(size=3)
                    u /= np.linalg.norm(u)
                    v = np.cross(u, rng.normal(size=3))
                    v /= np.linalg.norm(v)
                    o = (1.0 + off) * v - dist * u  # passes the unit sphere at distance 1 + off, in the sphere's space
                    d = u * rng.choice([1.0, 1.0, 10.0, 0.1])
                    rays.append([*(o * s + t), *(d * s)])
        for off in (0.0, 1e-16, 1e-12, 1e-9, 1e-5, -1e-12):
            for _ in range(6):
                n = rng.normal(size=3)
                n /= np.linalg.norm(n)
                tangent = np.cross(n, rng.normal(size=3))
                tangent /= np.linalg.norm(tangent)
                for d in (n, n + 1e-3 * tangent, tangent + 1e-9 * n, tangent):
                    rays.append([*((1.0 + off) * n * s + t), *(d * s)])
    hits = _lone_colours_agree(orc, scene, oworld, np.array(rays, dtype=np.float64))
    assert 0.2 * len(rays) < hits < 0.8 * len(rays), hits


def _lone_colours_agree(orc, scene, oworld, rays):
    """World::color_at of every ray alone in its wave (64 copies of it) against the oracle's; returns how many of the
    rays see anything.  Where the reference's own arithmetic gives NaN, NaN must be the answer."""
    want = np.array([oworld.color_at(orc.ray(r[:3], r[3:])) for r in rays])
    got = scene.color_at(np.repeat(rays, 64, axis=0)).reshape(len(rays), 64, 3)
    same = ((got == got[:, :1]) | (np.isnan(got) & np.isnan(got[:, :1]))).all(axis=(1, 2))
    assert same.all(), (np.nonzero(~same)[0][:8], rays[~same][:4])
    # relative where the colour is large (an unnormalised direction makes the specular power astronomical)
    with np.errstate(invalid="ignore", over="ignore"):
        diff = np.where((np.isnan(want) & np.isnan(got[:, 0])) | (want == got[:, 0]), 0.0, np.abs(got[:, 0] - want) / np.maximum(1.0, np.abs(want)))
    wrong = np.nonzero(~(diff.max(axis=1) <= 1e-12))[0]
    assert len(wrong) == 0, (len(wrong), rays[wrong[:5]], got[wrong[:5], 0], want[wrong[:5]])
    return int((want != 0.0).any(axis=1).sum())


def test_plane_skipping_keeps_the_hit_at_zero(gpu, rt, orc):
    """A wave skips a plane when every ray heads away from it (t = -o.y / d.y negative).  Negative, not zero: a
    quotient that underflows is -0.0, and -0.0 >= 0.0 makes it a hit in the reference (intersections.rs:37-39).
    Origins a denormal above the plane with steep directions, origins exactly on it (+0.0 and -0.0), ordinary rays
    towards and away from it, rays nearly parallel to it; a scaled and translated plane, so that its own space
    differs from the world's."""
    import tomli

    lines = ["[camera]", "width = 64", "height = 48", "fov = 60.0", "from = [0.0, 2.0, -30.0]", "to = [0.0, 0.0, 0.0]",
             "up = [0.0, 1.0, 0.0]", "[light]", "position = [-10.0, 500.0, -10.0]", "intensity = [1.0, 1.0, 1.0]",
             "[[objects]]", 'type = "plane"', "[objects.material]", "color = [0.25, 0.5, 0.75]", "ambient = 1.0", "diffuse = 0.0", "specular = 0.0",
             "[[objects]]", 'type = "plane"', "[objects.transform]", "translate = [0.0, -4.0, 0.0]", "scale = [2.0, 0.5, 3.0]",
             "[objects.material]", "color = [0.75, 0.25, 0.5]", "ambient = 1.0", "diffuse = 0.0", "specular = 0.0"]
    text = "\n".join(lines) + "\n"
    sf = rt.SceneFile.from_str(text)
    _, oworld = orc.scene_from_dict(tomli.loads(text))
    scene = sf.build_world().scene()
    rays = []
    for oy in (5e-324, 1e-320, 1e-310, 1e-300, 1e-200, 1e-20, 0.0, -0.0, 1.0, -5e-324, -1e-300, -1.0, 3.0, -4.0, -4.0 + 1e-15, -5.0):
        for dy in (1.0, -1.0, 1e5, -1e5, 1e150, -1e150, 1e300, -1e300, 2e-5, -2e-5, 0.9e-5, 1e-300, 0.0):
            rays.append([0.5, oy, -0.25, 0.1, dy, 0.2])
    hits = _lone_colours_agree(orc, scene, oworld, np.array(rays, dtype=np.float64))
    assert 0.3 * len(rays) < hits < 0.95 * len(rays), hits


@pytest.mark.parametrize("seed", range(1, 9))
def test_random_worlds_without_groups_lone_rays(gpu, rt, orc, seed):
    """Worlds of spheres and planes only (the kind whose traversal skips objects on a ball or a sign test), with
    random translations, scales of either sign from 0.03 to 30 (some spheres stretched past the hundredfold limit
    beyond which they get no ball), reflective and transparent materials; rays from random points in random directions
    and rays aimed at the spheres' silhouettes, each alone in its wave: World::color_at must be the oracle's."""
    import tomli

    rng = np.random.default_rng(1000 + seed)
    lines = ["[camera]", "width = 32", "height = 24", "fov = 60.0", "from = [0.0, 2.0, -30.0]", "to = [0.0, 0.0, 0.0]",
             "up = [0.0, 1.0, 0.0]", "[light]", f"position = [{rng.uniform(-20, 20):.3f}, {rng.uniform(5, 30):.3f}, {rng.uniform(-20, 20):.3f}]",
             "intensity = [1.0, 1.0, 1.0]"]
    balls = []
    for k in range(int(rng.integers(3, 11))):
        plane = k > 0 and rng.random() < 0.2
        t = rng.uniform(-8, 8, size=3)
        sc = np.exp(rng.uniform(math.log(0.03), math.log(30.0), size=3)) * rng.choice([-1.0, 1.0], size=3)
        if rng.random() < 0.6:
            sc[:] = sc[0]
        lines += ["[[objects]]", f'type = "{"plane" if plane else "sphere"}"', "[objects.transform]",
                  f"translate = [{t[0]!r}, {t[1]!r}, {t[2]!r}]", f"scale = [{sc[0]!r}, {sc[1]!r}, {sc[2]!r}]",
                  "[objects.material]", f"color = [{rng.random():.3f}, {rng.random():.3f}, {rng.random():.3f}]",
                  f"reflectivity = {rng.choice([0.0, 0.0, 0.6]):.1f}", f"transparency = {rng.choice([0.0, 0.0, 0.8]):.1f}",
                  "refractive_index = 1.5"]
        if not plane:
            balls.append((t, sc))
    text = "\n".join(lines).replace("np.float64(", "").replace(")", "") + "\n"
    sf = rt.SceneFile.from_str(text)
    _, oworld = orc.scene_from_dict(tomli.loads(text))
    scene = sf.build_world().scene()
    assert scene.info().n_groups == 0
    rays = []
    for _ in range(150):
        d = rng.normal(size=3)
        rays.append([*rng.uniform(-12, 12, size=3), *(d / np.linalg.norm(d) * rng.choice([1.0, 0.01, 50.0]))])
    for t, sc in balls:
        for _ in range(30):  # towards a point of the silhouette as seen along u, give or take
            u = rng.normal(size=3)
            u /= np.linalg.norm(u)
            v = np.cross(u, rng.normal(size=3))
            v /= np.linalg.norm(v)
            o = (1.0 + rng.choice([-1e-3, -1e-9, 0.0, 1e-9, 1e-3])) * v - rng.choice([2.0, 40.0, 3000.0]) * u
            rays.append([*(o * sc + t), *(u * sc)])
    _lone_colours_agree(orc, scene, oworld, np.array(rays, dtype=np.float64))


def test_cone_linear_shortcut_survives_range_skipping(gpu, rt, orc):
    """A ray with dx^2 - dy^2 + dz^2 below EPSILON in a cone's space: the reference takes t = -c / (2 b)
    (cones.rs:66-78), a point well off the cone and outside its bounding box, and accepts it because its y lies
    within the cone's limits.  Such a ray must get past every range-skip node above that cone."""
    import tomli

    from raytracer_rs_amd import synthetic

    text = synthetic.large_scene(320, 180)
    sf = rt.SceneFile.from_str(text)
    _, oworld = orc.scene_from_dict(tomli.loads(text))
    # met as a fifth-generation ray of pixel (1301, 586) of the 1920x1080 frame
    ray = np.array([[3.66955354, 0.42627462, -4.58568923, 0.70703312, -0.70710699, -0.01019148]])
    hit = oworld.hit(orc.ray(ray[0, :3], ray[0, 3:]))
    assert hit is not None
    _lone_rays_agree(orc, sf.build_world().scene(), oworld, ray)
    # and a family of them: rays at 45 degrees to the axes of a row of unbounded-radius cones with y limits -1..1,
    # alternating with cubes, in one group; origins far outside the row on both sides
    w = rt.World()
    ow = orc.World()
    w.light_source = rt.PointLight((-10, 10, -10), (1, 1, 1))
    ow.set_light((-10, 10, -10), (1, 1, 1))
    g = rt.Object.group()
    og = ow.add(orc.GROUP, orc.identity())
    for k in range(20):
        kind, okind = (rt.CONE, orc.CONE) if k % 2 else (rt.CUBE, orc.CUBE)
        g.add_child(rt.Object(kind, transformation=rt.Matrix4.translate(3.0 * k, 0, 0), minimum=-1.0, maximum=1.0))
        ow.add(okind, orc.translate(3.0 * k, 0, 0), None, og, -1.0, 1.0, False)
    w.objects = [g]
    ow.finalize()
    rays = []
    for k in range(1, 20, 2):
        for off in (-40.0, -7.5, 2.5, 9.0, 33.0):
            sx = 1.0 if off < 0 else -1.0
            # reaches y = 0.25 above the cone at x = 3k after |off| / (S2/2) along the ray
            rays.append([3.0 * k + off, 0.25 + abs(off), 0.3, sx * S2 / 2, -S2 / 2, 0.0])
    rays = np.array(rays)
    assert sum(ow.hit(orc.ray(r[:3], r[3:])) is not None for r in rays) >= 10
    _lone_rays_agree(orc, w.scene(), ow, rays)
    # Small cones (scale 0.03: the room between the reference's threshold and a node's bound shrinks with size^2) whose
    # axes differ by a few 1e-10 rad, in a group: their direction forms agree to within 1e-9 but not to within rounding,
    # so they must NOT be given one form between them -- a ray in one cone's shortcut zone would be culled by another's.
    w = rt.World()
    ow = orc.World()
    w.light_source = rt.PointLight((-10, 10, -10), (1, 1, 1))
    ow.set_light((-10, 10, -10), (1, 1, 1))
    g = rt.Object.group()
    og = ow.add(orc.GROUP, orc.identity())
    s = 0.03
    for k in range(24):
        tilt = 4e-10 * k
        if k % 3:
            t = rt.Matrix4.translate(0.5 * k, 0, 0) * rt.Matrix4.rotate_z(tilt) * rt.Matrix4.scale(s, s, s)
            ot = orc.mul(orc.mul(orc.translate(0.5 * k, 0, 0), orc.rotate_z(tilt)), orc.scale(s, s, s))
            g.add_child(rt.Object(rt.CONE, transformation=t, minimum=-1.0, maximum=1.0))
            ow.add(orc.CONE, ot, None, og, -1.0, 1.0, False)
        else:
            t = rt.Matrix4.translate(0.5 * k, 0, 0) * rt.Matrix4.scale(s, s, s)
            g.add_child(rt.Object(rt.CUBE, transformation=t))
            ow.add(orc.CUBE, orc.mul(orc.translate(0.5 * k, 0, 0), orc.scale(s, s, s)), None, og, 0.0, 0.0, False)
    w.objects = [g]
    ow.finalize()
    rays = []
    for k in range(24):
        if k % 3 == 0:
            continue
        tilt = 4e-10 * k
        for off in (-30.0, -4.0, 1.5, 12.0):
            for eps in (0.0, 2e-7, -3e-7):  # a = dx^2 - dy^2 + dz^2 in the cone's space: 0 and a little either side of it
                sx = 1.0 if off < 0 else -1.0
                dx, dy = sx * math.cos(math.pi / 4 + eps + sx * tilt), -math.sin(math.pi / 4 + eps + sx * tilt)
                # aimed at a point 0.25 cone units above the apex
                rays.append([0.5 * k - dx * abs(off), 0.25 * s - dy * abs(off), 0.003, dx, dy, 0.0])
    rays = np.array(rays)
    assert sum(ow.hit(orc.ray(r[:3], r[3:])) is not None for r in rays) >= 20
    _lone_rays_agree(orc, w.scene(), ow, rays)


def test_ties_between_objects_in_reordered_lists(gpu, rt, orc):
    """Long sibling lists are emitted in spatial order when that makes their range-skip boxes much smaller; the hit
    among exactly equal t's must still be the reference's, the object listed first in the scene (world.rs:34 sorts
    stably).  Pairs of unit spheres half a unit apart in y, hit by a ray halfway between them: the same t to the
    last bit, from two different objects whose order in the file is random.  The pairs stand so close together along
    x that the spatial order has to separate the upper spheres from the lower ones, which reverses about half the
    pairs."""
    import tomli

    rng = np.random.default_rng(17)
    n_pairs = 24
    centres = [(0.05 * k, 0.0, 0.0) for k in range(n_pairs)] + [(0.05 * k, 0.5, 0.0) for k in range(n_pairs)]
    order = rng.permutation(len(centres))
    # A frame one pixel high: half_height = half_view / 64 and (0 + 0.5) * pixel_size = half_view / 64 are the same
    # double, so every camera ray lies exactly in the plane y = 0.25 between the two spheres of each pair.
    lines = ["[camera]", "width = 512", "height = 1", "fov = 20.0", "from = [0.575, 0.25, -20.0]", "to = [0.575, 0.25, 0.0]",
             "up = [0.0, 1.0, 0.0]", "[light]", "position = [-10.0, 10.0, -10.0]", "intensity = [1.0, 1.0, 1.0]"]
    for i in order:
        c = centres[i]
        lines += ["[[objects]]", 'type = "sphere"', "[objects.transform]", f"translate = [{c[0]}, {c[1]}, {c[2]}]",
                  "[objects.material]", f"color = [{0.2 + 0.03 * (i % 24)}, {0.9 - 0.03 * (i % 24)}, {0.5 if i < 24 else 0.1}]",
                  "reflectivity = 0.3"]
    text = "\n".join(lines) + "\n"
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    scene = world.scene()
    rays = np.array([[0.05 * k, 0.25, -5.0, 0.0, 0.0, 1.0] for k in range(n_pairs)])
    t, obj = scene.intersect(rays)
    lists = scene.intersect_all(rays)
    first_of_pair_wins = 0
    for k, ray in enumerate(rays):
        want = oworld.intersect(orc.ray(ray[:3], ray[3:]))
        assert want[0][0] == want[1][0] and want[0][1] != want[1][1], "the pair must tie exactly"
        assert (t[k], obj[k]) == want[0], (k, (t[k], obj[k]), want[:2])
        assert lists[k] == [(a, b) for a, b in want]
        first_of_pair_wins += int(obj[k] == min(want[0][1], want[1][1]))
    assert first_of_pair_wins == n_pairs
    img, st = scene.render(cam, stats=True)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=1, counters=oc)
    tied = 0
    for x in range(cam.hsize):
        r = orc.ray_for_pixel(ocam, x, 0)
        xs = oworld.intersect(r)
        tied += int(len(xs) >= 2 and xs[0][0] == xs[1][0] and xs[0][1] != xs[1][1])
    assert tied >= 100, "the frame is meant to consist of tied primary hits"
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    plain, _ = scene.render(cam)
    assert np.array_equal(plain, img)


def test_groups_nested_twelve_deep(gpu, rt, orc):
    """World::collect_intersections recurses without a limit (world.rs:41-78).  A chain of twelve groups, each with a
    transformation of its own, two leaves beside the next group and a leaf after it (so that the traversal has to come
    back up to every level and go on with that level's ray): frame, counters, lone rays and normals against the
    oracle.  Levels two and deeper park their rays in the per-lane scratch array, level one in LDS."""
    import tomli

    lines = ["[camera]", "width = 96", "height = 64", "fov = 50.0", "from = [2.0, 3.0, -14.0]", "to = [1.0, 1.0, 0.0]",
             "up = [0.0, 1.0, 0.0]", "[light]", "position = [-8.0, 12.0, -10.0]", "intensity = [1.0, 1.0, 1.0]",
             "[[objects]]", 'type = "plane"', "[objects.transform]", "translate = [0.0, -2.0, 0.0]",
             "[objects.material]", "reflectivity = 0.3"]
    prefix = "objects"
    kinds = ("sphere", "cube", "cylinder", "cone")
    for level in range(12):
        lines += [f"[[{prefix}]]", 'type = "group"', f"[{prefix}.transform]",
                  f"translate = [{0.35 if level else -2.0}, {0.3 if level % 2 else -0.1}, {0.2 * (level % 3)}]",
                  f"rotate_y = {17.0 * (level + 1)}", f"rotate_z = {5.0 * level}", f"scale = [{0.95 + 0.01 * level}, 1.02, {1.0 - 0.01 * level}]"]
        child = prefix + ".children"
        for k in range(2):
            kind = kinds[(2 * level + k) % 4]
            lines += [f"[[{child}]]", f'type = "{kind}"']
            if kind in ("cylinder", "cone"):
                lines += ["minimum = -0.5", "maximum = 0.5", "closed = true"]
            lines += [f"[{child}.transform]", f"translate = [{1.1 * (k + 1)}, {0.6 * k}, {-0.8 * k}]", "scale = [0.45, 0.45, 0.45]",
                      f"[{child}.material]", f"color = [{0.2 + 0.06 * level}, {0.9 - 0.05 * level}, {0.3 + 0.3 * k}]",
                      f"reflectivity = {0.4 if k else 0.0}", f"transparency = {0.0 if k else 0.5}", "refractive_index = 1.3"]
        prefix = child  # the next group is a child of this one ...
    # ... and after the chain, going back up, every level gets one more leaf behind its nested group
    for level in range(11, -1, -1):
        child = "objects" + ".children" * (level + 1)
        lines += [f"[[{child}]]", 'type = "sphere"', f"[{child}.transform]", f"translate = [-0.9, {0.5 + 0.1 * level}, 0.7]",
                  "scale = [0.3, 0.3, 0.3]", f"[{child}.material]", f"color = [0.9, {0.08 * level}, 0.2]"]
    text = "\n".join(lines) + "\n"
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    scene = world.scene()
    assert scene.info().max_nesting == 12
    img, st = scene.render(cam, stats=True)
    plain, _ = scene.render(cam)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=8, counters=oc)
    assert_kernels_agree(scene, plain, img)
    assert_frames_match(orc, img, ref)
    assert_frames_match(orc, plain, ref)
    counters_equal(st, oc)
    assert st.reflect > 0 and st.refract > 0 and st.aabb_tests > 12 * 96 * 64 // 8
    rays = _probe_rays(orc, ocam, 31, 200, 200, ((-4, -1, -4), (6, 5, 4)))
    _lone_rays_agree(orc, scene, oworld, rays)
    # one level deeper than the limit is refused when the world is built, not rendered wrongly
    deeper = text[:text.index("[[objects]]")]
    prefix = "objects"
    for level in range(rt.MAX_GROUP_NESTING + 1):
        deeper += f"[[{prefix}]]\ntype = \"group\"\n"
        prefix += ".children"
    deeper += f"[[{prefix}]]\ntype = \"sphere\"\n"
    with pytest.raises(rt.RtError) as e:
        rt.SceneFile.from_str(deeper).build_world().scene()
    assert e.value.code in (-6, -7)


def test_degenerate_worlds(gpu, rt, orc):
    """Empty world, a childless group, no light source, and a flat list long enough to get range-skip nodes."""
    cam, ocam = rt.Camera(40, 30, 1.0), orc.camera(40, 30, 1.0)

    def both(w, ow):
        ow.finalize()
        img, st = w.scene().render(cam, stats=True)
        plain, _ = w.scene().render(cam)
        oc = orc.Counters()
        ref = orc.render(ocam, ow, counters=oc)
        assert np.array_equal(plain, img)
        assert_frames_match(orc, img, ref)
        counters_equal(st, oc)
        return img

    w, ow = rt.World(), orc.World()
    w.light_source = rt.PointLight((-10, 10, -10), (1, 1, 1))
    ow.set_light((-10, 10, -10), (1, 1, 1))
    assert not both(w, ow).any()
    w, ow = rt.World(), orc.World()
    w.light_source = rt.PointLight((-10, 10, -10), (1, 1, 1))
    ow.set_light((-10, 10, -10), (1, 1, 1))
    w.objects = [rt.Object.group(), rt.Object.sphere(transformation=rt.Matrix4.translate(0, 0, -5))]
    ow.add(orc.GROUP, orc.identity())
    ow.add(orc.SPHERE, orc.translate(0, 0, -5))
    assert both(w, ow).any()
    w, ow = rt.World(), orc.World()  # world.rs:86-89: no light, everything black, but the camera rays are still cast
    w.objects = [rt.Object.sphere(transformation=rt.Matrix4.translate(0, 0, -5))]
    ow.add(orc.SPHERE, orc.translate(0, 0, -5))
    assert not both(w, ow).any()
    w, ow = rt.World(), orc.World()
    w.light_source = rt.PointLight((-10, 10, -10), (1, 1, 1))
    ow.set_light((-10, 10, -10), (1, 1, 1))
    for i in range(40):
        x, y = float(i % 10) - 4.5, float(i // 10) - 1.5
        w.objects.append(rt.Object.sphere(transformation=rt.Matrix4.translate(x, y, -6.0) * rt.Matrix4.scale(0.4, 0.4, 0.4)))
        ow.add(orc.SPHERE, orc.mul(orc.translate(x, y, -6.0), orc.scale(0.4, 0.4, 0.4)))
    assert both(w, ow).any()


# --------------------------------------------------------------------------- rows, bands, sizes, depth
def test_row_ranges_and_bands_tile_the_frame(gpu, rt, orc):
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_11", 203, 77)
    scene = world.scene()
    full, st_full = scene.render(cam, stats=True)
    assert_frames_match(orc, full, orc.render(ocam, oworld, threads=8))
    part = np.full_like(full, -1.0)
    scene.render(cam, rows=(10, 45), out=part)
    assert np.array_equal(part[10:45], full[10:45]) and (part[:10] == -1).all() and (part[45:] == -1).all()
    for band_rows, world_size in ((16, 2), (16, 8), (4, 3), (32, 5)):
        acc = np.full_like(full, -1.0)
        rays = 0
        for r in range(world_size):
            _, st = scene.render(cam, bands=(band_rows, r, world_size), out=acc, stats=True)
            rays += st.rays
        assert np.array_equal(acc, full)
        assert rays == st_full.rays
    with pytest.raises(rt.RtError):
        scene.render(cam, rows=(50, 200))


def test_device_buffer_entry_point_and_band_packing(gpu, rt, orc):
    """rt_render_device: the asynchronous entry point the multi-GPU driver uses.  Rows and bands arrive packed in
    device memory; put back in place they are the frame rt_render returns."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    cam, world, _, _ = load_both(rt, orc, "chapter_11", 203, 77)
    scene = world.scene()
    full, _ = scene.render(cam)
    W, H = cam.hsize, cam.vsize
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), full.nbytes) == 0
    try:
        def fetch(n_rows):
            out = np.zeros((n_rows, W, 3))
            assert hip.hipDeviceSynchronize() == 0
            assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), buf, out.nbytes, 2) == 0
            return out

        scene.render_device(cam, buf.value)
        assert np.array_equal(fetch(H), full)
        scene.render_device(cam, buf.value, rows=(10, 45))
        assert np.array_equal(fetch(35), full[10:45])
        for band_rows, step in ((16, 2), (8, 3), (5, 4)):
            for first in range(step):
                scene.render_device_bands(cam, buf.value, band_rows, first, step)
                bands = list(range(first, (H + band_rows - 1) // band_rows, step))
                rows = sum(min(H, (b + 1) * band_rows) - b * band_rows for b in bands)
                got = fetch(rows)
                at = 0
                for b in bands:
                    y0, y1 = b * band_rows, min(H, (b + 1) * band_rows)
                    # every band occupies band_rows rows of the packed buffer except that the frame's last band may be short
                    assert np.array_equal(got[at:at + y1 - y0], full[y0:y1]), (band_rows, step, first, b)
                    at += band_rows if y1 - y0 == band_rows else y1 - y0
    finally:
        hip.hipFree(buf)


def test_render_kernel_quantises_its_own_output(gpu, rt, orc):
    """RT_RENDER_QUANTIZED: the 8-bit frame straight from the render kernel is the host quantiser (canvas.rs:123-130)
    over the f64 frame, for whole frames, row ranges and bands; the launch can be timed (RT_RENDER_TIMED)."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    for name, size in (("chapter_11", (203, 77)), ("chapter_10", (96, 64)), ("chapter_14", (120, 90))):
        cam, world, _, _ = load_both(rt, orc, name, *size)
        scene = world.scene()
        full, _ = scene.render(cam)
        expect = orc.quantize(full)
        W, H = cam.hsize, cam.vsize
        buf = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(buf), expect.size + 64) == 0
        try:
            def fetch(n_rows):
                out = np.zeros(n_rows * W * 3 + 64, dtype=np.uint8)
                assert hip.hipDeviceSynchronize() == 0
                assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), buf, out.size, 2) == 0
                assert not out[n_rows * W * 3:].any(), "wrote past the rows rendered"
                return out[:n_rows * W * 3].reshape(n_rows, W, 3)

            assert hip.hipMemset(buf, 0, expect.size + 64) == 0
            scene.render_device(cam, buf.value, quantized=True, timed=True)
            assert np.array_equal(fetch(H), expect)
            ordering_ms, render_ms = scene.render_times()
            assert 0.0 <= ordering_ms < 50.0 and 0.0 < render_ms < 500.0
            assert hip.hipMemset(buf, 0, expect.size + 64) == 0
            scene.render_device(cam, buf.value, rows=(10, 45), quantized=True)
            assert np.array_equal(fetch(35), expect[10:45])
            assert hip.hipMemset(buf, 0, expect.size + 64) == 0
            scene.render_device_bands(cam, buf.value, 16, 1, 2, quantized=True)
            bands = list(range(1, (H + 15) // 16, 2))
            rows = sum(min(H, (b + 1) * 16) - b * 16 for b in bands)
            got, at = fetch(rows), 0
            for b in bands:
                y0, y1 = b * 16, min(H, (b + 1) * 16)
                assert np.array_equal(got[at:at + y1 - y0], expect[y0:y1])
                at += y1 - y0
        finally:
            hip.hipFree(buf)
    with pytest.raises(rt.RtError):  # no timed launch on this handle yet
        load_both(rt, orc, "chapter_6")[1].scene().render_times()


@pytest.mark.parametrize("size", [(1, 1), (17, 9), (33, 65), (64, 1), (1, 40), (250, 130)])
def test_odd_frame_sizes(gpu, rt, orc, size):
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_13", *size)
    img, st = world.scene().render(cam, stats=True)
    oc = orc.Counters()
    assert_frames_match(orc, img, orc.render(ocam, oworld, counters=oc))
    counters_equal(st, oc)


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 5, 7, 8])
def test_recursion_budget(gpu, rt, orc, depth):
    """World::color_at_internal(ray, remaining) for other budgets than the default 5 (world.rs:117)."""
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_11", 60, 45)
    px = [x for y in range(45) for x in range(60)]
    py = [y for y in range(45) for x in range(60)]
    rays = np.array([list(r.origin) + list(r.direction) for r in cam.rays_for_pixels(px, py)])
    got, st = world.scene().color_at(rays, remaining=depth, stats=True)
    oc = orc.Counters()
    ref = np.array([oworld.color_at(orc.ray(r[:3], r[3:]), depth, oc) for r in rays])
    assert np.abs(got - ref).max() <= GUARD
    assert st.rays == oc.rays and st.reflect == oc.reflect and st.refract == oc.refract


def test_several_workers_share_one_frame(gpu, rt, orc):
    """rt_multi_*: the frame cut into bands that the workers take from a shared cursor.  Device 0 listed two and three
    times gives several workers (host threads, streams, scene handles) on the one GPU a test box has; whatever the
    interleaving, the frame must be the single-device frame bit for bit, in device memory (f64 and 8-bit) and in host
    memory, and the counters must add up to the single-device counters."""
    import ctypes
    import subprocess

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    for name, (w, h) in (("chapter_11", (403, 301)), ("chapter_14", (320, 203))):
        cam, world, _, _ = load_both(rt, orc, name, w, h)
        full, st_full = world.scene().render(cam, stats=True)
        buf = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(buf), full.nbytes + 64) == 0
        try:
            for devices, band_rows, units in (([0, 0], 0, 0), ([0, 0, 0], 8, 5), ([0], 24, 0), ([0, 0], 1000, 0), ([0, 0], 16, 400)):
                multi = world.multi_scene(devices)
                # frame in device memory, f64
                assert hip.hipMemset(buf, 0xA5, full.nbytes + 64) == 0
                st = multi.render_device(cam, buf.value, band_rows=band_rows, stats=True, units=units)
                got = np.zeros(full.nbytes + 64, dtype=np.uint8)
                assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), buf, got.nbytes, 2) == 0
                assert np.array_equal(got[:full.nbytes].view(np.float64).reshape(full.shape), full), (name, devices, band_rows)
                assert (got[full.nbytes:] == 0xA5).all(), "wrote past the frame"
                assert st.as_dict() == st_full.as_dict()
                info = multi.info()
                # no stripe height named: 16 rows (one worker alone: the frame as one launch); units never outnumber stripes
                stripes = (h + info["band_rows"] - 1) // info["band_rows"]
                assert info["band_rows"] == (band_rows or ((h + 7) // 8 * 8 if len(devices) == 1 else 16))
                assert info["n_units"] == (min(units, stripes) if units else 1 if len(devices) == 1 else info["n_units"]) and 1 <= info["n_units"] <= stripes
                assert (info["n_workers"], info["n_devices"], info["gather"], info["rccl_ranks"]) == (len(devices), 1, "none", 0)
                assert sum(multi.units_taken()) == info["n_units"] and len(multi.units_taken()) == len(devices)
                assert st.kernel_ms > 0.0 and st.gather_ms >= 0.0 and max(multi.worker_ms()) > 0.0
                # 8-bit frame
                st = multi.render_device(cam, buf.value, band_rows=band_rows, quantized=True, units=units)
                q = np.zeros(w * h * 3, dtype=np.uint8)
                assert hip.hipMemcpy(q.ctypes.data_as(ctypes.c_void_p), buf, q.nbytes, 2) == 0
                assert np.array_equal(q.reshape(h, w, 3), orc.quantize(full))
                # frame in host memory
                host, st = multi.render(cam, band_rows=band_rows, stats=True, units=units)
                assert np.array_equal(host, full) and st.rays == st_full.rays
                multi.close()
        finally:
            hip.hipFree(buf)
    # refused: row ranges, unknown devices
    cam, world, _, _ = load_both(rt, orc, "chapter_6", 64, 64)
    with pytest.raises(rt.RtError):
        world.multi_scene([0, 99])
    # the command-line renderer with --devices writes the reference's own file
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "raytracer_rs_amd", "rt_render")
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rt_multi_chapter_6.ppm")
    subprocess.run([exe, scene_path("chapter_6"), "-o", out, "--devices", "0,0"], check=True, capture_output=True)
    import hashlib
    import json
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        manifest = json.load(f)
    digest = hashlib.sha256(open(out, "rb").read()).hexdigest()
    assert digest == manifest["ppm"]["chapter_6_sphere"]["sha256_p3_text"], "the PPM written through --devices is not the reference's file"


def test_render_with_zero_budget(gpu, rt, orc):
    """Camera::render with `remaining` = 0 (world.rs:117): no reflection or refraction ray is spawned at all.  Through
    rt_render and rt_render_device the budget 0 needs RT_RENDER_EXACT_DEPTH (zeroed options mean the default 5)."""
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_11", 120, 90)
    scene = world.scene()
    for depth in (0, 1, 5):
        img, st = scene.render(cam, max_depth=depth, stats=True)
        oc = orc.Counters()
        px = [x for y in range(90) for x in range(120)]
        py = [y for y in range(90) for x in range(120)]
        rays = np.array([list(r.origin) + list(r.direction) for r in cam.rays_for_pixels(px, py)])
        ref = np.array([oworld.color_at(orc.ray(r[:3], r[3:]), depth, oc) for r in rays]).reshape(90, 120, 3)
        assert np.abs(img - ref).max() <= GUARD
        assert (st.rays, st.reflect, st.refract) == (oc.rays, oc.reflect, oc.refract)
        if depth == 0:
            assert st.reflect == 0 and st.refract == 0
    default, _ = scene.render(cam)
    five, _ = scene.render(cam, max_depth=5)
    assert np.array_equal(default, five)


def test_hits_from_inside_glass_balls_in_their_own_shadow(gpu, rt, orc, monkeypatch):
    """The render kernels walk no shadow ray (and compute no lit colour) for a hit from inside a sphere that the light is
    well outside of.  Glass balls for which that proof must hold and balls for which it must not be attempted -- the
    light INSIDE one, the light just outside another, a ball a hundred units across (the point the shadow ray starts
    from is then inside by less than the margin demanded), a tiny one, stretched ones, balls inside scaled and rotated
    groups -- against the oracle, as a single launch and level by level."""
    import tomli

    glass = """
[objects.material]
color = [0.9, 0.95, 1.0]
diffuse = 0.3
ambient = 0.1
specular = 0.9
shininess = 200.0
reflectivity = 0.6
transparency = 0.8
refractive_index = 1.4
"""
    kid_glass = glass.replace("[objects.material]", "[objects.children.material]")

    def ball(translate, scale, extra=""):
        return f"""
[[objects]]
type = "sphere"
[objects.transform]
scale = [{scale[0]}, {scale[1]}, {scale[2]}]
{extra}translate = [{translate[0]}, {translate[1]}, {translate[2]}]
{glass}"""

    text = """
[camera]
width = 192
height = 128
fov = 70.0
from = [0.0, 2.0, -9.0]
to = [0.0, 1.0, 0.0]
up = [0.0, 1.0, 0.0]
[light]
position = [-2.0, 1.0, -1.0]
intensity = [1.0, 1.0, 1.0]
[[objects]]
type = "plane"
[objects.material]
color = [0.8, 0.8, 0.8]
reflectivity = 0.2
"""
    text += ball((-2.0, 1.0, -1.0), (1.0, 1.0, 1.0))            # the light is at its centre
    text += ball((-0.2, 1.0, -1.0), (1.0, 1.0, 1.0))            # the light 1.8 radii from its centre: outside, not "well outside"
    text += ball((2.5, 1.0, 0.0), (1.0, 1.0, 1.0))              # an ordinary one
    text += ball((0.5, 0.3, -3.0), (0.3, 0.3, 0.3))
    text += ball((4.0, 0.05, -4.0), (0.05, 0.05, 0.05))         # tiny (much smaller and the reference calls its matrix singular)
    text += ball((0.0, 50.0, 120.0), (50.0, 50.0, 50.0))        # huge
    text += ball((-4.0, 0.8, 1.0), (1.5, 0.8, 0.4), "rotate_y = 35.0\n")  # stretched and turned
    text += f"""
[[objects]]
type = "group"
[objects.transform]
scale = [0.5, 2.0, 0.5]
rotate_z = 20.0
translate = [1.0, 1.0, 2.5]
[[objects.children]]
type = "sphere"
{kid_glass}
[[objects.children]]
type = "group"
[objects.children.transform]
rotate_x = 40.0
translate = [2.5, 0.0, 0.0]
[[objects.children.children]]
type = "sphere"
[objects.children.children.transform]
scale = [0.7, 0.7, 1.2]
{glass.replace("[objects.material]", "[objects.children.children.material]")}
"""
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    ref = orc.render(ocam, oworld, threads=16)
    for levels in ("0", "1"):
        monkeypatch.setenv("RT_RENDER_LEVELS", levels)
        sf = rt.SceneFile.from_str(text)
        cam, world = sf.build_camera(), sf.build_world()
        scene = world.scene()
        img, _ = scene.render(cam)
        assert_frames_match(orc, img, ref)
        counted, st = scene.render(cam, stats=True)  # (the counting kernel walks every shadow ray)
        assert np.abs(img - counted).max() <= 1e-13
        assert st.refract > 0 and st.shadow > 0
        world.release_scene()
    # ... and the same balls without the group, through the kernel for worlds without groups
    flat = text[:text.index('[[objects]]\ntype = "group"')]
    ocam, oworld = orc.scene_from_dict(tomli.loads(flat))
    sf = rt.SceneFile.from_str(flat)
    cam, world = sf.build_camera(), sf.build_world()
    scene = world.scene()
    assert scene.info().kernel_build == 0
    img, _ = scene.render(cam)
    assert_frames_match(orc, img, orc.render(ocam, oworld, threads=16))
    counted, _ = scene.render(cam, stats=True)
    assert np.array_equal(img, counted)


@pytest.mark.parametrize("seed", [501, 502, 505, 511, 517, 523, 531, 540, 556, 577, 579])
def test_random_worlds_without_groups(gpu, rt, orc, seed):
    """The table-of-leaves kernels on random worlds of up to twelve objects: glass and mirror balls (stretched, turned),
    cubes, cylinders, a floor that is transparent now and then, the light above the scene, close to a ball or inside
    one.  Against the oracle, and the plain kernel (shadow walks that start at the object left and stop early, hits
    proved to lie in their own ball's shadow) bit for bit against the counting kernel (which walks everything)."""
    import tomli

    from raytracer_rs_amd import synthetic

    text = synthetic.flat_scene(seed)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    scene = world.scene()
    assert scene.info().kernel_build in (0, 4)
    img, _ = scene.render(cam)
    oc = orc.Counters()
    assert_frames_match(orc, img, orc.render(ocam, oworld, threads=16, counters=oc))
    counted, st = scene.render(cam, stats=True)
    counters_equal(st, oc)
    assert np.array_equal(img, counted)


# --------------------------------------------------------------------------- full size
def test_full_4k_frame(gpu, rt, orc):
    """The benchmark configuration itself: chapter_11 at 3840x2160, whole frame against the oracle, ray count exact,
    two renders bit-identical, bands of 16 rows over 8 ranks reassemble the same frame."""
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_11", 3840, 2160)
    scene = world.scene()
    img, st = scene.render(cam, stats=True)
    again, _ = scene.render(cam)
    assert np.array_equal(img, again)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=16, counters=oc)
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    assert st.rays == st.primary + st.shadow + st.reflect + st.refract and st.primary == 3840 * 2160
    acc = np.zeros_like(img)
    for r in range(8):
        scene.render(cam, bands=(16, r, 8), out=acc)
    assert np.array_equal(acc, img)


def test_which_build_of_the_render_kernel_serves_which_world(gpu, rt, monkeypatch):
    """rt_scene_info.kernel_build: 0 = a table of leaves for the book's flat scenes; every world with groups is rendered
    level by level of its ray trees (3) unless RT_RENDER_LEVELS=0, read when a scene is created, asks for the single
    launch: then the build without cone directions for chapter_14 (groups, spheres and cylinders: 2) and the general one
    for hierarchies with cones under their range-skip nodes (1)."""
    from raytracer_rs_amd import synthetic

    def build(world):
        k = world.scene().info().kernel_build
        world.release_scene()
        return k

    ch11 = rt.SceneFile.from_file(scene_path("chapter_11")).build_world()
    ch14 = rt.SceneFile.from_file(scene_path("chapter_14")).build_world()
    large = rt.SceneFile.from_str(synthetic.large_scene(64, 36)).build_world()
    assert (build(ch11), build(ch14), build(large)) == (0, 3, 3)
    # nothing in chapter_7_and_8 reflects or transmits: the kernel of one round
    assert build(rt.SceneFile.from_file(scene_path("chapter_7_and_8")).build_world()) == 4
    # thirteen cubes and no group: a table with range-skip nodes, but a single launch unless levels are asked for
    ch12 = rt.SceneFile.from_file(scene_path("chapter_12")).build_world()
    assert build(ch12) in (1, 2)
    monkeypatch.setenv("RT_RENDER_LEVELS", "1")
    assert build(ch12) == 3
    monkeypatch.setenv("RT_RENDER_LEVELS", "0")
    assert (build(ch11), build(ch14), build(large)) == (0, 2, 1)
    monkeypatch.setenv("RT_RENDER_LEVELS", "1")
    assert (build(ch11), build(ch14), build(large)) == (0, 3, 3)


def test_full_hd_world_with_shadows(gpu, rt, orc):
    """chapter_7_and_8 (six spheres, point light, shadow rays) at 1920x1080, the size the benchmark configuration
    names: whole frame against the oracle, every counter, and exactly two rays per pixel only where geometry is hit."""
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_7_and_8", 1920, 1080)
    scene = world.scene()
    img, st = scene.render(cam, stats=True)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=16, counters=oc)
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    assert st.primary == 1920 * 1080 and st.reflect == 0 and st.refract == 0
    assert st.rays == st.primary + st.shadow and st.shadow == st.shades <= st.primary
    plain, _ = scene.render(cam)
    assert np.array_equal(plain, img)


def _full_size_checks(orc, scene, cam, ocam, oworld, n_sample_bands, band_rows=4):
    """What can be checked on a frame too large for the oracle to render whole: row bands spread over the frame
    against the oracle (colours and, for the same rows, every counter), of the counting kernel's frame and of the plain
    path's; the two agree on all of it; the ray count adds up; 16-row bands dealt to 8 ranks reassemble the frame."""
    h, w = cam.vsize, cam.hsize
    counted, st = scene.render(cam, stats=True)
    plain, _ = scene.render(cam)
    assert_kernels_agree(scene, plain, counted)
    assert st.primary == w * h and st.rays == st.primary + st.shadow + st.reflect + st.refract
    work = plain.copy()
    for k in range(n_sample_bands):
        y0 = ((h - band_rows) * k // (n_sample_bands - 1)) // band_rows * band_rows
        oc = orc.Counters()
        ref = orc.render(ocam, oworld, threads=16, rows=(y0, y0 + band_rows), counters=oc)
        assert_frames_match(orc, counted[y0:y0 + band_rows], ref[y0:y0 + band_rows])
        assert_frames_match(orc, plain[y0:y0 + band_rows], ref[y0:y0 + band_rows])
        del ref
        _, pst = scene.render(cam, rows=(y0, y0 + band_rows), stats=True, out=work)  # rows of the whole-frame buffer
        counters_equal(pst, oc)
        scene.render(cam, rows=(y0, y0 + band_rows), out=work)
        if by_levels(scene):
            # level by level a pixel's colour is a function of its own ray tree: any rows, the same bits
            assert np.array_equal(work[y0:y0 + band_rows], plain[y0:y0 + band_rows])
        else:
            # Rows that do not start and end on the 8-row tile grid put a pixel in a tile with other neighbours; in
            # scenes with groups lanes hand rays to each other, so the terms of a pixel's sum may be added in another order.
            assert np.abs(work[y0:y0 + band_rows] - plain[y0:y0 + band_rows]).max() <= 1e-13
        t0 = y0 // 8 * 8
        scene.render(cam, rows=(t0, t0 + 8), out=work)
        assert np.array_equal(work[t0:t0 + 8], plain[t0:t0 + 8])
    work.fill(0.0)
    for r in range(8):
        scene.render(cam, bands=(16, r, 8), out=work)
    assert np.array_equal(work, plain)


def test_full_4k_hexagon_groups(gpu, rt, orc):
    """chapter_14 (327 leaves under 17 groups) at 3840x2160: the WHOLE frame against the oracle, every counter, then
    the checks on row ranges and bands."""
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_14", 3840, 2160)
    scene = world.scene()
    img, st = scene.render(cam, stats=True)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=16, counters=oc)
    assert_frames_match(orc, img, ref)
    counters_equal(st, oc)
    del ref, img
    _full_size_checks(orc, scene, cam, ocam, oworld, n_sample_bands=6)


def test_full_8k_generated_hierarchy(gpu, rt, orc):
    """The 10 001-leaf, three-level hierarchy at 7680x4320."""
    import tomli

    from raytracer_rs_amd import synthetic

    text = synthetic.large_scene()
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    assert (cam.hsize, cam.vsize) == (7680, 4320)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    _full_size_checks(orc, world.scene(), cam, ocam, oworld, n_sample_bands=16)


def test_frame_beyond_four_gigabytes(gpu, rt, orc):
    """Offsets into the frame that need more than 32 bits: chapter_7_and_8 at 16384x12288 is 4.8 GB of f64 (4.2 million
    tiles).  Rendered into device memory; the rows at the top, either side of the 4 GiB mark and at the bottom are
    compared with the oracle and with the same rows rendered alone, and nothing is written past the end."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    W, H = 16384, 12288
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_7_and_8", W, H)
    scene = world.scene()
    row_bytes = W * 3 * 8
    total = H * row_bytes
    assert total > 1 << 32
    mark = (1 << 32) // row_bytes  # the row the 4 GiB offset falls in
    guard = 4096
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), total + guard) == 0
    try:
        assert hip.hipMemset(ctypes.c_void_p(buf.value + total), 0x5A, guard) == 0
        scene.render_device(cam, buf.value)
        assert hip.hipDeviceSynchronize() == 0
        tail = np.zeros(guard, dtype=np.uint8)
        assert hip.hipMemcpy(tail.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(buf.value + total), guard, 2) == 0
        assert (tail == 0x5A).all(), "wrote past the end of the frame"
        for y0 in (0, mark - 2, mark + 1, H // 2, H - 4):
            got = np.zeros((4, W, 3))
            assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(buf.value + y0 * row_bytes), got.nbytes, 2) == 0
            ref = orc.render(ocam, oworld, threads=16, rows=(y0, y0 + 4))
            assert_frames_match(orc, got, ref[y0:y0 + 4])
            del ref
            scene.render_device(cam, buf.value, rows=(y0, y0 + 4))  # packed at the start of the buffer
            assert hip.hipDeviceSynchronize() == 0
            alone = np.zeros((4, W, 3))
            assert hip.hipMemcpy(alone.ctypes.data_as(ctypes.c_void_p), buf, alone.nbytes, 2) == 0
            assert np.array_equal(alone, got)
    finally:
        hip.hipFree(buf)
