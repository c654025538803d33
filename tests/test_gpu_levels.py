"""Worlds rendered level by level of their ray trees (rt_scene_info.kernel_build 3): the path large hierarchies take by
default, forced here (RT_RENDER_LEVELS=1, read when a scene is created) onto every kind of world with groups the suite
has, so that its queues, its numbering of rays and its sums are held to the oracle on nested groups, cones under
range-skip nodes, reordered sibling lists and ties -- not only on the one generated scene that takes it by itself."""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import assert_frames_match, counters_equal, load_both, scene_path

pytestmark = pytest.mark.gpu


@pytest.fixture
def levels(monkeypatch):
    monkeypatch.setenv("RT_RENDER_LEVELS", "1")


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    return hip


def _fetch(hip, ptr, shape, dtype):
    out = np.zeros(shape, dtype=dtype)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ptr, out.nbytes, 2) == 0
    return out


@pytest.mark.parametrize("sort", [True, False])
@pytest.mark.parametrize("scene", ["chapter_12", "chapter_14"])
def test_book_scenes_with_groups_level_by_level(gpu, rt, orc, levels, monkeypatch, scene, sort):
    if not sort:
        monkeypatch.setenv("RT_LEVELS_SORT", "0")  # a level's rays left in the order of their parent waves
    cam, world, ocam, oworld = load_both(rt, orc, scene)
    s = world.scene()
    assert s.info().kernel_build == 3
    plain, _ = s.render(cam)
    counted, st = s.render(cam, stats=True)  # (the counting kernel stays the single launch)
    oc = orc.Counters()
    ref = orc.render(ocam, oworld, threads=16, counters=oc)
    assert_frames_match(orc, plain, ref)
    counters_equal(st, oc)
    assert np.abs(plain - counted).max() <= 1e-13
    again, _ = s.render(cam)
    assert np.array_equal(again, plain), "two renders of the same frame differ"


@pytest.mark.parametrize("sort", [True, False])
@pytest.mark.parametrize("seed,n_top", [(1, 6), (4, 6), (9, 6), (12, 6), (101, 25), (103, 60), (104, 90)])
def test_random_worlds_level_by_level(gpu, rt, orc, levels, monkeypatch, seed, n_top, sort):
    """Random hierarchies (every shape, cones under range-skip nodes, nested groups, patterns, glass and mirrors), with
    every level's rays sorted by kind, direction octant and the leaf they leave (the default) and left in the order of
    their parent waves (RT_LEVELS_SORT=0, read per call)."""
    import tomli

    from raytracer_rs_amd import synthetic

    if not sort:
        monkeypatch.setenv("RT_LEVELS_SORT", "0")
    text = synthetic.random_scene(seed, 96, 64, n_top)
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    s = world.scene()
    if s.info().max_nesting == 0:
        pytest.skip("this seed has no groups: the single launch serves it whatever the environment says")
    assert s.info().kernel_build == 3
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    plain, _ = s.render(cam)
    ref = orc.render(ocam, oworld, threads=8)
    assert_frames_match(orc, plain, ref)


def test_every_way_to_cut_a_frame_gives_the_same_bits(gpu, rt, orc, levels):
    """Level by level a pixel's colour is a function of its own ray tree alone: whole frame, row ranges on and OFF the
    8-row tile grid, bands, rows stored in place, the 8-bit frame and the host frame all carry the same values."""
    hip = _hip()
    w, h = 403, 277
    cam, world, _, _ = load_both(rt, orc, "chapter_14", w, h)
    s = world.scene()
    full, _ = s.render(cam)
    work = np.zeros_like(full)
    for rows in ((0, 8), (3, 20), (101, 102), (270, 277), (16, 200)):
        s.render(cam, rows=rows, out=work)
        assert np.array_equal(work[rows[0]:rows[1]], full[rows[0]:rows[1]]), rows
    work.fill(0.0)
    for r in range(3):
        s.render(cam, bands=(16, r, 3), out=work)
    assert np.array_equal(work, full)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), full.nbytes + 64) == 0
    try:
        assert hip.hipMemset(buf, 0xA5, full.nbytes + 64) == 0
        s.render_device(cam, buf.value)
        got = _fetch(hip, buf, full.nbytes + 64, np.uint8)
        assert np.array_equal(got[:full.nbytes].view(np.float64).reshape(full.shape), full)
        assert (got[full.nbytes:] == 0xA5).all(), "wrote past the frame"
        # stripes 1, 4, 7, ... of 16 rows, stored at their own rows of a whole frame
        assert hip.hipMemset(buf, 0, full.nbytes) == 0
        s.render_device_bands(cam, buf.value, 16, 1, 3, in_place=True)
        got = _fetch(hip, buf, full.shape, np.float64)
        for k in range((h + 15) // 16):
            rows = slice(16 * k, min(h, 16 * k + 16))
            want = full[rows] if k % 3 == 1 else np.zeros_like(full[rows])
            assert np.array_equal(got[rows], want), k
        # the 8-bit frame
        assert hip.hipMemset(buf, 0xA5, w * h * 3 + 64) == 0
        s.render_device(cam, buf.value, quantized=True)
        q = _fetch(hip, buf, w * h * 3 + 64, np.uint8)
        assert np.array_equal(q[:w * h * 3].reshape(h, w, 3), orc.quantize(full))
        assert (q[w * h * 3:] == 0xA5).all(), "wrote past the 8-bit frame"
    finally:
        hip.hipFree(buf)


def test_large_frames_are_rendered_in_chunks_of_rows(gpu, rt, orc, levels, monkeypatch):
    """Beyond a number of pixels (48 Mpixel; RT_LEVELS_CHUNK_PIXELS overrides, read per call) a frame goes through the
    queues in chunks of rows -- whole bands when the rows are bands -- and nothing about the result may change: packed
    rows, bands, rows in place, 8-bit values."""
    hip = _hip()
    w, h = 403, 277
    cam, world, _, _ = load_both(rt, orc, "chapter_14", w, h)
    s = world.scene()
    full, _ = s.render(cam)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), full.nbytes + 64) == 0
    try:
        for limit in (w * 64, w * 100, w * 16):  # chunks of 64 rows, 64 rows (rounded down), one band of 16
            monkeypatch.setenv("RT_LEVELS_CHUNK_PIXELS", str(limit))
            assert hip.hipMemset(buf, 0xA5, full.nbytes + 64) == 0
            s.render_device(cam, buf.value)
            got = _fetch(hip, buf, full.nbytes + 64, np.uint8)
            assert np.array_equal(got[:full.nbytes].view(np.float64).reshape(full.shape), full), limit
            assert (got[full.nbytes:] == 0xA5).all(), "wrote past the frame"
            s.render_device(cam, buf.value, rows=(3, 250))
            assert np.array_equal(_fetch(hip, buf, (247, w, 3), np.float64), full[3:250]), limit
            s.render_device(cam, buf.value, quantized=True)
            assert np.array_equal(_fetch(hip, buf, (h, w, 3), np.uint8), orc.quantize(full)), limit
            # stripes 2, 5, 8, ... packed one after the other, then the same stripes at their own rows
            s.render_device_bands(cam, buf.value, 16, 2, 3)
            want = np.concatenate([full[16 * k:16 * k + 16] for k in range(2, (h + 15) // 16, 3)])
            assert np.array_equal(_fetch(hip, buf, want.shape, np.float64), want), limit
            assert hip.hipMemset(buf, 0, full.nbytes) == 0
            s.render_device_bands(cam, buf.value, 16, 2, 3, in_place=True)
            got = _fetch(hip, buf, full.shape, np.float64)
            for k in range((h + 15) // 16):
                rows = slice(16 * k, min(h, 16 * k + 16))
                assert np.array_equal(got[rows], full[rows] if k % 3 == 2 else np.zeros_like(full[rows])), (limit, k)
    finally:
        hip.hipFree(buf)


def test_repeated_views_are_launched_from_remembered_counts(gpu, rt, orc, levels, monkeypatch):
    """The first frame of a view asks the GPU after every level how many rays the next one has; the handle remembers the
    answers per view (camera, rows, budget) and launches later frames of that view without waiting.  Views alternating
    through one handle -- two frame sizes, a row range, bands, another budget, more views than the handle remembers --
    must come out as they do when nothing is remembered (RT_LEVELS_REMEMBER=0)."""
    hip = _hip()
    sf = rt.SceneFile.from_file(scene_path("chapter_14"))
    world = sf.build_world()
    s = world.scene()
    cams = [sf.build_camera(160, 96), sf.build_camera(203, 131)]
    views = [(cams[0], {}), (cams[1], {}), (cams[0], {"rows": (8, 40)}), (cams[1], {"max_depth": 2}), (cams[0], {"rows": (0, 96)})]
    views += [(cams[0], {"rows": (k, k + 8)}) for k in range(0, 88, 2)]  # 44 more: beyond what a handle remembers
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), 203 * 131 * 24) == 0

    def frame(cam, kw):
        rows = kw.get("rows", (0, cam.vsize))
        s.render_device(cam, buf.value, **kw)
        return _fetch(hip, buf, (rows[1] - rows[0], cam.hsize, 3), np.float64)

    try:
        monkeypatch.setenv("RT_LEVELS_REMEMBER", "0")
        want = [frame(cam, kw) for cam, kw in views]
        monkeypatch.delenv("RT_LEVELS_REMEMBER")
        for _ in range(3):  # first pass fills the memory, later passes use it (and evict and refill part of it)
            for (cam, kw), ref in zip(views, want):
                assert np.array_equal(frame(cam, kw), ref), kw
        # the same view several times in a row: from the second frame on, the numbering of its rays is reused as well
        for k in (1, 0, 3):
            for _ in range(3):
                assert np.array_equal(frame(*views[k]), want[k]), k
        # a few views in turn (a frame rendered in row groups): the handle keeps the numbering of several
        for k in (0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 4, 0, 1):
            assert np.array_equal(frame(*views[k]), want[k]), k
        # the order of a level's rays is no part of the result: frames with the rays left unsorted, of views numbered
        # sorted before, and back
        monkeypatch.setenv("RT_LEVELS_SORT", "0")
        for k in (0, 0, 1):
            assert np.array_equal(frame(*views[k]), want[k]), k
        monkeypatch.delenv("RT_LEVELS_SORT")
        for k in (1, 1, 0):
            assert np.array_equal(frame(*views[k]), want[k]), k
        # nor is the order in which a level's waves are launched (by measured cost from a view's second frame on)
        monkeypatch.setenv("RT_LEVELS_ORDER", "0")
        for k in (0, 0, 0, 3, 3):
            assert np.array_equal(frame(*views[k]), want[k]), k
        monkeypatch.delenv("RT_LEVELS_ORDER")
        for k in (3, 3, 3, 1, 1, 1):
            assert np.array_equal(frame(*views[k]), want[k]), k
        bands = np.zeros((96, 160, 3))
        for _ in range(2):
            for r in range(3):
                s.render(cams[0], bands=(16, r, 3), out=bands)
            assert np.array_equal(bands, want[0])
    finally:
        hip.hipFree(buf)


def test_several_workers_level_by_level(gpu, rt, orc, levels):
    """rt_multi_* over a world rendered level by level: workers, units and stripes as for any world; the frame is the
    single-handle frame bit for bit (f64 in device memory, 8-bit, host), the counters are the counting kernel's."""
    hip = _hip()
    w, h = 320, 203
    cam, world, _, _ = load_both(rt, orc, "chapter_14", w, h)
    full, _ = world.scene().render(cam)
    _, st_full = world.scene().render(cam, stats=True)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), full.nbytes) == 0
    try:
        for devices, band_rows, units in (([0, 0], 0, 0), ([0, 0, 0], 8, 5), ([0], 24, 0)):
            multi = world.multi_scene(devices)
            multi.render_device(cam, buf.value, band_rows=band_rows, units=units)
            assert np.array_equal(_fetch(hip, buf, full.shape, np.float64), full), (devices, band_rows)
            st = multi.render_device(cam, buf.value, band_rows=band_rows, stats=True, units=units)
            assert st.as_dict() == st_full.as_dict()
            multi.render_device(cam, buf.value, band_rows=band_rows, quantized=True, units=units)
            assert np.array_equal(_fetch(hip, buf, (h, w, 3), np.uint8), orc.quantize(full))
            host, _ = multi.render(cam, band_rows=band_rows, units=units)
            assert np.array_equal(host, full)
            multi.close()
    finally:
        hip.hipFree(buf)


def test_noalloc_entry_point_wants_the_view_rendered_before(gpu, rt, orc, levels):
    """A world rendered level by level needs queues that grow with the frame and, for a new view, a wait per level:
    rt_render_device_noalloc refuses a view the handle has not rendered (RT_ERR_UNSUPPORTED, nothing launched) and
    serves it afterwards, with rt_render_device's bytes (tests/test_zz_no_alloc.py proves that it then allocates nothing)."""
    hip = _hip()
    w, h = 160, 96
    cam, world, _, _ = load_both(rt, orc, "chapter_14", w, h)
    s = world.scene()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(a), w * h * 24) == 0 and hip.hipMalloc(ctypes.byref(b), w * h * 24) == 0
    try:
        with pytest.raises(rt.RtError) as e:
            s.render_device_noalloc(cam, b.value)
        assert e.value.code == -7 and "level by level" in e.value.message
        s.render_device(cam, a.value)
        want = _fetch(hip, a, (h, w, 3), np.float64)
        s.render_device_noalloc(cam, b.value)
        assert np.array_equal(_fetch(hip, b, (h, w, 3), np.float64), want)
        with pytest.raises(rt.RtError):  # other rows are another view
            s.render_device_noalloc(cam, b.value, rows=(8, 40))
        s.render_device(cam, a.value, rows=(8, 40))
        s.render_device_noalloc(cam, b.value, rows=(8, 40))
        assert np.array_equal(_fetch(hip, b, (32, w, 3), np.float64), want[8:40])
        s.render_device(cam, a.value, quantized=True)  # (the same view: what is stored does not change the ray counts)
        s.render_device_noalloc(cam, b.value, quantized=True)
        assert np.array_equal(_fetch(hip, b, (h, w, 3), np.uint8), orc.quantize(want))
    finally:
        hip.hipFree(a)
        hip.hipFree(b)


def test_deep_trees_and_empty_levels(gpu, rt, orc, levels):
    """Budgets from 0 (one level, nothing appended) to the library's maximum of 8 (nine levels), a world nothing in which
    reflects (level 1 is empty), and a frame smaller than one block of tiles."""
    import tomli

    w, h = 72, 48
    cam, world, ocam, oworld = load_both(rt, orc, "chapter_14", w, h)
    s = world.scene()
    px = [x for y in range(h) for x in range(w)]
    py = [y for y in range(h) for x in range(w)]
    rays = np.array([list(r.origin) + list(r.direction) for r in cam.rays_for_pixels(px, py)])
    for depth in (0, 1, 8):
        img, _ = s.render(cam, max_depth=depth)
        ref = np.array([oworld.color_at(orc.ray(r[:3], r[3:]), depth) for r in rays]).reshape(h, w, 3)
        assert np.abs(img - ref).max() <= 1e-12, depth
    text = """
[camera]
width = 24
height = 17
fov = 60.0
from = [0.0, 1.5, -5.0]
to = [0.0, 1.0, 0.0]
up = [0.0, 1.0, 0.0]
[light]
position = [-10.0, 10.0, -10.0]
intensity = [1.0, 1.0, 1.0]
[[objects]]
type = "group"
[[objects.children]]
type = "sphere"
[objects.children.material]
color = [0.9, 0.2, 0.2]
[[objects.children]]
type = "cube"
[objects.children.transform]
translate = [2.0, 0.0, 1.0]
"""
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    assert world.scene().info().kernel_build == 3
    img, _ = world.scene().render(cam)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    assert_frames_match(orc, img, orc.render(ocam, oworld, threads=4))
