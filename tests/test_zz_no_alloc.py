"""rt_render_device_noalloc: the entry point that allocates nothing (its tile-ordering scratch is a static device
array, one copy per GPU) must give the very bytes rt_render_device gives, call after call, whatever the previous call
left in the shared array."""
import ctypes

import numpy as np
import pytest

from test_gpu_parity import load_both

pytestmark = pytest.mark.gpu


class Hip:
    def __init__(self):
        self.lib = ctypes.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        self.lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        self.lib.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        self.lib.hipFree.argtypes = [ctypes.c_void_p]

    def alloc(self, n):
        p = ctypes.c_void_p()
        assert self.lib.hipMalloc(ctypes.byref(p), n) == 0
        return p

    def fetch(self, p, shape, dtype):
        out = np.zeros(shape, dtype=dtype)
        assert self.lib.hipDeviceSynchronize() == 0
        assert self.lib.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), p, out.nbytes, 2) == 0
        return out


def test_noalloc_entry_point_matches_render_device(gpu, rt, orc):
    hip = Hip()
    cap = 1024 * 640 * 3 * 8 + 4096
    a, b = hip.alloc(cap), hip.alloc(cap)
    try:
        # two calls in a row on each scene, the second one smaller: the static array still holds the tile order of the
        # larger frame, and the smaller frame must not read any of it
        for name, sizes in (("chapter_11", ((1024, 640), (131, 67))), ("chapter_14", ((403, 277), (96, 40))),
                            ("chapter_6", ((200, 120), (64, 64)))):
            for (w, h) in sizes:
                cam, world, _, _ = load_both(rt, orc, name, w, h)
                scene = world.scene()
                for quantized, dtype, nbytes in ((False, np.float64, w * h * 24), (True, np.uint8, w * h * 3)):
                    assert hip.lib.hipMemset(a, 0xA5, cap) == 0 and hip.lib.hipMemset(b, 0xA5, cap) == 0
                    scene.render_device(cam, a.value, quantized=quantized)
                    scene.render_device_noalloc(cam, b.value, quantized=quantized)
                    want = hip.fetch(a, nbytes + 64, np.uint8)
                    got = hip.fetch(b, nbytes + 64, np.uint8)
                    assert np.array_equal(got, want), (name, w, h, quantized)
                    assert (got[nbytes:] == 0xA5).all(), "wrote past the rows rendered"
                # rows and bands land packed, as with rt_render_device
                scene.render_device(cam, a.value, rows=(8, 31))
                scene.render_device_noalloc(cam, b.value, rows=(8, 31))
                assert np.array_equal(hip.fetch(b, (23, w, 3), np.float64), hip.fetch(a, (23, w, 3), np.float64))
                scene.render_device_bands(cam, a.value, 16, 1, 3)
                scene.render_device_noalloc(cam, b.value, bands=(16, 1, 3))
                n_rows = sum(min(h, (k + 1) * 16) - k * 16 for k in range(1, (h + 15) // 16, 3))
                assert np.array_equal(hip.fetch(b, (n_rows, w, 3), np.float64), hip.fetch(a, (n_rows, w, 3), np.float64))
        # the whole frame against the oracle once, so that "equal to rt_render_device" is not the only anchor
        cam, world, ocam, oworld = load_both(rt, orc, "chapter_11", 160, 120)
        world.scene().render_device_noalloc(cam, b.value)
        got = hip.fetch(b, (120, 160, 3), np.float64)
        ref = orc.render(ocam, oworld, threads=8)
        assert np.abs(got - ref).max() <= 1e-6  # f64 colours within 1e-6 of the reference's arithmetic
        assert np.array_equal(orc.quantize(got), orc.quantize(ref))
    finally:
        hip.lib.hipFree(a)
        hip.lib.hipFree(b)


def test_noalloc_entry_point_refuses_what_it_cannot_do(gpu, rt, orc):
    from raytracer_rs_amd import _lib as L

    cam, world, _, _ = load_both(rt, orc, "chapter_11", 8192, 8192)  # 1 048 576 tiles: beyond the static array
    scene = world.scene()
    with pytest.raises(rt.RtError) as e:
        scene.render_device_noalloc(cam, 4096)  # refused before anything is launched: the pointer is never used
    assert e.value.code == -9 and "tiles" in e.value.message
    # a row range of the same frame that fits is accepted and agrees
    hip = Hip()
    a, b = hip.alloc(8192 * 16 * 24), hip.alloc(8192 * 16 * 24)
    try:
        scene.render_device(cam, a.value, rows=(4000, 4016))
        scene.render_device_noalloc(cam, b.value, rows=(4000, 4016))
        assert np.array_equal(hip.fetch(b, (16, 8192, 3), np.float64), hip.fetch(a, (16, 8192, 3), np.float64))
    finally:
        hip.lib.hipFree(a)
        hip.lib.hipFree(b)
    cd, opts = cam._desc(), L.RenderOpts()
    opts.collect_stats = 1
    st = L.Stats()
    assert L.lib().rt_render_device_noalloc(scene._h, ctypes.byref(cd), ctypes.byref(opts), ctypes.c_void_p(4096)) == -1
    opts.collect_stats, opts.flags = 0, L.RENDER_TIMED
    assert L.lib().rt_render_device_noalloc(scene._h, ctypes.byref(cd), ctypes.byref(opts), ctypes.c_void_p(4096)) == -1


def test_noalloc_entry_point_allocates_nothing(gpu, rt, orc):
    """The property the entry point exists for.  (1) hipMemGetInfo reports the same free memory before and after a run
    of calls on frames of growing size (rt_render_device would have grown its scratch).  (2) The call can be captured
    into a HIP graph in GLOBAL capture mode -- where an allocation, a synchronisation or an event creation by this
    thread invalidates the capture -- and the replayed graph renders the frame rt_render_device renders."""
    hip = Hip()
    lib = hip.lib
    lib.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.hipStreamBeginCapture.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.hipGraphInstantiate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    lib.hipGraphExecDestroy.argtypes = [ctypes.c_void_p]
    lib.hipGraphDestroy.argtypes = [ctypes.c_void_p]
    lib.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    w, h = 640, 360
    a, b = hip.alloc(w * h * 24), hip.alloc(w * h * 24)
    stream = ctypes.c_void_p()
    assert lib.hipStreamCreate(ctypes.byref(stream)) == 0
    try:
        for name in ("chapter_11", "chapter_14"):
            cam, world, _, _ = load_both(rt, orc, name, w, h)
            scene = world.scene()
            scene.render_device(cam, a.value)
            want = hip.fetch(a, (h, w, 3), np.float64)
            # first use on this stream done: code objects loaded, the array's address resolved, and the runtime has
            # given the stream's hardware queue its scratch memory for kernels with a per-lane stack (its doing, on a
            # queue's first such launch, not the entry point's)
            if scene.info().kernel_build == 3:
                # a world rendered level by level: the entry point serves views the handle has rendered before (their
                # queues exist and their ray counts are known); the largest view last, so that nothing grows afterwards
                for rows in ((0, 16), (0, 120), (0, h)):
                    scene.render_device(cam, a.value, rows=rows, stream=stream.value)
            scene.render_device_noalloc(cam, b.value, stream=stream.value)
            assert lib.hipDeviceSynchronize() == 0
            free0, total = ctypes.c_size_t(), ctypes.c_size_t()
            assert lib.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
            for rows in ((0, 16), (0, 120), (0, h)):
                scene.render_device_noalloc(cam, b.value, rows=rows, stream=stream.value)
            assert lib.hipStreamSynchronize(stream) == 0
            free1 = ctypes.c_size_t()
            assert lib.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
            assert free1.value == free0.value, "device memory was allocated during rt_render_device_noalloc"
            assert np.array_equal(hip.fetch(b, (h, w, 3), np.float64), want)
            # the same call under stream capture, replayed twice
            assert lib.hipMemset(b, 0, w * h * 24) == 0 and lib.hipDeviceSynchronize() == 0
            graph, graph_exec = ctypes.c_void_p(), ctypes.c_void_p()
            assert lib.hipStreamBeginCapture(stream, 0) == 0  # hipStreamCaptureModeGlobal
            try:
                scene.render_device_noalloc(cam, b.value, stream=stream.value)
            finally:
                rc = lib.hipStreamEndCapture(stream, ctypes.byref(graph))
            assert rc == 0 and graph.value, f"the capture was invalidated (hipStreamEndCapture: {rc})"
            assert lib.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, 0) == 0
            for _ in range(2):
                assert lib.hipMemset(b, 0, w * h * 24) == 0 and lib.hipDeviceSynchronize() == 0
                assert lib.hipGraphLaunch(graph_exec, stream) == 0
                assert lib.hipStreamSynchronize(stream) == 0
                assert np.array_equal(hip.fetch(b, (h, w, 3), np.float64), want)
            lib.hipGraphExecDestroy(graph_exec)
            lib.hipGraphDestroy(graph)
    finally:
        lib.hipStreamDestroy(stream)
        lib.hipFree(a)
        lib.hipFree(b)


def test_noalloc_entry_point_allocates_nothing_level_by_level(gpu, rt, orc, monkeypatch):
    """The same property for a world rendered level by level of its ray trees (every world with groups): once rt_render_device has rendered a view -- its ray counts are then
    remembered and its queues exist -- rt_render_device_noalloc renders that view without allocating (hipMemGetInfo) and
    without waiting for the GPU (it can be captured into a HIP graph in global capture mode), with the same bytes."""
    monkeypatch.setenv("RT_RENDER_LEVELS", "1")
    hip = Hip()
    lib = hip.lib
    lib.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    lib.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.hipStreamBeginCapture.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.hipGraphInstantiate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    lib.hipGraphExecDestroy.argtypes = [ctypes.c_void_p]
    lib.hipGraphDestroy.argtypes = [ctypes.c_void_p]
    lib.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    w, h = 640, 360
    a, b = hip.alloc(w * h * 24), hip.alloc(w * h * 24)
    stream = ctypes.c_void_p()
    assert lib.hipStreamCreate(ctypes.byref(stream)) == 0
    try:
        cam, world, _, _ = load_both(rt, orc, "chapter_14", w, h)
        scene = world.scene()
        assert scene.info().kernel_build == 3
        scene.render_device(cam, a.value, stream=stream.value)  # the view's first frame: counts learnt, queues made
        want = hip.fetch(a, (h, w, 3), np.float64)
        scene.render_device_noalloc(cam, b.value, stream=stream.value)  # (first use of the entry point on this stream)
        assert lib.hipDeviceSynchronize() == 0
        free0, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
        for _ in range(3):
            scene.render_device_noalloc(cam, b.value, stream=stream.value)
        assert lib.hipStreamSynchronize(stream) == 0
        free1 = ctypes.c_size_t()
        assert lib.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
        assert free1.value == free0.value, "device memory was allocated during rt_render_device_noalloc"
        assert np.array_equal(hip.fetch(b, (h, w, 3), np.float64), want)
        assert lib.hipMemset(b, 0, w * h * 24) == 0 and lib.hipDeviceSynchronize() == 0
        graph, graph_exec = ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.hipStreamBeginCapture(stream, 0) == 0  # hipStreamCaptureModeGlobal
        try:
            scene.render_device_noalloc(cam, b.value, stream=stream.value)
        finally:
            rc = lib.hipStreamEndCapture(stream, ctypes.byref(graph))
        assert rc == 0 and graph.value, f"the capture was invalidated (hipStreamEndCapture: {rc})"
        assert lib.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, 0) == 0
        for k in range(3):
            assert lib.hipMemset(b, 0, w * h * 24) == 0 and lib.hipDeviceSynchronize() == 0
            assert lib.hipGraphLaunch(graph_exec, stream) == 0
            assert lib.hipStreamSynchronize(stream) == 0
            assert np.array_equal(hip.fetch(b, (h, w, 3), np.float64), want)
            # Another (smaller) view through the same handle between two replays: the handle's numbering of rays is
            # then that view's, and the graph must not depend on it (it numbers its own rays).
            scene.render_device(cam, a.value, rows=(8, 120), stream=stream.value)
            scene.render_device_noalloc(cam, a.value, rows=(8, 120), stream=stream.value)
            assert np.array_equal(hip.fetch(a, (112, w, 3), np.float64), want[8:120]), k
        lib.hipGraphExecDestroy(graph_exec)
        lib.hipGraphDestroy(graph)
    finally:
        lib.hipStreamDestroy(stream)
        lib.hipFree(a)
        lib.hipFree(b)

