"""One frame over several workers / several GPUs through the C ABI's rt_multi_* entry points, at the benchmark's
frame sizes, and bench.py's single-process several-GPU path (all tests need an MI355X).

A test box has ONE GPU: several workers are then several host threads, streams and scene handles on that GPU (device 0
listed more than once), which exercises everything but the transfer between GPUs.  The tests at the end of the file run
only where a second GPU is visible and compare a frame assembled across two GPUs, by either way of collecting it, with
the single-GPU frame bit for bit."""
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import GOLDEN, scene_path

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(GOLDEN))


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    return hip


def _device_frame(hip, buf, nbytes, dtype, shape):
    got = np.zeros(nbytes, dtype=np.uint8)
    assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), buf, nbytes, 2) == 0
    return got.view(dtype).reshape(shape)


def _plain_env():
    """The environment without a launcher's variables: bench.py must take its single-process path."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RT_BENCH_FORCE_DIST", "RT_BENCH_BACKEND"):
        env.pop(k, None)
    return env


def test_bench_drives_several_workers_without_a_launcher(gpu):
    """`python bench.py --gpus 2` started plainly: one process, rt_multi_render_device, two workers (here both on GPU 0 via
    RT_BENCH_DEVICES), the collected frame equal to the single-GPU frame (`--check`, exit status 3 otherwise)."""
    env = dict(_plain_env(), RT_BENCH_DEVICES="0,0", RT_BENCH_CALIBRATE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--check",
           "--no-cpu-baseline", "--width", "1920", "--height", "1080"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-2000:]
    assert "collected frame equals the single-GPU frame" in done.stderr
    line = json.loads(done.stdout.strip().splitlines()[-1])
    assert line["workers"] == 2 and line["n_gpus"] == 1 and line["devices"] == [0, 0]
    assert line["config"]["path"] == "single process, rt_multi_render_device"
    assert line["gather"] == "none" and line["rccl_ranks"] == 0
    assert sum(line["units_taken"]) == line["units"] >= 2 and len(line["units_taken"]) == 2
    assert line["value"] > 0 and line["config"]["rays_per_frame"] > 1920 * 1080 and line["gather_ms"] >= 0.0
    assert line["render_ms"] > 0.0 and len(line["worker_render_ms"]) == 2
    # with several GPUs both ways of collecting the frame are tried on untimed frames and the faster one is timed;
    # RT_BENCH_CALIBRATE=1 rehearses that on the one GPU there is (both handles then collect nothing)
    cal = line["gather_calibration_ms"]
    assert set(cal) == {"rccl", "copy"} and all(isinstance(v, float) and v > 0.0 for v in cal.values())


def test_bench_refuses_more_gpus_than_there_are(gpu):
    """--gpus N with fewer than N GPUs visible is an error (status 4), not a silent one-GPU run."""
    n = gpu + 1  # the fixture's value: HIP devices visible to this process
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--width", "64", "--height", "64"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=_plain_env(), cwd=ROOT)
    assert done.returncode == 4 and "GPU(s) visible" in done.stderr, (done.returncode, done.stderr[-1000:])


@pytest.mark.parametrize("name", ["chapter_11", "chapter_14"])
def test_several_workers_at_the_benchmark_frame_size(gpu, rt, orc, name):
    """3840x2160, where band count, the two-bands-in-flight ring and 199 MB of pinned staging matter: two workers (and
    three with a band height of its own) against the single-device frame, bit for bit -- in device memory as f64 and as
    8-bit values, and in host memory; counters add up."""
    hip = _hip()
    w, h = 3840, 2160
    sf = rt.SceneFile.from_file(scene_path(name))
    cam, world = sf.build_camera(w, h), sf.build_world()
    scene = world.scene()
    full, st_full = scene.render(cam, stats=True)
    plain, _ = scene.render(cam)
    assert np.abs(plain - full).max() <= 1e-13
    q_full = orc.quantize(full)
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), full.nbytes + 64) == 0
    try:
        for devices, band_rows, units in (([0, 0], 0, 0), ([0, 0, 0], 200, 7), ([0], 0, 0)):
            multi = world.multi_scene(devices)
            assert hip.hipMemset(buf, 0xA5, full.nbytes + 64) == 0
            st = multi.render_device(cam, buf.value, band_rows=band_rows, stats=True, units=units)
            got = _device_frame(hip, buf, full.nbytes + 64, np.uint8, (-1,))
            assert np.array_equal(got[:full.nbytes].view(np.float64).reshape(full.shape), full), (devices, band_rows)
            assert (got[full.nbytes:] == 0xA5).all(), "wrote past the frame"
            assert st.as_dict() == st_full.as_dict()
            info = multi.info()
            assert sum(multi.units_taken()) == info["n_units"]
            if len(devices) == 2:  # the default plan: 16-row stripes dealt to two units per worker
                assert (info["band_rows"], info["n_units"]) == (16, 4)
            if len(devices) == 1:
                assert info["n_units"] == 1
            multi.render_device(cam, buf.value, band_rows=band_rows, quantized=True, units=units)
            assert np.array_equal(_device_frame(hip, buf, w * h * 3, np.uint8, (h, w, 3)), q_full)
            # (without statistics a world with groups is rendered level by level: the same pixels to within the order
            # of a sum, so the frame to compare with is the single handle's frame rendered the same way)
            host, st = multi.render(cam, band_rows=band_rows, units=units)
            assert np.array_equal(host, plain)
            del host
            multi.close()
    finally:
        hip.hipFree(buf)


def test_collection_by_copies_rehearsed_on_one_gpu(gpu, rt, orc, monkeypatch):
    """RT_MULTI_TEST_PEERS=1 makes every worker but the first behave as if it sat on another GPU: it renders its units
    packed into a buffer of its own, copies each behind its kernel (hipMemcpyPeerAsync) into the staging memory of
    devices[0], and the placement kernel there puts the stripes into the frame.  That is the whole RT_GATHER_COPY path,
    and on a second GPU only the copy's route differs.  Frames bit-identical to the single-device frame, odd sizes
    (short last stripe, widths that leave stripes 16-, 4- and 1-byte aligned) included."""
    hip = _hip()
    monkeypatch.setenv("RT_MULTI_TEST_PEERS", "1")
    for name, (w, h), band_rows, units in (("chapter_11", (1920, 1080), 0, 0), ("chapter_14", (403, 301), 8, 5), ("chapter_11", (250, 131), 16, 3),
                                            ("chapter_6", (64, 64), 0, 0)):
        sf = rt.SceneFile.from_file(scene_path(name))
        cam, world = sf.build_camera(w, h), sf.build_world()
        full, st_full = world.scene().render(cam, stats=True)
        buf = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(buf), full.nbytes + 64) == 0
        try:
            for devices in ([0, 0], [0, 0, 0]):
                multi = world.multi_scene(devices)
                assert multi.info()["gather"] == "copy"
                for quantized in (False, True):
                    assert hip.hipMemset(buf, 0xA5, full.nbytes + 64) == 0
                    st = multi.render_device(cam, buf.value, band_rows=band_rows, stats=not quantized, quantized=quantized, units=units)
                    got = _device_frame(hip, buf, full.nbytes + 64, np.uint8, (-1,))
                    if quantized:
                        assert np.array_equal(got[:w * h * 3].reshape(h, w, 3), orc.quantize(full)), (name, devices)
                        assert (got[w * h * 3:] == 0xA5).all(), "wrote past the 8-bit frame"
                    else:
                        assert np.array_equal(got[:full.nbytes].view(np.float64).reshape(full.shape), full), (name, devices)
                        assert (got[full.nbytes:] == 0xA5).all(), "wrote past the frame"
                        assert st.as_dict() == st_full.as_dict()
                    assert sum(multi.units_taken()) == multi.info()["n_units"]
                multi.close()
        finally:
            hip.hipFree(buf)


def test_rows_rendered_in_place(gpu, rt, orc):
    """RT_RENDER_IN_PLACE: a row range, and a selection of bands, stored at their own rows of a whole frame instead of
    packed; the rest of the frame is left alone."""
    hip = _hip()
    for name, (w, h) in (("chapter_11", (403, 301)), ("chapter_14", (320, 203))):
        sf = rt.SceneFile.from_file(scene_path(name))
        cam, world = sf.build_camera(w, h), sf.build_world()
        scene = world.scene()
        full, _ = scene.render(cam)
        buf = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(buf), full.nbytes) == 0
        try:
            for quantized in (False, True):
                nbytes = w * h * 3 if quantized else full.nbytes
                want = orc.quantize(full) if quantized else full
                dtype = np.uint8 if quantized else np.float64
                assert hip.hipMemset(buf, 0, nbytes) == 0
                scene.render_device(cam, buf.value, (40, 96), None, 0, False, False, quantized, in_place=True)
                assert hip.hipDeviceSynchronize() == 0
                got = _device_frame(hip, buf, nbytes, dtype, (h, w, 3))
                assert np.array_equal(got[40:96], want[40:96]) and not got[:40].any() and not got[96:].any()
                for first in (0, 1, 2):  # three calls complete the frame
                    scene.render_device_bands(cam, buf.value, 24, first, 3, None, 0, False, False, quantized, in_place=True)
                assert hip.hipDeviceSynchronize() == 0
                assert np.array_equal(_device_frame(hip, buf, nbytes, dtype, (h, w, 3)), want)
        finally:
            hip.hipFree(buf)


def test_one_worker_costs_what_the_single_device_call_costs(gpu, rt):
    """rt_multi_render_device with devices = [0] is rt_render_device plus a wait for completion: on the benchmark frame
    it must stay within a few per cent of it (generous bound here; bench.py reports the figure as multi_single_gpu_ms)."""
    hip = _hip()
    sf = rt.SceneFile.from_file(scene_path("chapter_11"))
    cam, world = sf.build_camera(3840, 2160), sf.build_world()
    scene, multi = world.scene(), world.multi_scene([0])
    frame = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(frame), 3840 * 2160 * 3) == 0

    def single():
        scene.render_device(cam, frame.value, None, None, 0, False, False, True)
        assert hip.hipDeviceSynchronize() == 0

    def several():
        multi.render_device(cam, frame.value, quantized=True)

    try:
        for _ in range(40):
            single()
            several()
        times = {}
        for label, fn in (("single", single), ("multi", several), ("single2", single), ("multi2", several)):
            t0 = time.perf_counter()
            for _ in range(30):
                fn()
            times[label] = (time.perf_counter() - t0) / 30
        a, b = min(times["single"], times["single2"]), min(times["multi"], times["multi2"])
        assert b <= 1.10 * a, times
    finally:
        multi.close()
        hip.hipFree(frame)


# --------------------------------------------------------------------------- two GPUs (skipped on a one-GPU box)
@pytest.mark.parametrize("gather", ["rccl", "copy"])
def test_frame_assembled_across_two_gpus(gpu, rt, orc, gather, monkeypatch):
    """Bands rendered on GPU 1 travel to the frame on GPU 0 -- with RCCL (grouped ncclSend / ncclRecv) and with peer
    copies -- and the frame is the single-GPU frame bit for bit.  Needs two GPUs."""
    if gpu < 2:
        pytest.skip("needs a second GPU")
    hip = _hip()
    monkeypatch.setenv("RT_MULTI_GATHER", gather)
    for name, (w, h) in (("chapter_11", (1920, 1080)), ("chapter_14", (1280, 720))):
        sf = rt.SceneFile.from_file(scene_path(name))
        cam, world = sf.build_camera(w, h), sf.build_world()
        full, st_full = world.scene(0).render(cam, stats=True)
        buf = ctypes.c_void_p()
        assert hip.hipSetDevice(0) == 0
        assert hip.hipMalloc(ctypes.byref(buf), full.nbytes) == 0
        try:
            for devices, band_rows in (([0, 1], 0), ([0, 1, 1], 64), ([1, 0], 128)):  # band_rows: the stripe height
                multi = world.multi_scene(devices)
                info = multi.info()
                assert info["gather"] == gather and info["n_devices"] == 2 and info["rccl_ranks"] == (2 if gather == "rccl" else 0)
                assert hip.hipSetDevice(devices[0]) == 0
                frame = ctypes.c_void_p()
                assert hip.hipMalloc(ctypes.byref(frame), full.nbytes) == 0
                for quantized in (False, True):
                    assert hip.hipMemset(frame, 0xA5, full.nbytes) == 0
                    st = multi.render_device(cam, frame.value, band_rows=band_rows, stats=not quantized, quantized=quantized)
                    if quantized:
                        assert np.array_equal(_device_frame(hip, frame, w * h * 3, np.uint8, (h, w, 3)), orc.quantize(full))
                    else:
                        assert np.array_equal(_device_frame(hip, frame, full.nbytes, np.float64, full.shape), full), (name, devices)
                        assert st.as_dict() == st_full.as_dict()
                    assert sum(multi.units_taken()) == multi.info()["n_units"]
                host, _ = multi.render(cam, band_rows=band_rows)
                assert np.array_equal(host, full)
                hip.hipFree(frame)
                multi.close()
        finally:
            hip.hipSetDevice(0)
            hip.hipFree(buf)
