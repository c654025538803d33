"""The book's worlds with groups as a single launch (RT_RENDER_LEVELS=0) and level by level (=1), at three frame
sizes; GPU time per frame of a repeated view, best of eight, 8-bit frame in device memory."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_rs_amd as rt  # noqa: E402

HIP = ctypes.CDLL("libamdhip64.so")
HIP.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
buf = ctypes.c_void_p()
assert HIP.hipMalloc(ctypes.byref(buf), 3840 * 2160 * 3) == 0
for name in ("chapter_12", "chapter_13", "chapter_14"):
    path = os.path.join(ROOT, "tests", "golden", "scenes", name + ".toml")
    for w, h in ((320, 200), (1280, 720), (3840, 2160)):
        row = []
        for mode in ("0", "1"):
            os.environ["RT_RENDER_LEVELS"] = mode
            sf = rt.SceneFile.from_file(path)
            world = sf.build_world()
            scene = world.scene()
            info = scene.info()
            cam = sf.build_camera(w, h)
            times = []
            for _ in range(10):
                scene.render_device(cam, buf.value, timed=True, quantized=True)
                assert HIP.hipDeviceSynchronize() == 0
                times.append(sum(scene.render_times()))
            row.append(min(times[2:]))
            world.release_scene()
        print(f"{name} ({info.n_leaves} leaves, nesting {info.max_nesting}) {w}x{h}: single launch {row[0]:7.3f} ms  levels {row[1]:7.3f} ms  ratio {row[1] / row[0]:.3f}", flush=True)
