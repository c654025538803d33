"""Where level-by-level rendering starts to pay: generated hierarchies of growing size (the benchmark scene's
structure: cells x cells groups of 2 x 2 sub-groups of 25 leaves, same density), each at 3840x2160 as a single launch
(RT_RENDER_LEVELS=0) and level by level (=1); GPU time per frame, best of five, 8-bit frame in device memory."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_rs_amd as rt  # noqa: E402
from raytracer_rs_amd import synthetic  # noqa: E402

HIP = ctypes.CDLL("libamdhip64.so")
HIP.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]

w, h = 3840, 2160
buf = ctypes.c_void_p()
assert HIP.hipMalloc(ctypes.byref(buf), w * h * 3) == 0
for cells in (1, 2, 3, 4, 5, 7, 10):
    text = synthetic.large_scene(w, h, cells=cells, extent=2.0 * cells)
    row = []
    for mode in ("0", "1"):
        os.environ["RT_RENDER_LEVELS"] = mode
        sf = rt.SceneFile.from_str(text)
        world = sf.build_world()
        scene = world.scene()
        cam = sf.build_camera(w, h)
        times = []
        for _ in range(6):
            scene.render_device(cam, buf.value, timed=True, quantized=True)
            assert HIP.hipDeviceSynchronize() == 0
            times.append(sum(scene.render_times()))
        row.append(min(times[1:]))
        leaves = scene.info().n_leaves
        world.release_scene()
    print(f"cells {cells:2d}  leaves {leaves:6d}  single launch {row[0]:8.3f} ms  levels {row[1]:8.3f} ms  ratio {row[1] / row[0]:.3f}", flush=True)
