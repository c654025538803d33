#!/usr/bin/env python3
"""Developer timing loop (GPU box): kernel milliseconds of the render kernel on the benchmark configurations.

    python tests/devtools/perf.py [--reps 5] [--configs c3,c4,c5]

c2 = chapter_7_and_8 at 1920x1080, c3 = chapter_11 at 3840x2160 (the headline), c4 = chapter_14 at 3840x2160,
c5 = the generated 10 000-leaf scene at 1920x1080.  Each config is also rendered at a reduced size and compared with
the CPU oracle (colours and ray count), so a fast wrong kernel shows immediately.
"""
import argparse
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import raytracer_rs_amd as rt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from raytracer_rs_amd import synthetic  # noqa: E402

HIP = ctypes.CDLL("libamdhip64.so")
HIP.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
HIP.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
HIP.hipFree.argtypes = [ctypes.c_void_p]

# algorithmic flops per operation of the reference, as bench.py prices them (add/sub/mul/div/sqrt/pow = 1 each)
FLOPS = {
    "ray_gen": 36, "ray_transform": 33, "aabb": 12,
    "prim_miss": (21, 0, 12, 44, 52), "prim_hit": (27, 1, 12, 44, 52),
    "hit_record": 79, "shadow_setup": 18, "lighting_unlit": 29, "lighting_lit": 61, "pattern": 40,
    "reflect": 3, "refract_setup": 26, "schlick": 16, "combine": 6, "combine_schlick_extra": 7,
}


def algorithmic_flops(s):
    f = FLOPS["ray_gen"] * s["primary"] + FLOPS["ray_transform"] * s["ray_transforms"] + FLOPS["aabb"] * s["aabb_tests"]
    for k in range(5):
        f += FLOPS["prim_hit"][k] * s["prim_hits"][k] + FLOPS["prim_miss"][k] * (s["prim_tests"][k] - s["prim_hits"][k])
    f += (FLOPS["hit_record"] + FLOPS["combine"]) * s["shades"] + FLOPS["shadow_setup"] * s["shadow"]
    f += FLOPS["lighting_lit"] * s["lit"] + FLOPS["lighting_unlit"] * (s["shades"] - s["lit"])
    f += FLOPS["pattern"] * s["patterns"] + FLOPS["reflect"] * s["reflect"] + FLOPS["refract_setup"] * s["refract_setups"]
    f += (FLOPS["schlick"] + FLOPS["combine_schlick_extra"]) * s["schlick"]
    return float(f)


def scene_text(name):
    if name == "c5":
        return synthetic.large_scene()
    fn = {"c2": "chapter_7_and_8", "c3": "chapter_11", "c4": "chapter_14"}[name]
    with open(os.path.join(ROOT, "tests", "golden", "scenes", fn + ".toml")) as f:
        return f.read()


SIZES = {"c2": (1920, 1080), "c3": (3840, 2160), "c4": (3840, 2160), "c5": (1920, 1080)}
CHECK = {"c2": (192, 108), "c3": (384, 216), "c4": (384, 216), "c5": (192, 108)}


def main():
    import tomli

    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--configs", default="c2,c3,c4,c5")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--size", default=None, help="WxH for every configuration instead of its own size")
    a = ap.parse_args()
    for name in a.configs.split(","):
        text = scene_text(name)
        sf = rt.SceneFile.from_str(text)
        world = sf.build_world()
        scene = world.scene()
        status = "unchecked"
        if not a.no_check:
            w, h = CHECK[name]
            cam = sf.build_camera(w, h)
            counted, st = scene.render(cam, stats=True)  # the counting kernel visits every node
            img, _ = scene.render(cam)                    # the timed kernel steps over runs of nodes
            ocam, oworld = orc.scene_from_dict(tomli.loads(text), w, h)
            oc = orc.Counters()
            ref = orc.render(ocam, oworld, threads=16, counters=oc)
            err = float(np.abs(img - ref).max())
            status = (f"check {w}x{h}: max|d|={err:.1e} counters={'ok' if st.as_dict() == oc.as_dict() else 'DIFFER'} "
                      f"kernels={'agree' if np.array_equal(img, counted) else 'DISAGREE'}")
        w, h = SIZES[name] if a.size is None else tuple(int(v) for v in a.size.split("x"))
        cam = sf.build_camera(w, h)
        # whole frames into device memory, one launch each (rt_render would cut the frame into groups of rows to
        # overlap the transfer to the host): per launch = per frame, for the profiler too
        buf = ctypes.c_void_p()
        assert HIP.hipMalloc(ctypes.byref(buf), w * h * 3) == 0
        st = scene.render_device(cam, buf.value, stats=True, quantized=True)
        times = []
        for _ in range(a.reps):
            scene.render_device(cam, buf.value, timed=True, quantized=True)
            assert HIP.hipDeviceSynchronize() == 0
            times.append(sum(scene.render_times()))
        img = np.zeros(w * h * 3, dtype=np.uint8)
        assert HIP.hipMemcpy(img.ctypes.data_as(ctypes.c_void_p), buf, img.nbytes, 2) == 0
        HIP.hipFree(buf)
        digest = hashlib.sha256(img.tobytes()).hexdigest()[:12]
        best = min(times)
        flops = algorithmic_flops(st.as_dict())
        print(f"{name} {w}x{h}: kernel min {best:.3f} ms  median {sorted(times)[len(times) // 2]:.3f} ms  rays {st.rays}  "
              f"{st.rays / best / 1e3:.0f} Mrays/s  flops {flops:.6g}  sha8 {digest}  {status}", flush=True)


if __name__ == "__main__":
    main()
