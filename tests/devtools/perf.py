#!/usr/bin/env python3
"""Developer timing loop (GPU box): kernel milliseconds of the render kernel on the benchmark configurations.

    python tests/devtools/perf.py [--reps 5] [--configs c3,c4,c5]

c2 = chapter_7_and_8 at 1920x1080, c3 = chapter_11 at 3840x2160 (the headline), c4 = chapter_14 at 3840x2160,
c5 = the generated 10 000-leaf scene at 1920x1080.  Each config is also rendered at a reduced size and compared with
the CPU oracle (colours and ray count), so a fast wrong kernel shows immediately.
"""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import raytracer_rs_amd as rt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from raytracer_rs_amd import synthetic  # noqa: E402


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
        img, st = scene.render(cam, stats=True)
        times = []
        for _ in range(a.reps):
            img, s2 = scene.render(cam)
            times.append(s2.kernel_ms)
        digest = hashlib.sha256(img.tobytes()).hexdigest()[:12]
        best = min(times)
        print(f"{name} {w}x{h}: kernel min {best:.3f} ms  median {sorted(times)[len(times) // 2]:.3f} ms  rays {st.rays}  "
              f"{st.rays / best / 1e3:.0f} Mrays/s  sha {digest}  {status}", flush=True)


if __name__ == "__main__":
    main()
