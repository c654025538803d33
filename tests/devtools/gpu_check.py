#!/usr/bin/env python3
"""Developer check (GPU box): renders every fixture scene with the HIP library and compares it with the CPU oracle.

    python tests/devtools/gpu_check.py [scene ...] [--scale S]

Prints, per scene: max |GPU - oracle| over all colour values, the number of 8-bit pixels that differ, whether the
operation counters agree, and the kernel time.  Exit status 1 if any scene breaks the parity bar
(1e-6 on f64 colours; +-1 LSB on >= 99.9 % of pixels).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import raytracer_rs_amd as rt  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SCENES = ["chapter_5", "chapter_6", "chapter_7_and_8", "chapter_9", "chapter_10", "chapter_11", "chapter_12",
          "chapter_13", "chapter_14"]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=SCENES)
    ap.add_argument("--scale", type=float, default=1.0, help="scale the scene's native resolution")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    bad = 0
    for name in args.scenes:
        path = os.path.join(ROOT, "tests", "golden", "scenes", name + ".toml")
        sf = rt.SceneFile.from_file(path)
        cam0 = sf.build_camera()
        w, h = max(1, int(cam0.hsize * args.scale)), max(1, int(cam0.vsize * args.scale))
        cam = sf.build_camera(w, h)
        world = sf.build_world()
        scene = world.scene()
        img, st = scene.render(cam, stats=True)
        t0 = time.time()
        img2, st2 = scene.render(cam)
        wall = time.time() - t0
        assert np.array_equal(img, img2), "counting and plain kernels disagree"
        ocam, oworld = orc.load_scene(path, w, h)
        oc = orc.Counters()
        t0 = time.time()
        ref = orc.render(ocam, oworld, threads=args.threads, counters=oc)
        cpu = time.time() - t0
        diff = np.abs(img - ref)
        q_gpu, q_ref = orc.quantize(img), orc.quantize(ref)
        qd = np.abs(q_gpu.astype(int) - q_ref.astype(int)).max(axis=2)
        counters_ok = st.as_dict() == {k: v for k, v in oc.as_dict().items()}
        frac_ok = float((qd <= 1).mean())
        ok = diff.max() <= 1e-6 and frac_ok >= 0.999 and counters_ok
        bad += 0 if ok else 1
        print(f"{name:16s} {w}x{h} max|d|={diff.max():.3e} px_differ={int((qd > 0).sum())} "
              f"px_off_by>1={int((qd > 1).sum())} counters={'ok' if counters_ok else 'DIFFER'} "
              f"rays={st.rays} kernel={st2.kernel_ms:.3f}ms ({st.rays / max(st2.kernel_ms, 1e-9) / 1e3:.1f} Mrays/s) "
              f"wall={wall * 1e3:.1f}ms oracle[{args.threads}thr]={cpu * 1e3:.0f}ms {'OK' if ok else 'FAIL'}", flush=True)
        if not counters_ok:
            print("  gpu   :", st.as_dict())
            print("  oracle:", oc.as_dict())
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
