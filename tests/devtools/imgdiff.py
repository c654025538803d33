#!/usr/bin/env python3
"""Developer tool (GPU box): render one perf.py configuration to a .npy, or compare two such files.

    RT_AMD_LIB=... python tests/devtools/imgdiff.py render c5 /tmp/a.npy
    python tests/devtools/imgdiff.py compare /tmp/a.npy /tmp/b.npy
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    if sys.argv[1] == "render":
        import raytracer_rs_amd as rt
        from tests.devtools.perf import SIZES, scene_text

        name, out = sys.argv[2], sys.argv[3]
        sf = rt.SceneFile.from_str(scene_text(name))
        scene = sf.build_world().scene()
        w, h = SIZES[name]
        img, _ = scene.render(sf.build_camera(w, h))
        np.save(out, img)
    else:
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        d = np.abs(a - b).max(axis=-1)
        ys, xs = np.nonzero(d > 0)
        print(f"{len(ys)} pixels differ; max |d| = {d.max():.3e}")
        order = np.argsort(-d[ys, xs])[:10]
        for k in order:
            y, x = ys[k], xs[k]
            print(f"  ({x},{y}) d={d[y, x]:.3e} a={a[y, x]} b={b[y, x]}")


if __name__ == "__main__":
    main()
