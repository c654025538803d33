#!/usr/bin/env python3
"""Developer tool (GPU box): walks the ray tree of one pixel with the CPU oracle and replays every ray (radiance and
shadow) through the library's rt_intersect / rt_is_shadowed, printing where they disagree.

    python tests/devtools/trace_pixel.py c5 1301 586
"""
import os
import sys

import numpy as np
import tomli

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import raytracer_rs_amd as rt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.devtools.perf import SIZES, scene_text  # noqa: E402


def main():
    name, px, py = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    text = scene_text(name)
    w, h = SIZES[name]
    ocam, oworld = orc.scene_from_dict(tomli.loads(text), w, h)
    sf = rt.SceneFile.from_str(text)
    scene = sf.build_world().scene()
    rays, points = [], []

    def walk(r, remaining, path):
        rays.append((path, r))
        hit = oworld.hit(r)
        if hit is None:
            return
        t, oid = hit
        c = oworld.prepare_computations(oid, t, r)
        points.append((path, c.over_point.tuple()))
        m = oworld.object_material(oid)
        if remaining > 0 and m.reflectivity != 0.0:
            walk(orc.ray(c.over_point.tuple(), c.reflect_vector.tuple()), remaining - 1, path + "R")
        if remaining > 0 and m.transparency != 0.0:
            n_ratio = c.n1 / c.n2
            eye, n = np.array(c.eye_vector.tuple()), np.array(c.normal_vector.tuple())
            cos_i = float(eye @ n)
            sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i)
            if sin2_t <= 1.0:
                cos_t = np.sqrt(1.0 - sin2_t)
                d = n * (n_ratio * cos_i - cos_t) - eye * n_ratio
                walk(orc.ray(c.under_point.tuple(), tuple(d)), remaining - 1, path + "T")

    walk(orc.ray_for_pixel(ocam, px, py), 5, "P")
    arr = np.array([list(r.origin.tuple()) + list(r.direction.tuple()) for _, r in rays])
    one = [scene.intersect(arr[k:k + 1]) for k in range(len(arr))]  # one ray per launch: no other lane keeps a box alive
    gt, gobj = np.array([t[0] for t, _ in one]), np.array([o[0] for _, o in one])
    bad = 0
    for k, (path, r) in enumerate(rays):
        hit = oworld.hit(r)
        ot, oo = (hit if hit is not None else (float("nan"), -1))
        same = (oo == gobj[k]) and (oo < 0 or ot == gt[k])
        if not same:
            bad += 1
            print(f"ray {path}: oracle (t={ot!r}, obj={oo}) library (t={gt[k]!r}, obj={gobj[k]}) origin={arr[k, :3]} dir={arr[k, 3:]}")
    pts = np.array([p for _, p in points])
    gs = np.array([scene.is_shadowed(pts[k:k + 1])[0] for k in range(len(pts))])
    for k, (path, p) in enumerate(points):
        os_ = oworld.is_shadowed(p)
        if bool(gs[k]) != bool(os_):
            bad += 1
            print(f"shadow at {path}: oracle {os_} library {bool(gs[k])} point={p}")
    print(f"{len(rays)} rays, {len(points)} shadow tests, {bad} disagreements")
    oc = oworld.color_at(rays[0][1])
    gc = scene.color_at(arr[:1])[0]
    cam = sf.build_camera(w, h)
    band, _ = scene.render(cam, rows=(py - py % 8, py - py % 8 + 8))
    print("oracle color_at", oc)
    print("library color_at", tuple(gc))
    print("library render  ", tuple(band[py, px]))


if __name__ == "__main__":
    main()
