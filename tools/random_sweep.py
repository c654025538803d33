"""One-off confidence sweep (GPU box): random worlds of every kind against the CPU oracle, more seeds than the test suite
holds, as a single launch and level by level.  python tools/random_sweep.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np
import tomli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_rs_amd as rt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from raytracer_rs_amd import synthetic  # noqa: E402



first = int(sys.argv[1]) if len(sys.argv) > 1 else 200
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for seed in range(first, first + count):
    n_top = 4 + seed % 11 if seed % 3 else 20 + seed % 40
    text = synthetic.random_scene(seed, 128, 80, n_top)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    ref = orc.render(ocam, oworld, threads=16)
    for mode in ("0", "1"):
        os.environ["RT_RENDER_LEVELS"] = mode
        sf = rt.SceneFile.from_str(text)
        cam, world = sf.build_camera(), sf.build_world()
        scene = world.scene()
        img, _ = scene.render(cam)
        d = float(np.abs(img - ref).max())
        worst = max(worst, d)
        info = scene.info()
        print(f"seed {seed} n_top {n_top} leaves {info.n_leaves} nesting {info.max_nesting} build {info.kernel_build}: max|d| {d:.2e}", flush=True)
        assert d <= 1e-11, (seed, mode, d)
        world.release_scene()
for seed in range(first, first + 2 * count):
    text = synthetic.flat_scene(seed)
    ocam, oworld = orc.scene_from_dict(tomli.loads(text))
    ref = orc.render(ocam, oworld, threads=16)
    sf = rt.SceneFile.from_str(text)
    cam, world = sf.build_camera(), sf.build_world()
    scene = world.scene()
    img, _ = scene.render(cam)
    counted, _ = scene.render(cam, stats=True)
    d = float(np.abs(img - ref).max())
    worst = max(worst, d)
    info = scene.info()
    print(f"flat seed {seed} leaves {info.n_leaves} build {info.kernel_build}: max|d| {d:.2e}  plain == counting kernel: {np.array_equal(img, counted)}", flush=True)
    assert d <= 1e-11 and info.kernel_build in (0, 4) and np.array_equal(img, counted), (seed, d)
    world.release_scene()
print("worst", worst)
