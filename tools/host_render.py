import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import raytracer_rs_amd as rt
for name in ("chapter_11", "chapter_14"):
    sf = rt.SceneFile.from_file(f"tests/golden/scenes/{name}.toml")
    world = sf.build_world(); scene = world.scene(); cam = sf.build_camera(3840, 2160)
    out = np.zeros((2160, 3840, 3))
    for _ in range(3): scene.render(cam, out=out)
    t = []
    for _ in range(8):
        t0 = time.perf_counter(); scene.render(cam, out=out); t.append((time.perf_counter() - t0) * 1e3)
    print(name, "rt_render 4K host frame ms: min %.2f median %.2f" % (min(t), sorted(t)[len(t)//2]), flush=True)
