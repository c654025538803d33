import os, sys
sys.path.insert(0, os.getcwd())
import raytracer_rs_amd as rt
from raytracer_rs_amd import synthetic
for name, src, size in (("c4", "file", (3840, 2160)), ("c5", "gen", (1920, 1080)), ("c3", "file11", (3840, 2160))):
    if src == "file": sf = rt.SceneFile.from_file("tests/golden/scenes/chapter_14.toml")
    elif src == "file11": sf = rt.SceneFile.from_file("tests/golden/scenes/chapter_11.toml")
    else: sf = rt.SceneFile.from_str(synthetic.large_scene(*size))
    world = sf.build_world(); scene = world.scene(); cam = sf.build_camera(*size)
    img, st = scene.render(cam, stats=True)
    d = st.as_dict()
    print(name, {k: d[k] for k in ("rays", "primary", "shadow", "reflect", "refract", "shades", "lit", "refract_setups", "schlick")}, flush=True)
