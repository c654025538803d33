"""Pins the CPU oracle against the images the reference itself rendered (tests/golden/, see make_golden.py).

Bit-level: the five scenes whose P3 files are in the reference checkout -- every 8-bit value of every pixel, and the
SHA-256 of the regenerated P3 text against the SHA-256 of the reference's own file.
The four scenes the reference checkout only holds as JPEG (quality 100): loosely, pixel by pixel after decoding (a
correct reading of groups, bounds, material inheritance, cubes and planes lands within JPEG noise of them, a wrong one
does not) -- and exactly: the oracle's frame, through the restated encoder (oracle/jpeg.py, itself pinned byte for byte
by the five scenes that have both files), gives the reference's JPEG FILE byte for byte, which it can only do when the
8-bit pixels are the reference's in all but freak cases.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path

PPM_PAIRS = [("chapter_5", "chapter_5_circle"), ("chapter_6", "chapter_6_sphere"), ("chapter_10", "chapter_10_patterns"),
             ("chapter_11", "chapter_11_reflections"), ("chapter_13", "chapter_13_rocket")]
JPG_PAIRS = [("chapter_7_and_8", "chapter_7_and_8_world"), ("chapter_9", "chapter_9_plane_scene"),
             ("chapter_12", "chapter_12_cubes"), ("chapter_14", "chapter_14_groups")]


def manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("scene,image", PPM_PAIRS)
def test_oracle_reproduces_reference_ppm(orc, scene, image):
    cam, world = orc.load_scene(scene_path(scene))
    frame = orc.render(cam, world, threads=8)
    golden = np.load(os.path.join(GOLDEN, image + ".npz"))["pixels"]
    info = manifest()["ppm"][image]
    assert hashlib.sha256(golden.tobytes()).hexdigest() == info["sha256_pixels"]
    assert frame.shape == golden.shape
    assert np.array_equal(orc.quantize(frame), golden)
    text = orc.to_ppm(frame)
    assert len(text) == info["p3_bytes"]
    assert hashlib.sha256(text).hexdigest() == info["sha256_p3_text"]
    assert max(len(line) for line in text.split(b"\n")) <= 69 and text.endswith(b"\n\n")


@pytest.mark.parametrize("scene,image", JPG_PAIRS)
def test_oracle_close_to_reference_jpeg(orc, scene, image):
    from PIL import Image

    cam, world = orc.load_scene(scene_path(scene))
    frame = orc.quantize(orc.render(cam, world, threads=8)).astype(np.int32)
    jpg = np.asarray(Image.open(os.path.join(GOLDEN, image + ".jpg")).convert("RGB")).astype(np.int32)
    assert jpg.shape == frame.shape
    d = np.abs(frame - jpg).max(axis=2)
    assert (d <= 6).mean() >= 0.99, f"only {(d <= 6).mean():.4f} of pixels within 6 levels of the JPEG"
    assert (d <= 16).mean() >= 0.9995, f"{int((d > 16).sum())} pixels beyond 16 levels"


@pytest.mark.parametrize("scene,image", JPG_PAIRS)
def test_oracle_reproduces_reference_jpeg_file(orc, scene, image):
    from oracle import jpeg

    cam, world = orc.load_scene(scene_path(scene))
    pixels = orc.quantize(orc.render(cam, world, threads=8))
    with open(os.path.join(GOLDEN, image + ".jpg"), "rb") as f:
        theirs = f.read()
    assert hashlib.sha256(theirs).hexdigest() == manifest()["jpg"][image]["sha256"]
    assert jpeg.jpeg_bytes(pixels) == theirs


def test_ray_counts_of_the_native_frames(orc):
    """Workload facts the benchmark relies on: rays per pixel of the reference's scenes at their native sizes."""
    expect = {"chapter_6": (1.0, 1.3), "chapter_7_and_8": (1.99, 2.01), "chapter_11": (12.0, 15.0), "chapter_14": (6.0, 8.5)}
    for scene, (lo, hi) in expect.items():
        cam, world = orc.load_scene(scene_path(scene))
        c = orc.Counters()
        orc.render(cam, world, threads=8, counters=c)
        rpp = c.rays / (cam.hsize * cam.vsize)
        assert lo <= rpp <= hi, (scene, rpp)
        assert c.rays == c.primary + c.shadow + c.reflect + c.refract
