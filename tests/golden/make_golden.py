#!/usr/bin/env python3
"""Builds the committed fixtures under tests/golden/ from the reference checkout's DATA files.

Run once in the development container (the reference checkout does not exist on the GPU box):

    python tests/golden/make_golden.py [/root/reference]

What it takes (data only, never source):
  * outputs/chapter_{5_circle,6_sphere,10_patterns,11_reflections,13_rocket}.ppm -- the images the reference
    itself rendered (P3 text).  Stored as the decoded 8-bit pixel array in a compressed .npz (the P3 text is
    3.6 MB per image; the pixel payload compresses to a few hundred KB), with the SHA-256 of the raw payload
    and the SHA-256 of the original P3 text recorded in MANIFEST.json.
  * outputs/chapter_{7_and_8_world,9_plane_scene,12_cubes,14_groups}.jpg -- the only reference renderings of those
    scenes (their PPMs are not in the checkout).  JPEG quality 100: a loose oracle, copied byte for byte.
  * outputs/chapter_{5_circle,6_sphere,10_patterns,11_reflections,13_rocket}.jpg -- the JPEG the reference wrote of
    the same canvas as each of those PPMs (Canvas::to_jpeg, quality 100): together with the PPM's pixels they pin the
    JPEG writer bit for bit.  Copied byte for byte.
  * scenes/*.toml -- the scene inputs, copied byte for byte.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PPMS = ["chapter_5_circle", "chapter_6_sphere", "chapter_10_patterns", "chapter_11_reflections", "chapter_13_rocket"]
JPGS = ["chapter_7_and_8_world", "chapter_9_plane_scene", "chapter_12_cubes", "chapter_14_groups"]


def parse_p3(text: bytes):
    toks = text.split()
    assert toks[0] == b"P3" and toks[3] == b"255"
    w, h = int(toks[1]), int(toks[2])
    vals = np.array(toks[4:], dtype=np.int64)
    assert vals.size == w * h * 3, (vals.size, w, h)
    assert vals.min() >= 0 and vals.max() <= 255
    return vals.astype(np.uint8).reshape(h, w, 3)


def main() -> None:
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    manifest = {"ppm": {}, "jpg": {}, "jpg_of_ppm": {}, "scenes": {}}
    for name in PPMS:
        with open(os.path.join(ref, "outputs", name + ".ppm"), "rb") as f:
            text = f.read()
        px = parse_p3(text)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pixels=px)
        lines = text.split(b"\n")
        manifest["ppm"][name] = {
            "width": int(px.shape[1]), "height": int(px.shape[0]),
            "sha256_pixels": hashlib.sha256(px.tobytes()).hexdigest(),
            "sha256_p3_text": hashlib.sha256(text).hexdigest(),
            "p3_bytes": len(text),
            "p3_max_line_len": max(len(l) for l in lines),
            "p3_ends_with_blank_line": text.endswith(b"\n\n"),
        }
    for name in JPGS:
        src = os.path.join(ref, "outputs", name + ".jpg")
        shutil.copyfile(src, os.path.join(HERE, name + ".jpg"))
        with open(src, "rb") as f:
            manifest["jpg"][name] = {"sha256": hashlib.sha256(f.read()).hexdigest()}
    for name in PPMS:
        src = os.path.join(ref, "outputs", name + ".jpg")
        shutil.copyfile(src, os.path.join(HERE, name + ".jpg"))
        with open(src, "rb") as f:
            manifest["jpg_of_ppm"][name] = {"sha256": hashlib.sha256(f.read()).hexdigest()}
    os.makedirs(os.path.join(HERE, "scenes"), exist_ok=True)
    for fn in sorted(os.listdir(os.path.join(ref, "scenes"))):
        if fn.endswith(".toml"):
            src = os.path.join(ref, "scenes", fn)
            shutil.copyfile(src, os.path.join(HERE, "scenes", fn))
            with open(src, "rb") as f:
                manifest["scenes"][fn] = {"sha256": hashlib.sha256(f.read()).hexdigest()}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
