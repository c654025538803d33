"""The N > 1 path on CPU: row bands dealt round-robin to ranks, one gather to rank 0, bands placed in the frame."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_band_assignment_covers_every_row_once():
    from raytracer_rs_amd.dist import band_rows_range, bands_of, bands_per_rank, n_bands

    for height in (1, 15, 16, 17, 50, 600, 2160):
        for world in (1, 2, 3, 4, 8):
            for rows in (4, 16):
                seen = np.zeros(height, dtype=int)
                for r in range(world):
                    mine = bands_of(r, world, height, rows)
                    assert len(mine) <= bands_per_rank(world, height, rows)
                    for b in mine:
                        y0, y1 = band_rows_range(b, height, rows)
                        seen[y0:y1] += 1
                assert (seen == 1).all()
                assert sum(len(bands_of(r, world, height, rows)) for r in range(world)) == n_bands(height, rows)


@pytest.mark.parametrize("world,width,height,band_rows", [(2, 40, 50, 16), (3, 33, 21, 4)])
def test_gather_assembles_the_frame(orc, tmp_path, world, width, height, band_rows):
    port = free_port()
    out = str(tmp_path / "frame.npy")
    scene = scene_path("chapter_11")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--rank", str(r),
                               "--world", str(world), "--port", str(port), "--scene", scene, "--width", str(width),
                               "--height", str(height), "--band-rows", str(band_rows), "--out", out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    cam, w = orc.load_scene(scene, width, height)
    expected = orc.render(cam, w)
    got = np.load(out)
    assert got.shape == expected.shape
    assert np.array_equal(got, expected)
