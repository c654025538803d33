"""One rank of the multi-process band-gather rehearsal (CPU, gloo).  Started by test_dist_gloo.py.

Each rank fills its send buffer with the rows of ITS bands -- rendered by the CPU oracle, standing in for the GPU
kernel, which needs a device -- and takes part in BandGather.gather(); rank 0 saves the assembled frame.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from raytracer_rs_amd.dist import BandGather, band_rows_range, bands_of  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--scene", required=True)
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    cam, world = orc.load_scene(a.scene, a.width, a.height)
    bg = BandGather(a.width, a.height, a.rank, a.world, torch.device("cpu"), a.band_rows)
    full = np.zeros((a.height, a.width, 3))
    for slot, band in enumerate(bands_of(a.rank, a.world, a.height, a.band_rows)):
        y0, y1 = band_rows_range(band, a.height, a.band_rows)
        rows = orc.render(cam, world, rows=(y0, y1))[y0:y1]  # render() returns a whole-frame array
        full[y0:y1] = rows
        flat = torch.from_numpy(np.ascontiguousarray(rows).reshape(-1))
        bg.mine[slot, : flat.numel()] = flat
    for _ in range(2):  # twice: the buffers are reused frame after frame
        frame = bg.gather()
    if a.rank == 0:
        np.save(a.out, bg.frame_rows().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
