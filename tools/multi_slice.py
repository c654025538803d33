#!/usr/bin/env python3
"""What ONE GPU of an N-GPU rt_multi run has to do, timed on the one GPU of a build box.

rt_multi cuts the frame into bands and every GPU takes the next band from a shared cursor, two in flight on two
streams.  With N equally fast GPUs that comes out close to "GPU r renders bands r, r+N, r+2N, ...", which is what
this script launches for every r, the way a worker does: separate launches per band through rt_render_device,
alternating between two scene handles on two streams, at most two bands in flight.  Prints, per N, the slowest r
(the frame time of an N-GPU run before any transfer) and N x that over the one-GPU time.

    python tools/multi_slice.py [--config c3] [--ns 1,2,4,8] [--band-rows R] [--frames 30]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracer_rs_amd as rt  # noqa: E402


def default_band_rows(width: int, height: int, workers: int) -> int:
    """rt_multi.cpp's default_band_rows."""
    whole = (height + 7) // 8 * 8
    if workers <= 1:
        return whole
    tiles_x = (width + 7) // 8
    tiles = tiles_x * ((height + 7) // 8)
    per_band = max(8192, tiles // (4 * workers))
    tile_rows = max(1, (per_band + tiles_x - 1) // tiles_x)
    return min(whole, tile_rows * 8)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--ns", default="1,2,4,8")
    ap.add_argument("--band-rows", type=int, default=0)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--f64", action="store_true")
    ap.add_argument("--interleaved", type=int, default=0, metavar="STRIPE",
                    help="units of interleaved STRIPE-row stripes (unit u of U = stripes u, u+U, ...) instead of contiguous bands; "
                         "--units-per-gpu of them per GPU")
    ap.add_argument("--units-per-gpu", type=int, default=2)
    a = ap.parse_args()
    sf, _ = bench.scene_file_of(a.config)
    name, W, H = bench.CONFIGS[a.config]
    cam = sf.build_camera(W, H)
    handles = [sf.build_world().scene(0) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    events = [torch.cuda.Event(), torch.cuda.Event()]
    bpp = 24 if a.f64 else 3
    frame = torch.empty(W * H * bpp, dtype=torch.uint8, device="cuda")
    base = None
    for n in [int(v) for v in a.ns.split(",")]:
        if a.interleaved:
            U = a.units_per_gpu * n if n > 1 else 1
            slowest = 0.0
            for r in range(n):
                mine = list(range(r, U, n))

                def one_frame():
                    for k, u in enumerate(mine):
                        slot = k & 1
                        if k >= 2:
                            events[slot].synchronize()
                        handles[slot].render_device_bands(cam, frame.data_ptr() + u * (W * H * bpp // U // 256 * 256), a.interleaved, u, U, None,
                                                          streams[slot].cuda_stream, False, False, not a.f64)
                        events[slot].record(streams[slot])
                    torch.cuda.synchronize()

                for _ in range(10):
                    one_frame()
                t0 = time.perf_counter()
                for _ in range(a.frames):
                    one_frame()
                slowest = max(slowest, (time.perf_counter() - t0) / a.frames * 1e3)
            base = base or slowest
            print(f"{name} {W}x{H}  N={n}  {U} units of interleaved {a.interleaved}-row stripes  slowest share {slowest:.4f} ms  "
                  f"N x share / one GPU = {n * slowest / base:.3f}", flush=True)
            continue
        rows = a.band_rows or default_band_rows(W, H, n)
        n_bands = (H + rows - 1) // rows
        slowest = 0.0
        for r in range(n):
            mine = list(range(r, n_bands, n))

            def one_frame():
                for k, b in enumerate(mine):
                    slot = k & 1
                    if k >= 2:
                        events[slot].synchronize()
                    y0, y1 = b * rows, min(H, (b + 1) * rows)
                    handles[slot].render_device(cam, frame.data_ptr() + y0 * W * bpp, (y0, y1), None, streams[slot].cuda_stream, False, False, not a.f64)
                    events[slot].record(streams[slot])
                torch.cuda.synchronize()

            for _ in range(10):
                one_frame()
            t0 = time.perf_counter()
            for _ in range(a.frames):
                one_frame()
            ms = (time.perf_counter() - t0) / a.frames * 1e3
            slowest = max(slowest, ms)
        base = base or slowest
        print(f"{name} {W}x{H}  N={n}  bands of {rows} rows ({n_bands})  slowest share {slowest:.4f} ms  N x share / one GPU = {n * slowest / base:.3f}", flush=True)


if __name__ == "__main__":
    main()
