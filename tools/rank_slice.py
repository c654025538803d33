#!/usr/bin/env python3
"""Developer tool (GPU box): what ONE rank of an N-GPU run does per frame, timed on one GPU.

    python tools/rank_slice.py [N ...]        (default 1 2 4 8; chapter_11 at 3840x2160, 16-row bands round-robin)

For each N and each rank r < N: the kernel time of the launch that renders bands r, r+N, ... (tile ordering pass
included), best of a few runs.  max over r is the compute time of an N-GPU frame before any gather."""
import os
import sys

import torch  # first: the library then binds to the HIP runtime torch has loaded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_rs_amd as rt  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    sf = rt.SceneFile.from_file(os.path.join(ROOT, "tests", "golden", "scenes", "chapter_11.toml"))
    cam = sf.build_camera(3840, 2160)
    scene = sf.build_world().scene()
    buf = torch.empty(3840 * 2160 * 3, dtype=torch.float64, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for n in ns:
        times = []
        for r in range(n):
            best = 1e9
            for _ in range(6):
                ev[0].record()
                scene.render_device_bands(cam, buf.data_ptr(), 16, r, n)
                ev[1].record()
                torch.cuda.synchronize()
                best = min(best, ev[0].elapsed_time(ev[1]))
            times.append(best)
        one = times[0] if n == 1 else None
        print(f"N={n}: per-rank kernel ms " + " ".join(f"{t:.3f}" for t in times) + f"  max {max(times):.3f}  sum {sum(times):.3f}",
              flush=True)


if __name__ == "__main__":
    main()
