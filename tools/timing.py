#!/usr/bin/env python3
"""Developer tool (GPU box): phase timing of the render kernel from a library built with -DRT_TIMING.

    RT_AMD_LIB=ab/timing.so python tools/timing.py [scene.toml] [width] [height]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_rs_amd as rt  # noqa: E402
from raytracer_rs_amd import _lib  # noqa: E402


def main():
    scene_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "scenes", "chapter_11.toml")
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
    if scene_path == "large":
        from raytracer_rs_amd import synthetic
        sf = rt.SceneFile.from_str(synthetic.large_scene())
    else:
        sf = rt.SceneFile.from_file(scene_path)
    scene = sf.build_world().scene()
    cam = sf.build_camera(w, h)
    lib = _lib.lib()
    import numpy as np

    bands = None
    rows_rendered = h
    if os.environ.get("RT_TIMING_BANDS"):  # "band_rows,first,step": time one rank's share of a multi-GPU frame
        bands = tuple(int(v) for v in os.environ["RT_TIMING_BANDS"].split(","))
        rows_rendered = sum(min(bands[0], h - y) for y in range(bands[1] * bands[0], h, bands[0] * bands[2]))
    n_waves = ((w + 7) // 8) * ((rows_rendered + 7) // 8)
    buf = np.zeros((n_waves, 8), dtype=np.uint64)
    # one launch per frame, into device memory (rt_render cuts a frame into groups of rows, each a launch of its own
    # whose waves would overwrite each other's records)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    frame = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(frame), w * rows_rendered * 3) == 0
    def launch(**kw):
        if bands:
            scene.render_device_bands(cam, frame.value, *bands, quantized=True, **kw)
        else:
            scene.render_device(cam, frame.value, quantized=True, **kw)

    for _ in range(3):
        launch()
    launch(timed=True)
    assert hip.hipDeviceSynchronize() == 0

    class _St:
        kernel_ms = sum(scene.render_times())
    st = _St()
    lib.rt_debug_timing(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(n_waves))
    m32, m40 = np.uint64(0xFFFFFFFF), np.uint64((1 << 40) - 1)
    trace, shade = (buf[:, 0] & m32).astype(float), (buf[:, 0] >> np.uint64(32)).astype(float)
    life, rlife = (buf[:, 3] & m40).astype(float), (buf[:, 3] >> np.uint64(40)).astype(float)
    rbegin = buf[:, 1].astype(np.int64)
    rend = rbegin + rlife.astype(np.int64)
    span = (rend.max() - rbegin.min()) / 100.0  # us, 100 MHz constant clock
    print(f"shader clock = {life.sum() / rlife.sum() * 100:.0f} MHz; first wave start to last wave end {span:.0f} us; "
          f"sum of wave lifetimes {rlife.sum() / 100:.0f} us = {rlife.sum() / 100 / span:.0f} waves resident on average")
    order = np.argsort(rbegin)
    late = order[-4096:]
    print(f"last 4096 waves to start: mean lifetime {rlife[late].mean() / 100:.0f} us, they start {(rbegin[late].min() - rbegin.min()) / 100:.0f} us in; "
          f"last wave ends {(rend.max() - rbegin.min()) / 100:.0f} us in; 99% of waves have ended by {(np.sort(rend)[int(0.99 * len(rend))] - rbegin.min()) / 100:.0f} us")
    hist_t = np.linspace(rbegin.min(), rend.max(), 21)
    res = [int(((rbegin <= t) & (rend > t)).sum()) for t in hist_t]
    print("resident waves over time:", res)
    iters = (buf[:, 2] & np.uint64(0xFFFFFFFF)).astype(float)
    lanes = (buf[:, 2] >> np.uint64(32)).astype(float)
    print(f"kernel {st.kernel_ms:.3f} ms; waves {n_waves}; iterations/wave {iters.mean():.2f} (max {iters.max():.0f}); "
          f"active lanes/iteration {lanes.sum() / iters.sum():.1f}")
    print(f"per wave: lifetime {life.mean():.0f} clk (max {life.max():.0f}), trace {trace.mean():.0f} clk, shade {shade.mean():.0f} clk")
    print(f"per iteration: trace {trace.sum() / iters.sum():.0f} clk, shade {shade.sum() / iters.sum():.0f} clk")
    hist = buf[:, 7]
    le = [int(((hist >> np.uint64(16 * k)) & np.uint64(0xFFFF)).sum()) for k in range(4)]
    print("rounds by lanes holding a ray afterwards: <= 8: %.1f %%, <= 16: %.1f %%, <= 32: %.1f %%, <= 48: %.1f %% of %d rounds"
          % tuple([100.0 * v / iters.sum() for v in le] + [int(iters.sum())]))
    skipn, groupn, leafn = (buf[:, 4] & m32).astype(float), (buf[:, 4] >> np.uint64(32)).astype(float), (buf[:, 5] & m32).astype(float)
    again = (buf[:, 5] >> np.uint64(32)).astype(float)
    nodes = skipn + groupn + leafn
    print(f"node records per traversal (wave level): skip {skipn.sum() / iters.sum():.1f}, group {groupn.sum() / iters.sum():.1f}, "
          f"leaf {leafn.sum() / iters.sum():.1f}, next record guessed wrong {again.sum() / iters.sum():.1f}; trace clocks per node {trace.sum() / max(1.0, nodes.sum()):.0f}")
    worst = int(np.argmax(life))
    print(f"longest wave: {iters[worst]:.0f} iterations, {lanes[worst] / iters[worst]:.1f} active lanes, {nodes[worst] / iters[worst]:.0f} nodes "
          f"per traversal, {trace[worst] / max(1.0, nodes[worst]):.0f} trace clocks per node, lifetime {life[worst] / 1e6:.1f} Mclk")
    t0 = rbegin.min()
    p99 = np.sort(rend)[int(0.99 * len(rend))]
    tail = np.nonzero(rend > p99)[0]
    tail = tail[np.argsort(-rend[tail])][:12]
    print("the tail (last waves to end): start us / lifetime us / rounds / start rank in the launch")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    tiles_x = (w + 7) // 8
    tile_of = buf[:, 6].astype(np.int64)
    for k in tail:
        print(f"   {(rbegin[k] - t0) / 100:7.0f} {rlife[k] / 100:7.0f} {iters[k]:4.0f} {rank[k]:8d}   tile x {tile_of[k] % tiles_x:4d} y {tile_of[k] // tiles_x:4d}")
    # how well the launch order anticipates the cost: lifetimes by place in the order
    by_rank = rlife[order] / 100
    parts = np.array_split(by_rank, 16)
    print("wave lifetime us by place in the launch order, sixteenths (mean / max):", " ".join(f"{p.mean():.0f}/{p.max():.0f}" for p in parts))
    started_late = np.nonzero((rend > p99) & (rbegin - t0 > 0.25 * (rend.max() - t0)))[0]
    print(f"waves ending after the 99th percentile: {len(np.nonzero(rend > p99)[0])}, of which started after the first quarter of the launch: {len(started_late)}")
    by_tile = np.zeros(n_waves)
    by_tile[tile_of] = life
    rows = by_tile.reshape(-1, tiles_x).mean(axis=1)
    step = max(1, len(rows) // 18)
    print("mean wave lifetime by tile row (top to bottom):", " ".join(f"{rows[i:i + step].mean() / 1000:.0f}k" for i in range(0, len(rows), step)))


if __name__ == "__main__":
    main()
