#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the 3840x2160 chapter_11 frame (reflection/refraction, depth 5).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one whole frame.  `--gpus N` uses N GPUs however the script is started:
  * started plainly (no WORLD_SIZE in the environment) with N > 1, ONE process drives all N GPUs through the
    library's rt_multi_* entry points: a worker thread per GPU, bands of rows taken from a shared atomic cursor as
    the GPUs become free (dynamic stealing), two bands in flight per GPU, finished bands collected on GPU 0 with RCCL
    (or peer copies) while other GPUs still render.  The measurement runs in a child process that the parent watches:
    if the RCCL collection fails or stalls, the run is repeated with peer copies and the line says so.  Fewer than N
    visible GPUs is an error (RT_BENCH_DEVICES=0,0 lists device ordinals explicitly, e.g. to rehearse two workers on one
    GPU).
  * started by torchrun, one process per GPU over torch.distributed, as described next.
`config.workload` says which of the two ran.

With N ranks (one process per GPU) the frame's rows are cut into 16-row bands dealt
round-robin to the ranks (band b -> rank b % N; band cost varies from ~1 ray/px in the sky to ~30 rays/px on the glass,
and the fine interleave evens that out); every rank renders its bands with the HIP library into device memory as the
8-bit frame a PPM is written from (canvas.rs:123-130 applied by the render kernel; `--frame f64` keeps the f64 canvas
instead), and the bands are gathered to rank 0 with one RCCL gather (grouped send/recv over xGMI) and copied into the
frame.  The gather of frame k runs on its own stream and overlaps the rendering of frame k+1.  The scene is
replicated, so the gather is the only collective.  The timed region starts with scene, camera and output buffers
resident in HBM; it covers kernel + gather + placement of the bands in the frame, K times, between
barrier+synchronize pairs, and rank 0 reports the maximum over ranks.

Rank 0 prints ONE JSON line.  `value` = rays of the frame (a ray = one World::intersect call of the reference:
camera, shadow, reflection and refraction rays; counted once by the library's counting kernel) x K / time.
`roofline` prices the render kernel against the FP64 vector-ALU peak (nothing on this path is a dense contraction and
the scene fits in cache, so neither "mfma" nor "hbm" is the bound that applies), both the nominal figure and the
rates a saturating v_fma_f64 / v_mul_f64+v_add_f64 stream reaches on this very GPU in this very run
(tools/fp64_peak.hip).  On one GPU the line also carries: `f64_frame` (the same frame stored as the f64 canvas),
`two_frames_in_flight` (frames alternating between two handles on two streams), `camera_render_ms` (the drop-in call itself, rt_render = Camera::render into HOST memory, PCIe included),
`other_configs` (the other benchmark configurations of BASELINE.json, each with rays, kernel time and roofline
fraction) and `cpu_baseline` (the CPU oracle -- the C restatement of the reference's algorithm, single-threaded like
the reference and repeating its per-ray matrix inversions -- on a bounded sample of rows of the same frame on this
box's host cores).
"""
import argparse
import ctypes
import hashlib
import json
import os
import re
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracer_rs_amd as rt  # noqa: E402
from raytracer_rs_amd.dist import BandGather  # noqa: E402

BAND_ROWS = 16
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 f64 lanes/clk x 2 (FMA) x 2.4 GHz = half the 157.3 TF f32 vector peak

# Algorithmic flops per operation of the reference (add/sub/mul/div/sqrt/pow = 1 each; compares, abs, neg, min/max,
# floor free).  ray_transform is the reference's full 3x4 product (33); the kernel skips the products with the exact
# zeros of a scale-and-translate inverse and executes 9 for those -- the algorithmic figure does not.
FLOPS = {
    "ray_gen": 36, "ray_transform": 33, "aabb": 12,
    "prim_miss": (21, 0, 12, 44, 52), "prim_hit": (27, 1, 12, 44, 52),  # sphere, plane, cube, cylinder, cone
    "hit_record": 79, "shadow_setup": 18, "lighting_unlit": 29, "lighting_lit": 61, "pattern": 40,
    "reflect": 3, "refract_setup": 26, "schlick": 16, "combine": 6, "combine_schlick_extra": 7,
}

CONFIGS = {  # BASELINE.json configs[1..4]
    "c2": ("chapter_7_and_8", 1920, 1080),
    "c3": ("chapter_11", 3840, 2160),
    "c4": ("chapter_14", 3840, 2160),
    "c5": ("generated 10 001-leaf hierarchy (raytracer_rs_amd.synthetic.large_scene)", 7680, 4320),
}


def algorithmic_flops(s: dict) -> float:
    f = FLOPS["ray_gen"] * s["primary"] + FLOPS["ray_transform"] * s["ray_transforms"] + FLOPS["aabb"] * s["aabb_tests"]
    for k in range(5):
        f += FLOPS["prim_hit"][k] * s["prim_hits"][k] + FLOPS["prim_miss"][k] * (s["prim_tests"][k] - s["prim_hits"][k])
    f += (FLOPS["hit_record"] + FLOPS["combine"]) * s["shades"] + FLOPS["shadow_setup"] * s["shadow"]
    f += FLOPS["lighting_lit"] * s["lit"] + FLOPS["lighting_unlit"] * (s["shades"] - s["lit"])
    f += FLOPS["pattern"] * s["patterns"] + FLOPS["reflect"] * s["reflect"] + FLOPS["refract_setup"] * s["refract_setups"]
    f += (FLOPS["schlick"] + FLOPS["combine_schlick_extra"]) * s["schlick"]
    return float(f)


def scene_file_of(config: str):
    """(SceneFile, path or None) of a benchmark configuration."""
    if config == "c5":
        from raytracer_rs_amd import synthetic

        return rt.SceneFile.from_str(synthetic.large_scene()), None
    path = os.path.join(ROOT, "tests", "golden", "scenes", CONFIGS[config][0] + ".toml")
    return rt.SceneFile.from_file(path), path


def measured_fp64_peak() -> dict:
    """FP64 vector rates this GPU delivers right now: saturating v_fma_f64 stream, and v_mul_f64 / v_add_f64 stream."""
    path = os.path.join(ROOT, "raytracer_rs_amd", "librt_peak.so")
    if not os.path.exists(path):
        return {}
    lib = ctypes.CDLL(path)
    f, u = ctypes.c_double(), ctypes.c_double()
    if lib.rt_fp64_peak(ctypes.byref(f), ctypes.byref(u)) != 0:
        return {}
    return {"fma": f.value, "unfused": u.value}


def device_source_hash() -> str:
    """Identifies the kernel sources a PMC figure belongs to: a hash of their code, comments and blank space left out
    (a reworded comment does not make a measurement stale)."""
    h = hashlib.sha256()
    for name in ("rt_device.hip", "rt_levels.inl", "rt_device_types.h", "rt_flatten.cpp"):
        with open(os.path.join(ROOT, "raytracer_rs_amd", "csrc", name), "r", encoding="utf-8") as f:
            text = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
        for line in text.splitlines():
            code = " ".join(line.split("//", 1)[0].split())
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def committed_pmc(config: str) -> dict:
    """HBM bytes and executed f64 flops per launch of a configuration's render kernel from the committed PMC passes
    (profiles/pmc_traffic.json, one entry per configuration) -- only while the kernel sources are the ones that were
    profiled (their hash travels with the figures)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception:
        return {"traffic": None, "traffic_note": "no committed PMC summary"}
    if t.get("source_sha256") != device_source_hash():
        return {"traffic": None, "traffic_note": "profiles/pmc_traffic.json was taken on other kernel sources; not carried over"}
    e = t.get("configs", {}).get(config)
    if not e:
        return {"traffic": None, "traffic_note": f"no committed PMC passes for {config}"}
    return {"traffic": e.get("hbm_bytes_per_launch"), "traffic_profile": e.get("profile"),
            "executed_flops": e.get("executed_f64_flops_per_launch")}


def kernel_name(scene) -> str:
    info = scene.info()
    return {0: "render_kernel<false, 0, false>", 2: "render_kernel<false, 2, true>", 4: "primary_kernel (one round: nothing in the world reflects or transmits)",
            3: "level_kernel (one launch per level of the ray trees, plus sums; scans and numbering for a view's first frame)"}.get(
                info.kernel_build, "render_kernel<false, 1, true>")


def time_config(scene, cam, stream, reps: int, quantized: bool, config: str = None, peak: dict = None) -> dict:
    """Rays, flops and GPU time of one configuration on one GPU (frame stays in HBM).  Two flop figures, never mixed:
    the REFERENCE's work for the same rays (every ray against every object of every list it enters; what `frac` prices,
    as the roofline contract asks) and the f64 flops the kernel EXECUTED (PMC passes committed under profiles/, carried
    only while the kernel sources are unchanged) -- range-skip boxes, balls and sign tests spare the kernel part of
    the reference's work, so on scenes with groups the first is a speed-up over the reference's algorithm, not a
    hardware utilisation."""
    W, H = cam.hsize, cam.vsize
    buf = torch.empty(W * H * 3, dtype=torch.uint8 if quantized else torch.float64, device="cuda")
    st = scene.render_device(cam, buf.data_ptr(), None, None, stream, True, False, quantized)
    counts = st.as_dict()
    for _ in range(2):
        scene.render_device(cam, buf.data_ptr(), None, None, stream, False, False, quantized)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        scene.render_device(cam, buf.data_ptr(), None, None, stream, False, True, quantized)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / reps * 1e3
    times = [scene.render_times(back) for back in range(min(reps, 32))]
    kernel_ms = sum(t[1] for t in times) / len(times)
    ordering_ms = sum(t[0] for t in times) / len(times)
    flops = algorithmic_flops(counts)
    ref_tf = flops / (kernel_ms * 1e-3) / 1e12
    out = {"rays": counts["rays"], "rays_per_pixel": round(counts["rays"] / (W * H), 3), "ms_per_frame": round(wall_ms, 4),
           "kernel_ms": round(kernel_ms, 4), "tile_ordering_ms": round(ordering_ms, 4),
           "mrays_per_s": round(counts["rays"] / wall_ms / 1e3, 1), "algorithmic_flops": flops,
           "reference_work_tflops": round(ref_tf, 3),
           "frac": round(ref_tf / FP64_VECTOR_PEAK_TFLOPS, 4), "kernel": kernel_name(scene),
           "scene_device_mb": round(scene.info().device_bytes / 1e6, 1)}  # tables, and the ray queues of worlds rendered level by level
    if peak:
        out["frac_reference_work_of_measured_unfused"] = round(ref_tf / peak["unfused"], 4)
    if scene.info().kernel_build == 3:
        # ms_per_frame above is a frame of a view the handle has rendered before: the numbering of its rays (scans, a
        # sort per level) and the launch order of its waves are remembered per view.  What a view's FIRST frame costs,
        # waits per level included: all rows but the last eight are a view of their own whose queues exist already.
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scene.render_device(cam, buf.data_ptr(), (0, H - 8), None, stream, False, False, quantized)
        torch.cuda.synchronize()
        out["new_view_frame_ms"] = round((time.perf_counter() - t0) * 1e3 * H / (H - 8), 3)
        out["new_view_note"] = "first frame of a view (camera, rows, depth): one wait per level, numbering and sort of every level's rays, wave costs recorded; ms_per_frame is every later frame of the view"
    if config:
        pmc = committed_pmc(config)
        out["hbm_bytes"] = pmc.get("traffic")
        if pmc.get("executed_flops"):
            ex_tf = pmc["executed_flops"] / (kernel_ms * 1e-3) / 1e12
            out["executed_flops"] = pmc["executed_flops"]
            out["executed_tflops"] = round(ex_tf, 3)
            out["frac_executed"] = round(ex_tf / FP64_VECTOR_PEAK_TFLOPS, 4)
            if peak:
                out["frac_executed_of_measured_unfused"] = round(ex_tf / peak["unfused"], 4)
            out["pmc_profile"] = pmc.get("traffic_profile")
        else:
            out["pmc_note"] = pmc.get("traffic_note")
    return out


def ppm_encode_ms(scene, cam) -> float:
    """Canvas::to_ppm_string (canvas.rs:44-90) for the configuration's frame: the P3 writer on the host, one call of
    rt_canvas_to_ppm into a buffer of the right size, frame already in host memory.  Not part of ms/frame (reported
    beside it)."""
    import ctypes

    from raytracer_rs_amd import _lib

    img, _ = scene.render(cam)
    canvas = rt.Canvas(cam.hsize, cam.vsize, img)
    n = _lib.lib().rt_canvas_to_ppm(canvas._ptr(), cam.hsize, cam.vsize, None, 0)
    buf = ctypes.create_string_buffer(n)
    t0 = time.perf_counter()
    _lib.lib().rt_canvas_to_ppm(canvas._ptr(), cam.hsize, cam.vsize, buf, n)
    dt = (time.perf_counter() - t0) * 1e3
    del buf, canvas, img
    return round(dt, 1)


def cpu_baseline(scene_path, width: int, height: int, seconds_target: float = 12.0, scene_text: str = None) -> dict:
    """Times the CPU oracle on evenly spaced rows of the same frame (checker / baseline only).  `scene_path` is a
    scene file; a generated scene comes as `scene_text`."""
    from oracle import oracle as orc

    import tomli

    def load(w, h):
        if scene_path:
            return orc.load_scene(scene_path, w, h)
        return orc.scene_from_dict(tomli.loads(scene_text), w, h)

    # The reference's own schedule: per-ray inverses, intersection list + sort, one thread.  Calibrate on the middle
    # row of a 64x smaller frame, then sample rows of the frame itself -- or, when a single row of it would take more
    # than a quarter of the time target (thousands of objects), of the same picture at a lower resolution.
    cam, world = load(max(8, width // 64), max(8, height // 64))
    world.set_mode(1)
    t0 = time.perf_counter()
    orc.render(cam, world, threads=1, rows=(cam.vsize // 2, cam.vsize // 2 + 1), counters=orc.Counters())
    per_pixel = (time.perf_counter() - t0) / cam.hsize
    shrink = 1
    while shrink < 64 and per_pixel * (width // shrink) > seconds_target / 4:
        shrink *= 2
    sw, sh = width // shrink, height // shrink
    cam, world = load(sw, sh)
    world.set_mode(1)
    n_rows = int(max(1, min(sh, seconds_target / max(per_pixel * sw, 1e-6))))
    ys = sorted(set(int((i + 0.5) * sh / n_rows) for i in range(n_rows)))
    c = orc.Counters()
    t0 = time.perf_counter()
    for y in ys:
        orc.render(cam, world, threads=1, rows=(y, y + 1), counters=c)
    dt = time.perf_counter() - t0
    of_what = f"the {width}x{height} frame" if shrink == 1 else f"the same picture at {sw}x{sh} (a row of the {width}x{height} frame alone would take {per_pixel * width:.0f} s)"
    out = {"value": round(c.rays / dt / 1e6, 6), "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"{len(ys)} evenly spaced rows of {of_what} ({c.rays} rays, {dt:.1f} s), "
                     f"oracle in reference-equivalent mode (per-ray 4x4 inverses, intersection list + stable sort)"}
    cam, world = load(width, height)
    # the same arithmetic with inverses cached, all host cores the box gives us
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    world.set_mode(0)
    c2 = orc.Counters()
    t0 = time.perf_counter()
    orc.render(cam, world, threads=1, rows=(height // 2, height // 2 + 1), counters=c2)
    per_row = time.perf_counter() - t0
    half = int(max(1, min(40, seconds_target * cores / max(per_row, 1e-6) / 2 / 2)))
    c2 = orc.Counters()
    rows2 = (height // 2 - half, height // 2 + half)
    t0 = time.perf_counter()
    orc.render(cam, world, threads=cores, rows=rows2, counters=c2)
    dt2 = time.perf_counter() - t0
    out["optimised"] = {"value": round(c2.rays / dt2 / 1e6, 3), "unit": "Mrays/s", "cores": cores,
                        "sample": f"rows {rows2[0]}..{rows2[1]} ({c2.rays} rays, {dt2:.1f} s), inverses and bounds cached"}
    return out


def bench_devices(args) -> list:
    """Device ordinals of the single-process several-GPU run: RT_BENCH_DEVICES, else 0 .. N-1 (all of which must exist)."""
    listed = os.environ.get("RT_BENCH_DEVICES")
    if listed:
        return [int(v) for v in listed.split(",") if v.strip() != ""]
    visible = torch.cuda.device_count()
    if visible < args.gpus:
        print(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible to this process "
              f"(RT_BENCH_DEVICES=0,0,... rehearses several workers on fewer GPUs)", file=sys.stderr, flush=True)
        raise SystemExit(4)
    return list(range(args.gpus))


def workload_of(args):
    """(SceneFile, scene path or None, name, width, height) of the run."""
    if args.scene:
        sf, scene_path, name = rt.SceneFile.from_file(args.scene), args.scene, os.path.basename(args.scene)
        width, height = args.width or 3840, args.height or 2160
    else:
        sf, scene_path = scene_file_of(args.config)
        name, width, height = CONFIGS[args.config]
        width, height = args.width or width, args.height or height
    return sf, scene_path, name, width, height


def data_note(args, name: str) -> str:
    """The `data` field: nothing here is measured data; the scenes are the reference's own files or generated."""
    if args.scene:
        return f"synthetic (scene file {name}; no datasets on this path)"
    if args.config == "c5":
        return "synthetic (scene generated by raytracer_rs_amd.synthetic.large_scene, seed 0x5EED0001, in the reference's TOML schema)"
    return (f"synthetic in the sense that no dataset exists for this path: the reference's own scene file scenes/{name}.toml "
            f"(byte-identical copy under tests/golden/scenes/), camera resized to the benchmark's frame")


def run_multi_parent(args) -> None:
    """bench.py --gpus N without a launcher: runs the measurement in a child process (this one never touches a GPU)
    and watches it; a failed or stalled RCCL collection is repeated with peer copies."""
    import subprocess

    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ["--multi-child"]
    limit_all = float(os.environ.get("RT_BENCH_CHILD_TIMEOUT", "420"))
    attempts = [dict()] if os.environ.get("RT_MULTI_GATHER") else [dict(), dict(RT_MULTI_GATHER="copy")]
    note = None
    for k, extra in enumerate(attempts):
        env = dict(os.environ, **extra)
        if k == 0 and len(attempts) > 1:
            limit = min(limit_all, 300.0)  # (the repeat with peer copies must still fit if RCCL stalls)
        else:
            limit = limit_all
        if note:
            env["RT_BENCH_NOTE"] = note
        try:
            done = subprocess.run(base, env=env, capture_output=True, text=True, timeout=limit, start_new_session=True)
        except subprocess.TimeoutExpired as e:
            note = f"first attempt (gather via RCCL) did not finish within {limit:.0f} s and was stopped"
            sys.stderr.write((e.stderr or b"").decode(errors="replace")[-2000:] if isinstance(e.stderr, bytes) else (e.stderr or "")[-2000:])
            continue
        sys.stderr.write(done.stderr[-4000:])
        if done.returncode == 0 and done.stdout.strip():
            print(done.stdout.strip().splitlines()[-1], flush=True)
            return
        note = f"first attempt (gather via RCCL) exited with status {done.returncode}"
        if k + 1 == len(attempts) or done.returncode in (3, 4):  # a differing frame or too few GPUs: nothing a repeat would cure
            raise SystemExit(done.returncode or 1)
    raise SystemExit(1)


def run_multi_child(args) -> None:
    """One process, N GPUs: rt_multi_* (worker thread per GPU, shared band cursor, two bands in flight per GPU,
    collection on devices[0] overlapped with rendering)."""
    devices = bench_devices(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    first = devices[0]
    torch.cuda.set_device(first)
    dev = torch.device("cuda", first)
    sf, scene_path, name, width, height = workload_of(args)
    cam = sf.build_camera(width, height)
    world = sf.build_world()
    W, H = cam.hsize, cam.vsize
    u8 = args.frame == "u8"
    frame = torch.empty(W * H * 3, dtype=torch.uint8 if u8 else torch.float64, device=dev)
    band_rows = int(os.environ.get("RT_BENCH_BAND_ROWS", "0"))  # developer knobs: stripe height and number of units
    units = int(os.environ.get("RT_BENCH_UNITS", "0"))
    # Two ways for the other GPUs' units to reach the first GPU exist (RCCL send/receive pairs, peer copies behind each
    # kernel); which one is faster depends on the frame and the number of GPUs.  With more than one GPU and no choice made
    # in the environment both are set up and tried on a few untimed frames, and the timed region runs the faster one.
    calibration = {}
    t_create = time.perf_counter()
    # (RT_BENCH_CALIBRATE=1 runs the comparison on one GPU too, where both handles collect nothing: a rehearsal of this code)
    if (len(set(devices)) > 1 or os.environ.get("RT_BENCH_CALIBRATE")) and not os.environ.get("RT_MULTI_GATHER"):
        best = None
        for mode in ("rccl", "copy"):
            os.environ["RT_MULTI_GATHER"] = mode
            try:
                candidate = world.multi_scene(devices)
            except rt.RtError as e:
                calibration[mode] = f"not available: {e}"
                continue
            finally:
                del os.environ["RT_MULTI_GATHER"]
            for _ in range(10):
                candidate.render_device(cam, frame.data_ptr(), band_rows=band_rows, quantized=u8, units=units)
            for d in sorted(set(devices)):
                torch.cuda.synchronize(d)
            t0 = time.perf_counter()
            for _ in range(20):
                candidate.render_device(cam, frame.data_ptr(), band_rows=band_rows, quantized=u8, units=units)
            for d in sorted(set(devices)):
                torch.cuda.synchronize(d)
            ms = (time.perf_counter() - t0) / 20 * 1e3
            calibration[mode] = round(ms, 4)
            if best is None or ms < best[0]:
                if best is not None:
                    best[1].close()
                best = (ms, candidate)
            else:
                candidate.close()
        if best is None:
            raise SystemExit("neither way of collecting the frame could be set up: " + json.dumps(calibration))
        multi = best[1]
    else:
        multi = world.multi_scene(devices)
    create_ms = (time.perf_counter() - t_create) * 1e3

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    st = multi.render_device(cam, frame.data_ptr(), band_rows=band_rows, stats=True, quantized=u8, units=units)
    counts = st.as_dict()
    rays_frame = counts["rays"]
    flops = algorithmic_flops(counts)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.1:
        multi.render_device(cam, frame.data_ptr(), band_rows=band_rows, quantized=u8, units=units)
    prewarm_s = time.perf_counter() - t_pre
    for _ in range(args.warmup):
        multi.render_device(cam, frame.data_ptr(), band_rows=band_rows, quantized=u8, units=units)
    sync_all()
    render_ms, gather_ms, taken, worker_ms = [], [], [0] * len(devices), [0.0] * len(devices)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = multi.render_device(cam, frame.data_ptr(), band_rows=band_rows, quantized=u8, units=units)
        render_ms.append(st.kernel_ms)
        gather_ms.append(st.gather_ms)
    sync_all()
    elapsed = time.perf_counter() - t0
    # one more (untimed) frame for the per-worker figures
    multi.render_device(cam, frame.data_ptr(), band_rows=band_rows, quantized=u8, units=units)
    taken, worker_ms, info = multi.units_taken(), multi.worker_ms(), multi.info()

    scene = world.scene(first)
    stream = torch.cuda.current_stream().cuda_stream
    if args.check:
        whole = torch.zeros(H * W * 3, dtype=torch.float64, device=dev)
        scene.render_device(cam, whole.data_ptr(), None, None, stream, False)
        if u8:
            q = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
            rt.quantize_device(whole.data_ptr(), whole.numel(), q.data_ptr(), stream)
            whole = q
        torch.cuda.synchronize()
        same = torch.equal(frame, whole)
        print(f"check: collected frame {'equals' if same else 'DIFFERS FROM'} the single-GPU frame", file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(3)
    # the same frame through the one-GPU entry point on devices[0], for the ratio
    single = time_config(scene, cam, stream, 10, u8)

    n_distinct = len(set(devices))
    ms_per_step = elapsed / max(1, args.steps) * 1e3
    achieved_tf = flops / (ms_per_step * 1e-3) / 1e12
    headline = (args.config == "c3" and not args.scene and (W, H) == (3840, 2160))
    line = {
        "metric": "Mrays/s, 3840x2160 chapter_11 scene" if headline else f"Mrays/s, {W}x{H} frame of {name}",
        "value": round(rays_frame * args.steps / elapsed / 1e6, 2),
        "unit": "Mrays/s",
        "n_gpus": n_distinct, "workers": len(devices), "devices": devices,
        "steps": args.steps, "warmup": args.warmup, "prewarm_s": round(prewarm_s, 3),
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": data_note(args, name),
        "config": {"workload": f"{name} at {W}x{H}, depth 5; ONE process, a worker thread per GPU, {info['n_units']} units of interleaved "
                               f"{info['band_rows']}-row stripes taken from a shared atomic cursor (two in flight per GPU), f64 colours -> "
                               + ("8-bit frame (canvas.rs:123-130 applied by the render kernel) " if u8 else "f64 frame ")
                               + f"assembled on GPU {first} ({info['gather']} collection)",
                   "path": "single process, rt_multi_render_device",
                   "rays_per_frame": rays_frame, "rays_per_pixel": round(rays_frame / (W * H), 3)},
        "gather": info["gather"], "rccl_ranks": info["rccl_ranks"], "gather_note": info["note"],
        "gather_calibration_ms": calibration or None,
        "units": info["n_units"], "stripe_rows": info["band_rows"], "units_taken": taken,
        "worker_render_ms": [round(v, 4) for v in worker_ms],
        "render_ms": round(sum(render_ms) / max(1, len(render_ms)), 4),
        "gather_ms": round(sum(gather_ms) / max(1, len(gather_ms)), 4),
        "multi_create_ms": round(create_ms, 1),
        "single_gpu_same_run": {"ms_per_frame": single["ms_per_frame"], "kernel_ms": single["kernel_ms"], "mrays_per_s": single["mrays_per_s"],
                                "note": f"rt_render_device on GPU {first} alone, same frame, same process"},
        "roofline": {"bound": "valu_f64", "achieved": round(achieved_tf, 3), "peak": FP64_VECTOR_PEAK_TFLOPS * n_distinct,
                     "unit": "TFLOP/s", "frac": round(achieved_tf / (FP64_VECTOR_PEAK_TFLOPS * n_distinct), 4),
                     "algorithmic_flops_per_frame": flops, "traffic": None,
                     "note": "whole job: the frame's algorithmic flops over the wall-clock time per frame (bands of several "
                             "launches per GPU, collection included), against the FP64 vector peak of the GPUs used"},
        "cpu_baseline": None,
    }
    if os.environ.get("RT_BENCH_NOTE"):
        line["note"] = os.environ["RT_BENCH_NOTE"] + "; this line is the repeat with peer copies"
    print(json.dumps(line), flush=True)
    multi.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=25,
                    help="untimed frames first; the GPU takes about 20 frames of this size to reach its steady clocks")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3",
                    help="benchmark configuration of BASELINE.json: c3 (default) is the headline, chapter_11 at 3840x2160")
    ap.add_argument("--scene", default=None, help="a scene file instead of --config")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip other_configs, f64_frame and camera_render_ms")
    ap.add_argument("--check", action="store_true",
                    help="after the timed steps, rank 0 renders the whole frame by itself and compares the collected frame with it")
    ap.add_argument("--frame", choices=["f64", "u8"], default="u8",
                    help="what rank 0 collects: the 8-bit frame a PPM is written from, quantised on each GPU "
                         "(canvas.rs:123-130) before the gather (default), or the f64 canvas itself (8x the bytes)")
    ap.add_argument("--multi-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if not launched and (args.gpus > 1 or os.environ.get("RT_BENCH_DEVICES")):
        # one process for all GPUs; the measuring child is started by a parent that never touches a GPU itself
        if args.multi_child:
            run_multi_child(args)
        else:
            run_multi_parent(args)
        return

    if os.environ.get("RT_BENCH_WATCHDOG"):  # developer aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    # RT_BENCH_FORCE_DIST=1: run the several-ranks path (process group, band buffers, gather on its own stream) even
    # with one rank -- the way to exercise that path with the nccl backend on a box that has one GPU
    multi = world_size > 1 or bool(os.environ.get("RT_BENCH_FORCE_DIST"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    # Rehearsal on a box with fewer GPUs than ranks (RT_BENCH_BACKEND=gloo): ranks share devices and the gather goes
    # through gloo (staged through host memory: gloo is a CPU backend); the real launch is one rank per GPU over nccl.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    reduce_dev = dev if backend == "nccl" else torch.device("cpu")  # where tensors for all_reduce live
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    sf, scene_path, name, width, height = workload_of(args)
    cam = sf.build_camera(width, height)
    world = sf.build_world()
    t_create = time.perf_counter()
    scene = world.scene(device_index)
    scene_create_ms = (time.perf_counter() - t_create) * 1e3
    W, H = cam.hsize, cam.vsize
    # device buffers: my bands (padded to equal size so the gather is uniform) and, on rank 0, the frame
    u8 = args.frame == "u8"
    bg = BandGather(W, H, rank, world_size, dev, BAND_ROWS, dtype=torch.uint8 if u8 else torch.float64,
                    collective_even_alone=multi)
    stream = torch.cuda.current_stream().cuda_stream

    def render_mine(k: int = 0, stats: bool = False, timed: bool = False):
        """Frame k: this rank's bands, rendered (and quantised) into the buffer the gather of frame k reads.
        `timed`: the library records HIP events on the launch's stream before the tile-ordering pass, between it and
        the render kernel and after the render kernel (no synchronisation); read back after the timed region."""
        target = bg.frame if not multi else bg.buffer(k)
        if multi:
            bg.begin_render(k)
        # with 8-bit collection the render kernel quantises its own output (RT_RENDER_QUANTIZED)
        if not multi:
            st = scene.render_device(cam, target.data_ptr(), None, None, stream, stats, timed, u8)
        else:
            st = scene.render_device_bands(cam, target.data_ptr(), BAND_ROWS, rank, world_size, None, stream, stats, timed, u8)
        return st

    def step(k: int):
        render_mine(k)
        if multi:
            bg.gather(k)

    # rays of the frame, counted once by the counting kernel (untimed)
    st = render_mine(0, stats=True)
    counts = st.as_dict()
    flat = []
    for k in rt.Stats.COUNT_FIELDS:
        v = counts[k]
        flat.extend(v if isinstance(v, list) else [v])
    tot = torch.tensor(flat, dtype=torch.float64, device=reduce_dev)
    if multi:
        dist.all_reduce(tot)
    tot = [int(v) for v in tot.tolist()]
    total_counts, i = {}, 0
    for k in rt.Stats.COUNT_FIELDS:
        if isinstance(counts[k], list):
            total_counts[k] = tot[i:i + 5]
            i += 5
        else:
            total_counts[k] = tot[i]
            i += 1
    rays_frame = total_counts["rays"]
    my_flops = algorithmic_flops(counts)

    # The GPU takes ~40 ms of continuous work to reach its steady clocks (the first frames of a run are 15 % slower):
    # whatever --warmup says, frames are rendered for a tenth of a second before anything is timed.  On one GPU that
    # happens here; with several ranks after the --warmup steps, as whole steps.
    prewarm_s = 0.0
    if not multi:
        t_pre = time.perf_counter()
        k_pre = 0
        while time.perf_counter() - t_pre < 0.1:
            render_mine(k_pre)
            k_pre += 1
            if k_pre % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        prewarm_s = time.perf_counter() - t_pre
    for k in range(args.warmup):
        step(k)
    if multi:
        # the same warm-up for several ranks, as whole steps (render and gather, exactly what the timed loop does);
        # the ranks agree on when a tenth of a second has passed, so that all of them stop after the same step
        t_pre = time.perf_counter()
        k = args.warmup
        while True:
            for _ in range(16):
                step(k)
                k += 1
            bg.finish()
            torch.cuda.synchronize()
            spent = torch.tensor([time.perf_counter() - t_pre], dtype=torch.float64, device=reduce_dev)
            dist.all_reduce(spent, op=dist.ReduceOp.MAX)
            if spent.item() >= 0.1:
                prewarm_s = spent.item()
                break
    bg.finish()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        render_mine(k, timed=True)
        if multi:
            bg.gather(k)  # on its own stream: overlaps the rendering of frame k+1
    bg.finish()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the render kernel's own duration (and the ordering pass before it): the last launches of the timed region
    # (the handle keeps the events of 32)
    times = [scene.render_times(back) for back in range(min(args.steps, 32))]
    ordering_ms = sum(t[0] for t in times) / max(1, len(times))
    kernel_ms = sum(t[1] for t in times) / max(1, len(times))
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=reduce_dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = t.tolist()

    if args.check and rank == 0:
        whole = torch.zeros(H * W * 3, dtype=torch.float64, device=dev)
        scene.render_device(cam, whole.data_ptr(), None, None, stream, False)
        if u8:
            q = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
            rt.quantize_device(whole.data_ptr(), whole.numel(), q.data_ptr(), stream)
            whole = q
        torch.cuda.synchronize()
        same = torch.equal(bg.frame_rows().reshape(-1), whole)
        print(f"check: collected frame {'equals' if same else 'DIFFERS FROM'} the single-GPU frame", file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(3)

    if rank == 0:
        ms_per_step = elapsed / max(1, args.steps) * 1e3
        achieved_tf = my_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        headline = (args.config == "c3" and not args.scene and (W, H) == (3840, 2160))
        line = {
            "metric": "Mrays/s, 3840x2160 chapter_11 scene" if headline else f"Mrays/s, {W}x{H} frame of {name}",
            "value": round(rays_frame * args.steps / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "prewarm_s": round(prewarm_s, 3),  # frames rendered before the --warmup frames, until the GPU's clocks have settled
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": data_note(args, name),
            "config": {"workload": (f"{name} at {W}x{H}, depth 5, {BAND_ROWS}-row bands round-robin over {world_size} ranks "
                                    f"(one process per GPU, torch.distributed), f64 colours -> " if multi else
                                    f"{name} at {W}x{H}, depth 5, whole frame in one launch on one GPU, f64 colours -> ")
                                   + ("8-bit frame (canvas.rs:123-130 applied by the render kernel) " if u8 else "f64 frame ")
                                   + ("collected on rank 0 with one RCCL gather per frame" if multi else "in HBM"),
                       "path": "one process per GPU, rt_render_device + torch.distributed.gather" if multi else "single GPU, rt_render_device",
                       "rays_per_frame": rays_frame, "rays_per_pixel": round(rays_frame / (W * H), 3),
                       "kernel_ms_max_over_ranks": round(kernel_ms_max, 4)},
            "roofline": {"bound": "valu_f64", "achieved": round(achieved_tf, 3), "peak": FP64_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved_tf / FP64_VECTOR_PEAK_TFLOPS, 4),
                         "frac_of_unfused_ceiling": round(2.0 * achieved_tf / FP64_VECTOR_PEAK_TFLOPS, 4),
                         "kernel": kernel_name(scene), "kernel_ms": round(kernel_ms, 4),
                         "tile_ordering_ms": round(ordering_ms, 4),
                         "launches_averaged": len(times),
                         "algorithmic_flops_per_launch": my_flops,
                         "note": "FP64 vector ALU is the bounding unit (no dense contraction, scene cache-resident); "
                                 "peak counts an FMA as 2 flops, the kernel is built without FMA contraction to keep "
                                 "the reference's rounding, so its own ceiling is half of peak; `achieved` and `frac` price "
                                 "the REFERENCE's work for the same rays (every ray against every object, 33 flops per "
                                 "ray transform), as the roofline contract asks; the kernel executes 9 per "
                                 "scale-and-translate inverse and skips a sphere or plane that no ray of a wave can hit, ends a shadow walk when every ray of the wave has its occluder, and answers the shadow query of a hit from inside a sphere by proof (the sphere's own shadow) instead of a walk: "
                                 "executed_flops (PMC) is what it issued and frac_executed prices that"},
        }
        # the committed PMC figures are per launch of the WHOLE frame: they describe a one-GPU run only
        line["roofline"].update(committed_pmc(args.config) if headline and u8 and world_size == 1 else {"traffic": None})
        if line["roofline"].get("executed_flops") and kernel_ms > 0:
            ex_tf = line["roofline"]["executed_flops"] / (kernel_ms * 1e-3) / 1e12
            line["roofline"]["executed_tflops"] = round(ex_tf, 3)
            line["roofline"]["frac_executed"] = round(ex_tf / FP64_VECTOR_PEAK_TFLOPS, 4)
        if world_size == 1 and not multi:
            line["scene_create_ms"] = round(scene_create_ms, 2)  # flatten + upload (rt_scene_create), outside ms/frame
            peak = measured_fp64_peak()
            if peak:
                line["roofline"]["peak_measured_fma"] = round(peak["fma"], 2)
                line["roofline"]["peak_measured_unfused"] = round(peak["unfused"], 2)
                line["roofline"]["frac_reference_work_of_measured_fma"] = round(achieved_tf / peak["fma"], 4)
                line["roofline"]["frac_reference_work_of_measured_unfused"] = round(achieved_tf / peak["unfused"], 4)
                if line["roofline"].get("executed_tflops"):
                    line["roofline"]["frac_executed_of_measured_unfused"] = round(line["roofline"]["executed_tflops"] / peak["unfused"], 4)
            if not args.no_others:
                line["ppm_encode_ms"] = ppm_encode_ms(scene, cam)  # Canvas::to_ppm_string on the host, outside ms/frame
                # the same frame through the several-GPU entry point with one worker on this GPU: what that path costs
                # beyond rt_render_device (one band = the whole frame; the call returns when the frame is complete)
                m1 = world.multi_scene([device_index])
                probe = torch.empty(W * H * 3, dtype=torch.uint8 if u8 else torch.float64, device=dev)
                for _ in range(5):
                    m1.render_device(cam, probe.data_ptr(), quantized=u8)
                t1 = time.perf_counter()
                n_m1 = max(10, args.steps)
                for _ in range(n_m1):
                    m1.render_device(cam, probe.data_ptr(), quantized=u8)
                line["multi_single_gpu_ms"] = round((time.perf_counter() - t1) / n_m1 * 1e3, 4)
                m1.close()
                del probe
                f64 = time_config(scene, cam, stream, 20, False)
                line["f64_frame"] = {"ms_per_step": f64["ms_per_frame"], "value": f64["mrays_per_s"], "kernel_ms": f64["kernel_ms"],
                                     "note": "the same frame stored as the f64 canvas (Canvas.pixels, 24 B per pixel) in HBM"}
                # two frames in flight: a second handle of the same world on a second stream renders frame k+1 while
                # frame k drains, so the under-occupied start and end of consecutive launches overlap (what a caller
                # rendering a sequence of frames would do; the headline above is one frame at a time)
                world2 = sf.build_world()
                scene2 = world2.scene(device_index)
                side = torch.cuda.Stream(device=dev)
                frames = [torch.empty(W * H * 3, dtype=torch.uint8, device=dev) for _ in range(2)]
                handles, streams = [scene, scene2], [stream, side.cuda_stream]
                for k in range(8):
                    handles[k % 2].render_device(cam, frames[k % 2].data_ptr(), None, None, streams[k % 2], False, False, True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_pipe = max(8, args.steps)
                for k in range(n_pipe):
                    handles[k % 2].render_device(cam, frames[k % 2].data_ptr(), None, None, streams[k % 2], False, False, True)
                torch.cuda.synchronize()
                pipe_ms = (time.perf_counter() - t1) / n_pipe * 1e3
                line["two_frames_in_flight"] = {"ms_per_frame": round(pipe_ms, 4), "value": round(rays_frame / pipe_ms / 1e3, 1),
                                                "note": "frames alternate between two handles of the world on two streams (8-bit frames in HBM)"}
                scene2.close()
                del frames
                # the drop-in call: Camera::render into a host buffer, PCIe transfer and host copy included
                host = np.zeros((H, W, 3))
                for _ in range(3):
                    scene.render(cam, out=host)
                ts = []
                for _ in range(10):
                    t1 = time.perf_counter()
                    scene.render(cam, out=host)
                    ts.append((time.perf_counter() - t1) * 1e3)
                line["camera_render_ms"] = round(sorted(ts)[len(ts) // 2], 3)
                line["camera_render_mrays_per_s"] = round(rays_frame / sorted(ts)[len(ts) // 2] / 1e3, 1)
                del host
                others = []
                for cfg in ("c2", "c3", "c4", "c5"):
                    if cfg == args.config and not args.scene and (W, H) == CONFIGS[cfg][1:]:
                        continue
                    osf, _ = scene_file_of(cfg)
                    oworld = osf.build_world()
                    t1 = time.perf_counter()
                    oscene = oworld.scene(device_index)
                    ocreate_ms = (time.perf_counter() - t1) * 1e3
                    ocam = osf.build_camera(CONFIGS[cfg][1], CONFIGS[cfg][2])
                    r = time_config(oscene, ocam, stream, 20 if cfg != "c5" else 5, True, cfg, peak)
                    r["config"] = cfg
                    r["workload"] = f"{CONFIGS[cfg][0]} at {CONFIGS[cfg][1]}x{CONFIGS[cfg][2]}, depth 5, 8-bit frame in HBM"
                    r["scene_create_ms"] = round(ocreate_ms, 2)
                    r["ppm_encode_ms"] = ppm_encode_ms(oscene, ocam)
                    if cfg == "c4":  # the drop-in call on a world rendered level by level (four row groups = four views)
                        host = np.zeros((CONFIGS[cfg][2], CONFIGS[cfg][1], 3))
                        ts = []
                        for k in range(9):
                            t1 = time.perf_counter()
                            oscene.render(ocam, out=host)
                            ts.append((time.perf_counter() - t1) * 1e3)
                        r["camera_render_ms"] = round(sorted(ts[3:])[3], 3)
                        del host
                    others.append(r)
                    oscene.close()
                line["other_configs"] = others
        if not args.no_cpu_baseline and world_size == 1:
            text = None
            if not scene_path:
                from raytracer_rs_amd import synthetic

                text = synthetic.large_scene()
            line["cpu_baseline"] = cpu_baseline(scene_path, W, H, scene_text=text)
        elif world_size == 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
