#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (written by tools/profile_cfg.sh) into profiles/<tag>_summary.md and
profiles/<tag>_kernel_stats.csv.

    python tools/summarize_cfg.py <tag> [kernel substring] [--flops F] [--resource-usage file]

--flops: algorithmic flops of one launch (bench.py's table over the counting kernel's counters), for the fraction of
the FP64 vector peak.  --resource-usage: output of `make -C raytracer_rs_amd/csrc resource-usage`, from which the lines
of the profiled kernel are appended, so that registers, spills, scratch and occupancy can be read from profiles/ alone.
"""
import argparse
import csv
import os
import re
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP64_PEAK_TF = 78.6


def short_name(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("needle", nargs="?", default="render_kernel<false")
    ap.add_argument("--flops", type=float, default=0.0, help="default: the figure perf.py printed in the kernel-trace run")
    ap.add_argument("--resource-usage", default=None)
    ap.add_argument("--note", default="")
    ap.add_argument("--frames", type=int, default=0, help="the needle's kernel is launched several times per frame (the level renderer): "
                    "figures are summed over all its launches and divided by this number of frames (perf.py's --reps of the pass)")
    ap.add_argument("--config", default=None, help="also record HBM bytes and executed flops under this name in profiles/pmc_traffic.json")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", f"prof_{a.tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(src, "trace.out")) as f:
        perf_line = f.read().strip().splitlines()[-1]
    lines = [f"# rocprofv3 summary `{a.tag}`", "",
             "Command per pass: `rocprofv3 <pass args> --output-format csv -- python3 tests/devtools/perf.py --configs <cfg> "
             "--size WxH --no-check --reps R` (tools/profile_cfg.sh); kernel-trace pass R=10, counter passes R=2.", "",
             f"perf.py in the kernel-trace run: `{perf_line}`", ""]
    if a.note:
        lines += [a.note, ""]
    m = re.search(r"flops ([0-9.e+]+)", perf_line)
    if m and not a.flops:
        a.flops = float(m.group(1))

    stats = os.path.join(src, "trace_kernel_stats.csv")
    shutil.copyfile(stats, os.path.join(dst, f"{a.tag}_kernel_stats.csv"))
    lines += ["## Kernel time (`--kernel-trace --stats`)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    avg_ms = None
    with open(stats) as f:
        for row in csv.DictReader(f):
            name = row["Name"]
            lines.append(f"| `{short_name(name)}` | {row['Calls']} | {float(row['AverageNs']) / 1e6:.4f} | {float(row['MinNs']) / 1e6:.4f} | "
                         f"{float(row['MaxNs']) / 1e6:.4f} | {float(row['Percentage']):.2f} |")
            if a.needle in name and avg_ms is None and not a.frames:
                avg_ms = float(row["AverageNs"]) / 1e6
            if a.needle in name and a.frames:
                avg_ms = (avg_ms or 0.0) + float(row["TotalDurationNs"]) / 1e6 / a.frames
    lines.append("")
    if a.frames:
        lines += [f"`{a.needle}` is launched once per level of the ray trees: summed over its launches, {avg_ms:.4f} ms per frame "
                  f"(the kernel-trace pass rendered {a.frames} frames; every figure below is per frame likewise).", ""]
    kernel_full = None
    with open(os.path.join(src, "trace_kernel_trace.csv")) as f:
        for row in csv.DictReader(f):
            if a.needle in row["Kernel_Name"]:
                kernel_full = row["Kernel_Name"]
                lines += [f"Dispatch of `{short_name(kernel_full)}`: grid {row['Grid_Size_X']} work-items, workgroup {row['Workgroup_Size_X']}; "
                          f"as rocprofv3 reports them: VGPR_Count {row['VGPR_Count']}, Accum_VGPR_Count {row['Accum_VGPR_Count']}, "
                          f"SGPR_Count {row['SGPR_Count']}, Scratch_Size {row['Scratch_Size']} B per lane, LDS_Block_Size {row['LDS_Block_Size']} B "
                          f"(static LDS only: the ray stack's LDS is dynamic and is not in this figure; the compiler's own "
                          f"numbers are in the resource-usage section below).", ""]
                break
    if a.flops and avg_ms:
        tf = a.flops / (avg_ms * 1e-3) / 1e12
        lines += [f"Algorithmic flops per launch {a.flops:.4g} / {avg_ms:.4f} ms = {tf:.2f} TFLOP/s = **{tf / FP64_PEAK_TF:.4f} of the "
                  f"FP64 vector peak** ({FP64_PEAK_TF} TFLOP/s, FMA counted as 2; {2 * tf / FP64_PEAK_TF:.4f} of the ceiling of code built without contraction).", ""]

    counters = defaultdict(list)
    for fn in sorted(os.listdir(src)):
        if not fn.endswith("_counter_collection.csv"):
            continue
        with open(os.path.join(src, fn)) as f:
            per_dispatch = defaultdict(dict)
            for row in csv.DictReader(f):
                if a.needle in row["Kernel_Name"]:
                    d = per_dispatch[row["Dispatch_Id"]]
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for d in per_dispatch.values():
                for k, v in d.items():
                    counters[k].append(v)
    if a.frames:
        # counter passes run fewer frames than the kernel-trace pass: launches per frame x frames of that pass
        with open(os.path.join(src, "trace_kernel_trace.csv")) as f:
            per_frame = sum(1 for r in csv.DictReader(f) if a.needle in r["Kernel_Name"]) / a.frames
        mean = {k: sum(v) / (len(v) / per_frame) for k, v in counters.items()}
    else:
        mean = {k: sum(v) / len(v) for k, v in counters.items()}
    lines += [f"## Counters, mean per {'frame' if a.frames else 'launch'} of `{a.needle}` (`--pmc`, one pass per group)", "", "| counter | value |", "|---|---|"]
    for k in sorted(mean):
        lines.append(f"| {k} | {mean[k]:.6g} |")
    lines.append("")
    derived = []
    g = mean.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        derived.append(f"VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)) = "
                       f"{100.0 * mean['SQ_THREAD_CYCLES_VALU'] / (mean['SQ_ACTIVE_INST_VALU'] * 64.0):.1f} %")
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD issues VALU work, summed over the SIMDs of every SE-level SQ
        # instance; SQ_BUSY_CYCLES is per SQ instance: 4 SIMDs x (CUs per instance) share one busy counter.
        derived.append(f"SQ_ACTIVE_INST_VALU x 4 / SQ_WAVE_CYCLES = {4 * mean['SQ_ACTIVE_INST_VALU'] / mean['SQ_WAVE_CYCLES']:.3f} "
                       f"(share of a resident wave's cycles in which it issues VALU work)")
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in mean:
                derived.append(f"{k} / SQ_WAVE_CYCLES = {mean[k] / mean['SQ_WAVE_CYCLES']:.3f}")
    if g("SQ_INSTS_VALU"):
        f64 = [g(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
        if sum(f64):
            derived.append(f"f64 VALU wave-instructions per launch: add {f64[0]:.4g}, mul {f64[1]:.4g}, fma {f64[2]:.4g}, trans {f64[3]:.4g} "
                           f"= {100 * sum(f64) / mean['SQ_INSTS_VALU']:.1f} % of all VALU instructions; x64 lanes -> executed f64 flops (fma = 2) "
                           f"<= {(f64[0] + f64[1] + 2 * f64[2] + f64[3]) * 64:.4g}")
        derived.append("instruction mix per launch: VALU {:.4g}, SALU {:.4g}, SMEM {:.4g}, VMEM rd {:.4g}, VMEM wr {:.4g}, LDS {:.4g}, branch {:.4g}".format(
            g("SQ_INSTS_VALU", 0), g("SQ_INSTS_SALU", 0), g("SQ_INSTS_SMEM", 0), g("SQ_INSTS_VMEM_RD", 0), g("SQ_INSTS_VMEM_WR", 0),
            g("SQ_INSTS_LDS", 0), g("SQ_INSTS_BRANCH", 0)))
    if g("SQC_DCACHE_REQ"):
        derived.append(f"scalar data cache: {g('SQC_DCACHE_REQ'):.4g} requests, hits {100 * g('SQC_DCACHE_HITS', 0) / g('SQC_DCACHE_REQ'):.1f} %, "
                       f"misses {100 * g('SQC_DCACHE_MISSES', 0) / g('SQC_DCACHE_REQ'):.1f} %")
    if g("SQ_INST_LEVEL_SMEM") and g("SQ_INSTS_SMEM"):
        derived.append(f"mean scalar-load latency (SQ_INST_LEVEL_SMEM / SQ_INSTS_SMEM) = {g('SQ_INST_LEVEL_SMEM') / g('SQ_INSTS_SMEM'):.0f} cycles")
    if "FETCH_SIZE" in mean or "WRITE_SIZE" in mean:
        fetch_b = g("FETCH_SIZE", 0.0) * 1024.0
        write_b = g("WRITE_SIZE", 0.0) * 1024.0
        derived.append(f"HBM traffic per launch ~ {(2 * fetch_b + write_b) / 1e6:.1f} MB: fetch {2 * fetch_b / 1e6:.1f} MB (FETCH_SIZE KB x 1024 x 2, "
                       f"the gfx950 correction for wide reads), write {write_b / 1e6:.1f} MB (WRITE_SIZE KB x 1024)")
    lines += ["## Derived", ""] + [f"* {d}" for d in derived] + [""]
    if a.config and ("FETCH_SIZE" in mean or "WRITE_SIZE" in mean):
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from summarize_profile import merge_pmc_entry

        entry = {"fetch_kb_reported": g("FETCH_SIZE"), "write_kb_reported": g("WRITE_SIZE"),
                 "hbm_bytes_per_launch": 2.0 * g("FETCH_SIZE", 0.0) * 1024.0 + g("WRITE_SIZE", 0.0) * 1024.0, "profile": f"{a.tag}_summary.md"}
        f64 = [g(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")]
        if sum(f64):
            entry["executed_f64_flops_per_launch"] = (f64[0] + f64[1] + 2 * f64[2] + f64[3]) * 64
        merge_pmc_entry(dst, a.config, entry)

    if a.resource_usage and kernel_full:
        import subprocess

        want = short_name(kernel_full)
        with open(a.resource_usage) as f:
            text = f.read()
        block, keep = [], False
        for ln in text.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", ln)
            if m:
                demangled = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                keep = short_name(demangled) == want
                if keep:
                    block.append(short_name(demangled))
                continue
            if keep and "remark:" in ln:
                block.append(ln.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").rstrip())
        lines += ["## Compiler resource usage (`-Rpass-analysis=kernel-resource-usage`)", "", "```"] + block + ["```", ""]

    with open(os.path.join(dst, f"{a.tag}_summary.md"), "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
