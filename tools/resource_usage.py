#!/usr/bin/env python3
"""Per-kernel registers, spills, scratch and occupancy of rt_device.hip as the compiler reports them:
    python tools/resource_usage.py [needle]      (runs `make -C raytracer_rs_amd/csrc resource-usage`)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needle = sys.argv[1] if len(sys.argv) > 1 else ""
text = subprocess.run(["make", "-C", os.path.join(ROOT, "raytracer_rs_amd", "csrc"), "resource-usage"], capture_output=True, text=True)
text = text.stdout + text.stderr
cur, row = None, {}


def flush():
    if cur and needle in cur:
        print(f"{cur:64s} SGPR {row.get('TotalSGPRs', '?'):>3} VGPR {row.get('VGPRs', '?'):>3} spill S {row.get('SGPRs Spill', '?'):>3} V {row.get('VGPRs Spill', '?'):>3} "
              f"scratch {row.get('ScratchSize [bytes/lane]', '?'):>4} B occupancy {row.get('Occupancy [waves/SIMD]', '?')} LDS {row.get('LDS Size [bytes/block]', '?')}")


for ln in text.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        flush()
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        row = {}
    else:
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", ln)
        if m and cur:
            row[m.group(1).strip()] = m.group(2)
flush()
if "error" in text:
    print(text[-3000:])
