#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in a `hipcc -S` listing, per innermost loop:
    python tools/isa_loops.py /tmp/rt_device.s 'render_kernelILb0ELi1ELb1'"""
import collections
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(needle) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], None
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
    if m:
        cur = {"name": m.group(1), "hdr": m.group(2) or "", "ins": []}
        blocks.append(cur)
        continue
    if l.startswith("; %bb."):
        cur = {"name": l.split()[1], "hdr": l, "ins": []}
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur is None:
        cur = {"name": "entry", "hdr": "", "ins": []}
        blocks.append(cur)
    cur["ins"].append(t.split()[0])


def klass(op):
    if op.startswith("v_") and ("f64" in op) and not op.startswith(("v_cmp", "v_cndmask", "v_mov")):
        return "f64"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "vmov"
    if op.startswith("v_cndmask"):
        return "sel"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "br"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    return "salu"


tot = collections.Counter()
for b in blocks:
    c = collections.Counter(klass(o) for o in b["ins"])
    tot.update(c)
    depth = re.search(r"Depth=(\d+)", b["hdr"])
    print(f"{b['name']:12s} d={depth.group(1) if depth else '-'} n={len(b['ins']):4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("TOTAL", dict(tot))
