#!/bin/bash
# A/B timing on one GPU box: tools/ab.sh <libA.so> <libB.so> [...]  -- runs bench.py (kernel only) on each library,
# twice, interleaved, so that box-to-box and clock differences cancel.
for round in 1 2; do
  for lib in "$@"; do
    out=$(RT_AMD_LIB=$PWD/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5) || exit 1
    echo "$lib $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel_ms"], "ms kernel,", d["value"], d["unit"])')"
  done
done
