#!/bin/bash
# Level renderer: every level's rays left in parent-wave order (RT_LEVELS_SORT=0) or sorted by kind, octant and leaf (1); frame times of a repeated view.
for cfg in "c4 3840x2160" "c5 7680x4320"; do
  set -- $cfg
  for mode in ${MODES:-0 1}; do
    echo "== RT_LEVELS_SORT=$mode"
    RT_LEVELS_SORT=$mode timeout -k 10 200 python3 tests/devtools/perf.py --configs $1 --no-check --size $2 --reps 8 || exit 1
  done
done
