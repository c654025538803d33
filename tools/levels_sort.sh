#!/bin/bash
# Level renderer: the next level's rays sorted within their bins by the leaf they leave (RT_LEVELS_SORT), steady-state frame times.
for cfg in "c4 3840x2160" "c5 7680x4320"; do
  set -- $cfg
  for mode in ${MODES:-0 1 2 3}; do
    echo "== RT_LEVELS_SORT=$mode"
    RT_LEVELS_SORT=$mode timeout -k 10 200 python3 tests/devtools/perf.py --configs $1 --no-check --size $2 --reps 8 || exit 1
  done
done
