#!/bin/bash
# Level renderer: waves launched in their own order (0) or by measured cost, expensive first (RT_LEVELS_ORDER=1).
for round in 1 2; do
for mode in 0 1; do
  echo "== RT_LEVELS_ORDER=$mode"
  RT_LEVELS_ORDER=$mode timeout -k 10 200 python3 tests/devtools/perf.py --configs c4 --no-check --reps 8 || exit 1
  RT_LEVELS_ORDER=$mode timeout -k 10 200 python3 tests/devtools/perf.py --configs c5 --no-check --reps 8 || exit 1
  RT_LEVELS_ORDER=$mode timeout -k 10 200 python3 tests/devtools/perf.py --configs c5 --no-check --size 7680x4320 --reps 6 || exit 1
done
done
