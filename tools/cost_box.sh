#!/bin/bash
# The plan's price of a range-skip test (RT_COST_BOX, vector instructions; 40 shipped) against frame time, final kernels.
for cb in 40 20 30 60 40; do
  echo "== RT_COST_BOX=$cb"
  RT_COST_BOX=$cb timeout -k 10 200 python3 tests/devtools/perf.py --configs c4 --no-check --reps 8 || exit 1
  RT_COST_BOX=$cb timeout -k 10 200 python3 tests/devtools/perf.py --configs c5 --no-check --size 3840x2160 --reps 6 || exit 1
done
