#!/bin/bash
# Single launch against level-by-level rendering of the same library, interleaved (RT_RENDER_LEVELS=0 / 1):
# chapter_14 at 4K, the generated scene at 1080p, 4K and 8K.
for round in 1 2; do
  for mode in 0 1; do
    echo "== RT_RENDER_LEVELS=$mode"
    RT_RENDER_LEVELS=$mode timeout -k 10 400 python tests/devtools/perf.py --configs c4,c5 --reps 5 || exit 1
    RT_RENDER_LEVELS=$mode timeout -k 10 400 python tests/devtools/perf.py --configs c5 --size 3840x2160 --no-check --reps 5 || exit 1
    RT_RENDER_LEVELS=$mode timeout -k 10 400 python tests/devtools/perf.py --configs c5 --size 7680x4320 --no-check --reps 4 || exit 1
  done
done
