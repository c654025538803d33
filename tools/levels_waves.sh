#!/bin/bash
# Occupancy of the level kernel: builds with RT_LEVEL_WAVES = 4, 5, 6 on the generated scene (levels by default) and
# chapter_14 (levels forced), interleaved.
for round in 1 2; do
  for lib in ab/lv4.so ab/lv5.so ab/lv6.so; do
    echo "== $lib"
    RT_AMD_LIB=$PWD/$lib RT_RENDER_LEVELS=1 timeout -k 10 300 python tests/devtools/perf.py --configs c4,c5 --no-check --reps 5 || exit 1
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tests/devtools/perf.py --configs c5 --size 7680x4320 --no-check --reps 4 || exit 1
  done
done
