#!/bin/bash
# tools/levels_libs.sh <libA.so> <libB.so> ...: the generated scene at 1080p, 4K and 8K (levels by default) and
# chapter_14 at 4K with levels forced, on each library, twice, interleaved.
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    RT_AMD_LIB=$PWD/$lib RT_RENDER_LEVELS=1 timeout -k 10 300 python tests/devtools/perf.py --configs c4,c5 --no-check --reps 5 || exit 1
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tests/devtools/perf.py --configs c5 --size 3840x2160 --no-check --reps 5 || exit 1
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tests/devtools/perf.py --configs c5 --size 7680x4320 --no-check --reps 4 || exit 1
  done
done
