#!/bin/bash
# tools/levels_c5.sh <libA.so> <libB.so> ...: the generated scene at 1080p and 8K on each library, twice, interleaved.
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tests/devtools/perf.py --configs c5 --no-check --reps 5 || exit 1
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tests/devtools/perf.py --configs c5 --size 7680x4320 --no-check --reps 4 || exit 1
  done
done
