#!/bin/bash
# A/B kernel timing on one GPU box: tools/abperf.sh "<perf.py args>" <libA.so> <libB.so> [...] -- runs
# tests/devtools/perf.py on each library, twice, interleaved, so that box-to-box and clock differences cancel.
ARGS=$1
shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    RT_AMD_LIB=$PWD/$lib timeout -k 10 300 python tests/devtools/perf.py $ARGS || exit 1
  done
done
