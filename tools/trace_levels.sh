#!/bin/bash
export TMPDIR=/tmp
for cfg in "c4 3840x2160" "c5 7680x4320"; do
  set -- $cfg
  OUT=gpurun_out/trace_$1
  mkdir -p $OUT
  timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tests/devtools/perf.py --configs $1 --no-check --size $2 --reps 6 > $OUT/out.txt 2> $OUT/err.txt || exit 1
  tail -1 $OUT/out.txt
done
