#!/bin/bash
# Scalar-cache counters and scalar-load latency of one configuration of tests/devtools/perf.py:
#   tools/pmc_scalar.sh <tag> <config>      (CSV under gpurun_out/prof_<tag>/)
TAG=${1:-x}
CFG=${2:-c4}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT" -o scalar -- python3 tests/devtools/perf.py --configs "$CFG" --no-check --reps 2 > "$OUT/scalar.out" 2> "$OUT/scalar.err"
echo "exit=$?"
