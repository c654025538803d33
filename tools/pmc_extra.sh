#!/bin/bash
# Extra counter passes (latency levels, instruction/scalar cache) over bench.py; CSVs under gpurun_out/prof_<tag>/.
TAG=${1:-x}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
    local name=$1
    shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/$name.bench.json" 2> "$OUT/$name.err"
    echo "pass $name exit=$?" >&2
}
run lvl1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES || exit 1
run lvl2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_BUSY_CYCLES || exit 1
run sqc1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ || exit 1
run sqc2 SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_INST_REQ || exit 1
