#!/bin/bash
# rocprofv3 passes over one configuration of tests/devtools/perf.py (GPU box).
#   tools/profile_cfg.sh <tag> <config> [WxH]        e.g.  tools/profile_cfg.sh r02_c4 c4 3840x2160
# Writes CSVs under gpurun_out/prof_<tag>/ ; tools/summarize_cfg.py turns them into profiles/<tag>_summary.md.
# One kernel-trace pass, then one run per PMC group (counters are never combined with tracing, and each group stays
# within the per-block slot limits: 8 SQ, 4 TCC where FETCH_SIZE takes 3 and WRITE_SIZE 2, 2 GRBM).
TAG=${1:?tag}
CFG=${2:?config}
SIZE=${3:-}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=(--configs "$CFG" --no-check)
[ -n "$SIZE" ] && ARGS+=(--size "$SIZE")

run() { # name, timeout, reps, rocprof args...
    local name=$1 to=$2 reps=$3
    shift 3
    echo "== pass $name" >&2
    timeout -s KILL "$to" rocprofv3 "$@" --output-format csv -d "$OUT" -o "$name" -- python3 tests/devtools/perf.py \
        "${ARGS[@]}" --reps "$reps" > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    echo "   exit=$rc $(tail -n 1 "$OUT/$name.out")" >&2
    return $rc
}

run trace 300 10 --kernel-trace --stats || exit 1
run pmc_sq1 240 2 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run pmc_sq2 240 2 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
run pmc_sq3 240 2 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_CYCLES || exit 1
run pmc_sq4 240 2 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM || exit 1
run pmc_fetch 240 2 --pmc FETCH_SIZE || exit 1
run pmc_write 240 2 --pmc WRITE_SIZE || exit 1
ls "$OUT" >&2
