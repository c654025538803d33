#!/bin/bash
# rocprofv3 passes over bench.py (GPU box).  Usage: tools/profile.sh <tag> [extra bench args...]
# Writes CSVs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.
# One kernel-trace pass, then one run per PMC group (counters are never combined with tracing, and each group stays
# within the per-block slot limits: 8 SQ, 4 TCC where FETCH_SIZE takes 3 and WRITE_SIZE 2, 2 GRBM).
TAG=${1:-r01}
shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
EXTRA=("$@")
STEPS=(--steps 50 --warmup 25)

run() { # name, timeout, rocprof args...
    local name=$1 to=$2
    shift 2
    echo "== pass $name" >&2
    timeout -s KILL "$to" rocprofv3 "$@" --output-format csv -d "$OUT" -o "$name" -- python3 bench.py --no-cpu-baseline --no-others \
        "${STEPS[@]}" "${EXTRA[@]}" > "$OUT/$name.bench.json" 2> "$OUT/$name.err"
    local rc=$?
    echo "   exit=$rc" >&2
    return $rc
}

run trace 300 --kernel-trace --stats || exit 1
STEPS=(--steps 3 --warmup 1)
run pmc_sq1 240 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY || exit 1
run pmc_sq2 240 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
run pmc_sq3 240 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_ANY SQ_CYCLES || exit 1
run pmc_fetch 240 --pmc FETCH_SIZE || exit 1
run pmc_write 240 --pmc WRITE_SIZE || exit 1
run pmc_grbm 240 --pmc GRBM_GUI_ACTIVE || exit 1
ls "$OUT" >&2
