#!/bin/bash
# Two PMC passes (SQ busy / GRBM clock) for the dominant kernel.  Usage: bash scripts/pmc_quick.sh <tag> [bench args...]
set -u
TAG=${1:-q}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o $name -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
