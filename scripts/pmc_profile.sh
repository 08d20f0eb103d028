#!/bin/bash
# Collect PMC counters for the dominant kernel (separate passes; never combined with --sys-trace etc.).
# Usage (on the GPU box, from the repo root):  bash scripts/pmc_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -o $name -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc3 TCC_REQ_sum TCC_READ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
ls $OUT
