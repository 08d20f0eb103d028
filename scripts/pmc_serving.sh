#!/bin/bash
# HBM traffic of the single-query scans (separate --pmc passes, kernel trace only): the serving-shaped scan claims the HBM
# roofline, so its FETCH_SIZE per launch must be the size of the scan copy.  Usage on the GPU box: bash scripts/pmc_serving.sh
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_serving
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kind in sift marco; do
  for pass in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$kind -o ${kind}_$pass -- \
        python3 $R/scripts/latency_serving.py --kind $kind --nq 1 > $OUT/${kind}_$pass.log 2>&1 || echo "pass $kind $pass failed"
  done
  python3 $R/scripts/pmc_summarize.py $OUT/$kind scan_i8_kernel scan16_kloop_kernel > $OUT/${kind}_summary.txt
done
cat $OUT/*_summary.txt
