#!/bin/bash
# One bench leg under rocprofv3 --kernel-trace --stats + its per-search breakdown (on the GPU box, from the repo root):
#   bash scripts/prof_leg.sh <out-name> <breakdown-filter-or-''> <bench args...>
# Writes gpurun_out/<out-name>/{bench.json, kernel_stats.csv, breakdown.txt}.
set -u
R=$GRAFT_REPO_ROOT
NAME=$1; FILTER=$2; shift 2
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
    python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || echo "rocprof failed"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 $R/scripts/trace_breakdown.py $t $FILTER > $OUT/breakdown.txt 2>&1
rm -rf $OUT/trace
cat $OUT/breakdown.txt
