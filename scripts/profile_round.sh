#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py commands (CSV summaries to copy into profiles/).
# Usage on the GPU box: bash scripts/profile_round.sh <tag> <name> <bench args...>   (one profiled command per call)
set -u
TAG=$1; NAME=$2; shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME -o $NAME -- \
    python3 $R/bench.py "$@" > $OUT/$NAME.json 2> $OUT/$NAME.err || echo "rocprof $NAME failed"
f=$(find $OUT/$NAME -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${NAME}_kernel_stats.csv && head -14 $OUT/${NAME}_kernel_stats.csv | cut -c1-150
