#!/bin/bash
# End-of-round evidence: bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, the
# config-5 shard, and BASELINE configs 1-4.  Usage on the GPU box: bash scripts/final_profiles.sh <tag>
set -u
TAG=${1:-r01j}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 3 > $OUT/bench_sift1m.json 2> $OUT/bench_sift1m.err || echo "bench failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sift1m -o sift1m -- \
    python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_sift1m_rocprof.json 2> $OUT/stats_sift1m.err || echo "rocprof sift1m failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_marco -o marco -- \
    python3 $R/bench.py --workload marco12.5m --steps 5 --warmup 2 > $OUT/bench_marco12m.json 2> $OUT/stats_marco.err || echo "rocprof marco failed"
timeout -k 10 600 python3 $R/scripts/bench_configs.py --out $OUT/configs.jsonl > $OUT/configs.log 2>&1 || echo "configs failed"
find $OUT -name "*kernel_stats.csv" | head
