#!/bin/bash
# IVF (BASELINE config 4) evidence of the final build: bench lines, rocprofv3 kernel stats + per-search breakdown for nprobe
# 8 / 32 / 128, and the FETCH/WRITE PMC passes of the list scan at nprobe 8.  Run on the GPU box from the repo root.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ivf_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for p in 8 32 128; do
  timeout -k 10 300 python3 $R/bench.py --workload ivf1024 --nprobe $p > $OUT/bench_ivf1024_nprobe$p.json 2> $OUT/bench_ivf$p.err || echo "ivf bench $p failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ivf$p -o ivf$p -- \
      python3 $R/bench.py --workload ivf1024 --nprobe $p --steps 10 --warmup 2 --no-cpu-baseline > $OUT/ivf${p}_rocprof.json 2> $OUT/ivf$p.err || echo "rocprof $p failed"
  f=$(find $OUT/ivf$p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/ivf${p}_kernel_stats.csv
  python3 $R/scripts/trace_breakdown.py $OUT/ivf$p/ivf${p}_kernel_trace.csv > $OUT/ivf${p}_breakdown.txt 2>&1
  rm -rf $OUT/ivf$p
done
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc8 -o pmc8_$pass -- \
      python3 $R/bench.py --workload ivf1024 --nprobe 8 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc8_$pass.log 2>&1 || echo "pmc $pass failed"
done
python3 $R/scripts/pmc_summarize.py $OUT/pmc8 scan_i8_kernel > $OUT/pmc_ivf8_list_scan.txt
rm -rf $OUT/pmc8
ls $OUT
