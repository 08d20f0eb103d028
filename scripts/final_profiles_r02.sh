#!/bin/bash
# End-of-round evidence for round 2 (run on the GPU box from the repo root: bash scripts/final_profiles_r02.sh).
# Writes gpurun_out/final_r02/: bench lines, rocprofv3 --kernel-trace --stats summaries of the same commands, per-search
# kernel breakdowns, PMC passes of the dominant kernels.  The files judged are copied into profiles/ afterwards.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- \
      python3 $R/bench.py "$@" > $OUT/${name}_rocprof.json 2> $OUT/$name.err || echo "rocprof $name failed"
  local f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
python3 $R/__graft_entry__.py smoke > $OUT/smoke.txt 2>&1 || echo "smoke failed"
timeout -k 10 400 python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_sift1m.json 2> $OUT/bench_sift1m.err || echo "bench failed"
prof sift1m --steps 20 --warmup 3 --no-cpu-baseline --no-extras
python3 $R/scripts/trace_flat.py $OUT/sift1m/sift1m_kernel_trace.csv > $OUT/sift1m_breakdown.txt 2>&1
for p in 8 32 128; do
  timeout -k 10 300 python3 $R/bench.py --workload ivf1024 --nprobe $p > $OUT/bench_ivf1024_nprobe$p.json 2> $OUT/bench_ivf$p.err || echo "ivf bench $p failed"
  prof ivf$p --workload ivf1024 --nprobe $p --steps 10 --warmup 2
  python3 $R/scripts/trace_breakdown.py $OUT/ivf$p/ivf${p}_kernel_trace.csv > $OUT/ivf${p}_breakdown.txt 2>&1
done
VDBHIP_BENCH_FORCE_SHARDED=1 timeout -k 10 400 python3 $R/bench.py --workload marco12.5m --steps 5 --warmup 2 > $OUT/bench_marco12.5m_sharded_path.json 2> $OUT/marco.err || echo "marco failed"
prof marco --workload marco12.5m --steps 5 --warmup 2
timeout -k 10 300 python3 $R/bench.py --workload glove1.2m --no-cpu-baseline > $OUT/bench_glove1.2m.json 2> $OUT/glove.err || echo "glove failed"
timeout -k 10 300 python3 $R/bench.py --workload gaussian1m --no-cpu-baseline > $OUT/bench_gaussian1m.json 2> $OUT/gauss.err || echo "gauss failed"
python3 $R/bench.py --gpus 2 > $OUT/gpus2_on_one_gpu.txt 2>&1; echo "exit code $?" >> $OUT/gpus2_on_one_gpu.txt
cd $R
bash scripts/pmc_profile.sh final_i8 > /dev/null 2>&1
python3 scripts/pmc_summarize.py gpurun_out/pmc_final_i8 scan_i8_kernel refine_list select_kernel > $OUT/pmc_scan_i8_kernel.txt
rm -rf $OUT/sift1m $OUT/ivf8 $OUT/ivf32 $OUT/ivf128 $OUT/marco
ls $OUT
