#!/bin/bash
# End-of-round evidence for round 4 (on the GPU box, from the repo root: bash scripts/final_profiles_r03.sh [quick]).
# Writes gpurun_out/final_r04/: the driver's bench line, rocprofv3 --kernel-trace --stats summaries of the same commands,
# per-search kernel breakdowns, PMC passes (separate --pmc passes, kernel trace only) of the dominant kernels.
# The files judged are copied into profiles/ afterwards (r04_*).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o $name -- \
      python3 $R/bench.py "$@" > $OUT/${name}_rocprof.json 2> $OUT/$name.err || echo "rocprof $name failed"
  local f=$(find $OUT/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv
}
pmc() {    # name, counters..., then -- bench args
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout -k 10 240 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $OUT/pmc_$name -o $name -- \
      python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/pmc_$name.log 2>&1 || echo "pmc pass $name failed"
}
PHASE=${1:-all}     # all | stats (bench line, kernel stats, breakdowns) | pmc | pmc1 | pmc2 (counter passes, stamps) | pmcx (the flat scans only)
if [ "$PHASE" = "all" ] || [ "$PHASE" = "stats" ]; then
python3 $R/__graft_entry__.py smoke > $OUT/smoke.txt 2>&1 || echo "smoke failed"
echo "== bench (driver command)"; date
timeout -k 10 500 python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
echo "== kernel stats"; date
prof sift1m --steps 20 --warmup 5 --no-cpu-baseline --no-extras
python3 $R/scripts/trace_breakdown.py $(find $OUT/sift1m -name '*kernel_trace.csv' | head -1) > $OUT/sift1m_breakdown.txt 2>&1
for p in 8 32 128; do
  prof ivf$p --workload ivf1024 --nprobe $p --steps 10 --warmup 2 --no-cpu-baseline
  python3 $R/scripts/trace_breakdown.py $(find $OUT/ivf$p -name '*kernel_trace.csv' | head -1) ivf_select > $OUT/ivf${p}_breakdown.txt 2>&1
done
prof msmarco_ivf --workload msmarco_ivf --steps 10 --warmup 2 --no-cpu-baseline
python3 $R/scripts/trace_breakdown.py $(find $OUT/msmarco_ivf -name '*kernel_trace.csv' | head -1) ivf_select > $OUT/msmarco_ivf_breakdown.txt 2>&1
if true; then
  prof gaussian1m --workload gaussian1m --steps 20 --warmup 5 --no-cpu-baseline
  prof glove --workload glove1.2m --steps 20 --warmup 5 --no-cpu-baseline
  prof marco --workload marco12.5m --steps 5 --warmup 2
fi
fi   # stats phase
if [ "$PHASE" = "pmcx" ]; then      # the three flat scans on layout "x16" (the IVF / K-loop kernels did not change)
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
pmc sift1m_sq $SQ1 --
pmc sift1m_fetch FETCH_SIZE --
pmc sift1m_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --
pmc sift1m_grbm GRBM_GUI_ACTIVE GRBM_COUNT --
for w in gaussian1m glove1.2m; do
  n=${w%%1*}
  pmc ${n}_sq $SQ1 -- --workload $w
  pmc ${n}_fetch FETCH_SIZE -- --workload $w
  pmc ${n}_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- --workload $w
  pmc ${n}_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- --workload $w
done
cd $R
exit 0
fi
if [ "$PHASE" != "stats" ]; then
echo "== PMC passes"; date
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
if [ "$PHASE" != "pmc2" ]; then      # pmc1: the int8 headline and config 4
pmc sift1m_sq $SQ1 --
pmc sift1m_fetch FETCH_SIZE --
pmc sift1m_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --
pmc sift1m_grbm GRBM_GUI_ACTIVE GRBM_COUNT --
for p in 8 128; do
  pmc ivf${p}_sq $SQ1 -- --workload ivf1024 --nprobe $p
  pmc ivf${p}_fetch FETCH_SIZE -- --workload ivf1024 --nprobe $p
  pmc ivf${p}_write WRITE_SIZE -- --workload ivf1024 --nprobe $p
  pmc ivf${p}_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- --workload ivf1024 --nprobe $p
done
pmc ivf32_fetch FETCH_SIZE -- --workload ivf1024 --nprobe 32
pmc ivf32_write WRITE_SIZE -- --workload ivf1024 --nprobe 32
fi
if [ "$PHASE" != "pmc1" ]; then      # pmc2: the embedding-shaped IVF leg and the real-valued flat scans
pmc msmarco_sq $SQ1 -- --workload msmarco_ivf
pmc msmarco_fetch FETCH_SIZE -- --workload msmarco_ivf
pmc msmarco_write WRITE_SIZE -- --workload msmarco_ivf
pmc msmarco_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- --workload msmarco_ivf
for w in gaussian1m glove1.2m; do
  n=${w%%1*}
  pmc ${n}_sq $SQ1 -- --workload $w
  pmc ${n}_fetch FETCH_SIZE -- --workload $w
  pmc ${n}_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- --workload $w
  pmc ${n}_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- --workload $w
done
echo "== stamps of the fp16 flat scan (ablations build, scan_variant 6)"; date
for w in gaussian1m glove1.2m; do
  timeout -k 10 200 python3 $R/scripts/stamp_scan.py $w > $OUT/stamps_$w.txt 2>&1 || echo "stamps $w failed"
done
fi
cd $R
fi   # pmc phase
cd $R
for n in sift1m ivf8 ivf32 ivf128 msmarco gaussian glove; do
  mkdir -p $OUT/pmcsum_$n
  cp $OUT/pmc_${n}_*/*/*counter_collection.csv $OUT/pmcsum_$n/ 2>/dev/null || find $OUT -path "*pmc_${n}_*" -name "*counter_collection.csv" -exec cp --backup=numbered {} $OUT/pmcsum_$n/ \;
done
python3 scripts/pmc_summarize_r03.py $OUT > $OUT/pmc_summary_$PHASE.txt 2>&1
rm -rf $OUT/sift1m $OUT/ivf8 $OUT/ivf32 $OUT/ivf128 $OUT/msmarco_ivf $OUT/gaussian1m $OUT/glove $OUT/marco
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT
