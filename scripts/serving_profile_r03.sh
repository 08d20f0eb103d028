set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03s; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for tag in "flat:--nq 1" "ivf8:--nq 1 --ivf 8" "flat64:--nq 64"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o $name -- python3 $R/scripts/latency_serving.py $args > $OUT/$name.log 2>&1 || echo "$name failed"
  f=$(find $OUT/$name -name '*kernel_trace.csv' | head -1)
  python3 $R/scripts/trace_breakdown.py $f > $OUT/${name}_breakdown.txt 2>&1
  python3 $R/scripts/trace_timeline.py $f 14 > $OUT/${name}_timeline.txt 2>&1
  rm -rf $OUT/$name
done
cd $R
timeout -k 10 200 python3 scripts/latency_serving.py > $OUT/latency_plain.txt 2>&1 || echo "latency failed"
timeout -k 10 200 python3 scripts/latency_serving.py --option fused_stats=0 > $OUT/latency_separate_stats.txt 2>&1 || echo "latency (separate stats) failed"
timeout -k 10 200 python3 scripts/latency_serving.py > $OUT/latency_plain_b.txt 2>&1 || echo "latency failed"
timeout -k 10 300 python3 scripts/soak.py > $OUT/soak.txt 2>&1; echo "soak rc=$?"
tail -3 $OUT/soak.txt
