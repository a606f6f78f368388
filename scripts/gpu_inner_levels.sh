#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/inner; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for dt in f16 f32; do
  ST=$([ $dt = f16 ] && echo 4 || echo 2)
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/$dt" -- python $R/bench.py --dtype $dt --steps $ST --warmup 1 --no-cpu-baseline --no-legs --no-roofline > "$OUT/bench_$dt.log" 2>&1
  echo "== $dt" ; python $R/scripts/inner_levels.py "$OUT/$dt" $ST 2>&1 | tail -45
done > "$OUT/inner_levels.txt" 2>&1
cat "$OUT/inner_levels.txt"
find "$OUT" -name "*.csv" -size +3M -delete
