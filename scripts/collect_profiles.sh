#!/bin/bash
# Collect the judged profile set on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py, FETCH_SIZE and WRITE_SIZE in separate PMC passes, and plain bench runs.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16 f16 f32s; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${dt}_stats" -o bench -- python "$R/bench.py" --dtype $dt --steps 8 --no-cpu-baseline --no-bf16-leg > "$OUT/${dt}_bench_profiled.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${dt}_fetch" -o pmc -- python "$R/bench.py" --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-bf16-leg > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${dt}_write" -o pmc -- python "$R/bench.py" --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-bf16-leg > /dev/null 2>&1
  find "$OUT" -name "*kernel_trace.csv" -delete
done
python "$R/bench.py" > "$OUT/f32_bench.log" 2>&1
python "$R/bench.py" --dtype bf16 --no-cpu-baseline > "$OUT/bf16_bench.log" 2>&1
python "$R/bench.py" --dtype f16 --no-cpu-baseline > "$OUT/f16_bench.log" 2>&1
python "$R/bench.py" --dtype f32s --no-cpu-baseline > "$OUT/f32s_bench.log" 2>&1
du -sh "$OUT"
