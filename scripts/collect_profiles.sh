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
# profiles/traffic.json from THIS collection first (round 5 wrote the kept lines before it: every one said traffic_is_current false)
python "$R/scripts/summarize_round.py" $TAG --traffic-only > "$OUT/traffic_summary.log" 2>&1
cd "$R"
python bench.py --tables "$OUT/f32_bench_full.json" > "$OUT/f32_bench.log" 2>&1
python bench.py --dtype bf16 --no-cpu-baseline --tables "$OUT/bf16_bench_full.json" > "$OUT/bf16_bench.log" 2>&1
python bench.py --dtype f16 --no-cpu-baseline --tables "$OUT/f16_bench_full.json" > "$OUT/f16_bench.log" 2>&1
python bench.py --dtype f32s --no-cpu-baseline --tables "$OUT/f32s_bench_full.json" > "$OUT/f32s_bench.log" 2>&1
du -sh "$OUT"
