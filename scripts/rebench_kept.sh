#!/bin/bash
# Only the four kept bench lines of a round (profiles/<tag>_<dtype>_bench.json), against the profiles/traffic.json already in the tree:
# for a change that leaves the hourglass kernels' code objects alone.  Summarise with  python scripts/summarize_round.py <tag>.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python bench.py --tables "$OUT/f32_bench_full.json" > "$OUT/f32_bench.log" 2>&1
python bench.py --dtype bf16 --no-cpu-baseline --tables "$OUT/bf16_bench_full.json" > "$OUT/bf16_bench.log" 2>&1
python bench.py --dtype f16 --no-cpu-baseline --tables "$OUT/f16_bench_full.json" > "$OUT/f16_bench.log" 2>&1
python bench.py --dtype f32s --no-cpu-baseline --tables "$OUT/f32s_bench_full.json" > "$OUT/f32s_bench.log" 2>&1
tail -1 "$OUT/f32_bench.log" | cut -c1-400
