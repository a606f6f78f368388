#!/bin/bash
# bundle adjustment: tests, then wall time per LSMR form and grid size (T = 1000 window, T = 15 sample)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/ba; mkdir -p "$OUT"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_ba.py -q -x -s > "$OUT/pytest_ba.log" 2>&1; tail -5 "$OUT/pytest_ba.log"
{
for T in 1000 15; do
  echo "== two kernels per iteration (round 4), T=$T"; DF3D_LSMR_KERNELS=2 timeout 300 python tests/perf/probe_ba.py $T 2>&1 | tail -2
  for G in 0 128; do
    echo "== persistent with grid barriers, launch-based arithmetic, grid $G (0 = default), T=$T"; DF3D_LSMR_KERNELS=1 DF3D_LSMR_GRID=$G timeout 300 python tests/perf/probe_ba.py $T 2>&1 | tail -2
  done
  echo "== data-local persistent form (default), T=$T"; timeout 300 python tests/perf/probe_ba.py $T 2>&1 | tail -2
  echo "== sections, default form, T=$T"; timeout 300 python tests/perf/probe_ba_sections.py $T 2>&1 | tail -16
done
} > "$OUT/timings.txt" 2>&1
cat "$OUT/timings.txt"
[ -x scratch/mfma_f16_denorm ] && scratch/mfma_f16_denorm | tail -12
