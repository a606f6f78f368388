#!/bin/bash
# run-to-run reproducibility of the default LSMR form, its per-segment cycles (DF3D_LSMR_DEBUG), then tests + timings
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R"
for T in 1000 15; do DF3D_LSMR_DEBUG=1 timeout 300 python tests/perf/lsmr_stress.py $T 3 0 2>&1 | grep "cycles\|runs" | tail -2; done
for T in 1000 15 300 700; do
  timeout 300 python tests/perf/lsmr_stress.py $T 300 0 2>&1 | tail -1
  timeout 300 python tests/perf/lsmr_stress.py $T 300 9 2>&1 | tail -1
done
bash scripts/gpu_ba_check.sh 2>&1 | grep -v "subnormal\|RESULT"
