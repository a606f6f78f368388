#!/bin/bash
# run-to-run reproducibility of the default LSMR form (and of the two-kernel form), then the timings
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R"
DF3D_LSMR_DEBUG=1 timeout 300 python tests/perf/lsmr_stress.py 1000 20 7 2>&1 | grep -v "results != ordered sum of inputs: 0, workgroups disagreeing: 0, values differing from the first run: 0" | tail -8
for T in 1000 15 300 700; do
  timeout 300 python tests/perf/lsmr_stress.py $T 300 0 2>&1 | tail -1
  timeout 300 python tests/perf/lsmr_stress.py $T 300 9 2>&1 | tail -1
done
