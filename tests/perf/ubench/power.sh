#!/bin/bash
# watts and clock of each steady-state loop of power.hip (run on the GPU box): bash tests/perf/ubench/power.sh
cd "$(dirname "$0")"
hipcc -O2 --offload-arch=gfx950 power.hip -o /tmp/power_ubench || exit 1
for mode in mfma_rand mfma_zero lds mix hbm l2; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level"; sleep 0.1; done ) > /tmp/power_$mode.txt &
  S=$!
  /tmp/power_ubench $mode 4
  kill $S
  python3 - <<PY
import re
t = open("/tmp/power_$mode.txt").read()
pw = sorted(float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", t)); ck = sorted(int(x) for x in re.findall(r"\((\d+)Mhz\)", t))
print("   $mode: power median of the upper half %.0f W (max %.0f), sclk median %d MHz" % (pw[3 * len(pw) // 4], pw[-1], ck[len(ck) // 2]))
PY
done
