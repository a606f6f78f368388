#!/bin/bash
# Is a kernel mix clock- / power-limited?  Samples rocm-smi (socket power, sclk) every 50 ms while bench.py runs a dtype.
#   DT=f16 [STEPS=60] bash scripts/power_probe.sh      -> gpurun_out/power_$DT.txt (summary on stdout)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
DT=${DT:-f16}
rocm-smi --showpower --showclocks --showmaxpower > $OUT/power_${DT}_idle.txt 2>&1
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level|Average Graphics Package Power|Current Socket" ; echo "--"; sleep 0.05; done ) > $OUT/power_${DT}_samples.txt &
SAMPLER=$!
python $R/bench.py --dtype $DT --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-legs --no-roofline > $OUT/power_${DT}_bench.log 2>&1
kill $SAMPLER
python - <<PY
import re
txt = open("$OUT/power_${DT}_samples.txt").read()
pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
ck = [int(x) for x in re.findall(r"\((\d+)Mhz\)", txt)]
print("$DT", "power samples", len(pw), "max %.0f W" % max(pw) if pw else "", "median of top half %.0f W" % (sorted(pw)[3 * len(pw) // 4] if pw else 0), "sclk min/median/max", (min(ck), sorted(ck)[len(ck) // 2], max(ck)) if ck else None)
PY
grep -i "max\|cap" $OUT/power_${DT}_idle.txt | head -5
tail -1 $OUT/power_${DT}_bench.log | cut -c1-200
