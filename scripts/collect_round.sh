#!/bin/bash
# One GPU call for everything profiles/<tag>_* holds: kernel stats + FETCH / WRITE passes + plain bench lines (collect_profiles.sh), the SQ
# counter groups (collect_counters.sh), one rank's full share of configs[3] / configs[4], the strong-scaled stream at N = 1, the front-end
# rates, the power / clock probe.  Summarise afterwards with  python scripts/summarize_round.py <tag>.
#   bash scripts/collect_round.sh r04
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r05}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
bash scripts/collect_profiles.sh $TAG > "$OUT/collect_profiles.log" 2>&1
bash scripts/collect_counters.sh $TAG > "$OUT/collect_counters.log" 2>&1
python bench.py --rank-share 8 --stream-frames 100000 --force-collective --no-cpu-baseline --no-roofline > "$OUT/rankshare_f32_cfg3.log" 2>&1
python bench.py --rank-share 8 --stream-frames 100000 --ba-window 1000 --force-collective --dtype f16 --no-cpu-baseline --no-roofline > "$OUT/rankshare_f16_cfg4.log" 2>&1
python bench.py --rank-share 8 --stream-frames 100000 --ba-window 1000 --force-collective --dtype bf16 --no-cpu-baseline --no-roofline > "$OUT/rankshare_bf16_cfg4.log" 2>&1
python bench.py --strong --stream-frames 12800 --ba-window 1000 --dtype f16 --no-cpu-baseline --no-roofline > "$OUT/strong_f16_cfg4_n1.log" 2>&1
python bench.py --strong --stream-frames 2560 --no-cpu-baseline --no-roofline > "$OUT/strong_f32_cfg3_n1.log" 2>&1
for dt in f16 f32 f32s; do DT=$dt STEPS=$([ $dt = f32 ] && echo 12 || echo 40) bash scripts/power_probe.sh > "$OUT/power_$dt.txt" 2>&1; cp gpurun_out/power_${dt}_samples.txt "$OUT/" 2>/dev/null; done
python tests/perf/probe_ba.py 1000 > "$OUT/ba_1000.txt" 2>&1; python tests/perf/probe_ba.py 15 > "$OUT/ba_15.txt" 2>&1
python tests/perf/probe_ba_sections.py 1000 > "$OUT/ba_sections_1000.txt" 2>&1; python tests/perf/probe_ba_sections.py 15 > "$OUT/ba_sections_15.txt" 2>&1
du -sh "$OUT"; tail -1 "$OUT/f16_bench.log" | cut -c1-300
