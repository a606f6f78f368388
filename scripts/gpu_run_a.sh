#!/bin/bash
# round-2 first GPU pass: the -m gpu suite, the default bench line, per-dispatch kernel traces for both dtypes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02a
mkdir -p "$OUT"
cd "$R"
python -m pytest tests -m gpu -x -q -k "not peaked" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
python bench.py > "$OUT/bench_default.log" 2>&1; tail -c 600 "$OUT/bench_default.log"
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/${dt}_trace" -o bench -- python "$R/bench.py" --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-bf16-leg > "$OUT/${dt}_trace.log" 2>&1
done
python - <<PY
import csv, glob, collections, re
for dt in ("f32", "bf16"):
    f = glob.glob("$OUT/%s_trace/*kernel_trace.csv" % dt)
    if not f: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:70]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"))
        a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open("$OUT/%s_per_dispatch.txt" % dt, "w") as o:
        for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            o.write("%-72s grid %-9s lds %-7s vgpr %-4s n %-4d avg_us %10.1f total_us %12.1f\n" % (*k, n, us / n, us))
PY
find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/bf16_per_dispatch.txt" | head -30
