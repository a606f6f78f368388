#!/bin/bash
# full -m gpu suite + smoke + both bench legs
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/full_check
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -4 "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
for dt in f32 bf16; do
timeout 400 python bench.py --dtype $dt --no-cpu-baseline --no-bf16-leg > "$OUT/bench_$dt.log" 2>&1
python - <<PY
import json
l=[x for x in open("$OUT/bench_$dt.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$dt frames/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2))
    for k in d["roofline"]["kernels"][:6]: print("  ", k["kernel"], k["launches"], round(k["avg_us"],1), round(k["tflops"],1))
else:
    print(open("$OUT/bench_$dt.log").read()[-2000:])
PY
done
