#!/bin/bash
# hourglass parity tests + selected others, then bench legs per dtype
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/hg_check
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_hourglass.py} -m gpu -q -s ${K:+-k "$K"} > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
grep -E "peaked|worst|rel err|passed|failed|FAILED|rc=" "$OUT/pytest.log" | tail -40
for dt in ${DTYPES:-}; do
timeout 400 python bench.py --dtype $dt --full --no-cpu-baseline --no-bf16-leg > "$OUT/bench_$dt.log" 2>&1
python - <<PY
import json
l=[x for x in open("$OUT/bench_$dt.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$dt frames/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2))
    for k in d["roofline"]["kernels"][:9]: print("  ", k["kernel"], k["launches"], round(k["avg_us"],1), round(k["tflops"],1))
else:
    print(open("$OUT/bench_$dt.log").read()[-2000:])
PY
done
