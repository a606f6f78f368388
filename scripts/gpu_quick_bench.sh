#!/bin/bash
# quick A/B loop: bit-identity tests of choice (K=...), then the per-kernel table of bench.py for the given dtypes
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/quick
mkdir -p "$OUT"
cd "$R"
if [ -n "${K:-}" ]; then
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_hourglass.py} -m gpu -x -q -k "$K" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"; tail -5 "$OUT/pytest.log"
fi
for dt in ${DTYPES:-f16}; do
timeout 400 python bench.py --dtype $dt --full --no-cpu-baseline --no-legs > "$OUT/bench_$dt.log" 2>&1
python - <<PY
import json
l=[x for x in open("$OUT/bench_$dt.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$dt frames/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2))
    for k in d["roofline"]["kernels"][:${NK:-10}]: print("  ", k["kernel"], k["launches"], round(k["avg_us"],1), round(k["tflops"],1), "min/pmc GB", round(k["bytes_min"]/1e9,2), k["bytes_pmc"] and round(k["bytes_pmc"]/1e9,2))
else:
    print(open("$OUT/bench_$dt.log").read()[-2000:])
PY
done
