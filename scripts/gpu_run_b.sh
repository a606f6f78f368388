#!/bin/bash
# ring kernel: correctness, then per-dispatch timing of the bf16 bench
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02b
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_hourglass.py -m gpu -x -q -k "ring or bf16" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
timeout 300 python bench.py --dtype bf16 --no-cpu-baseline > "$OUT/bench_bf16.log" 2>&1
python - <<PY
import json
l=[x for x in open("$OUT/bench_bf16.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("bf16 frames/s", d["value"], "ms/step", d["ms_per_step"])
    for k in d["roofline"]["kernels"][:8]: print(k["kernel"], k["launches"], round(k["avg_us"],1), round(k["tflops"],1))
else:
    print(open("$OUT/bench_bf16.log").read()[-3000:])
PY
