#!/bin/bash
# The round's acceptance run on the GPU box: the full -m gpu suite exactly as the driver runs it (-x), then smoke, then the default bench.
#   gpurun --timeout 2400 -- "HEAD_REV=$(git rev-parse HEAD) DIRTY=$(git status --porcelain | wc -l) bash scripts/gpu_full_check.sh"
# The log names the head it ran at (there is no .git on the box: the caller passes it) and is copied to profiles/rNN_pytest_gpu.log.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/full_check
mkdir -p "$OUT"
cd "$R"
{
  echo "head ${HEAD_REV:-unknown} (uncommitted files: ${DIRTY:-?})  $(date -u +%FT%TZ)"
  echo "libdf3d_hip.so sha256 $(sha256sum deepfly3d_amd/libdf3d_hip.so | cut -c1-16)"
  echo "\$ python -m pytest tests -m gpu -x -q ${PYTEST_EXTRA:-}"
} > "$OUT/pytest.log"
timeout ${PYTEST_TIMEOUT:-2000} python -m pytest tests -m gpu ${PYTEST_X--x} -q -rs --durations=15 ${PYTEST_EXTRA:-} >> "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -25 "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -2 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench_default.log" 2>&1
python - <<PY
import json
l=[x for x in open("$OUT/bench_default.log") if x.startswith("{")]
if l:
    print("stdout JSON lines:", len(l), "bytes of the line:", len(l[-1].encode()))
    d=json.loads(l[-1])
    d=json.load(open("$R/"+d["tables"]))   # the whole record beside the short line
    print("f32 frames/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "settle", d.get("warmup_settle"), "frac", round(d["roofline"]["frac"],4))
    for k in ("config1_f32_split","config2_bf16","config2_f16","config4_share"):
        if k in d: print(k, round(d[k].get("value",0),1), d[k].get("error",""))
    print("cpu", d.get("cpu_baseline"))
    for k in d["roofline"]["kernels"][:8]: print("  ", k["kernel"], k["launches"], round(k["avg_us"],1), round(k["tflops"],1))
else:
    print(open("$OUT/bench_default.log").read()[-3000:])
PY
