#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02f
mkdir -p "$OUT"
cd "$R"
timeout 300 python scripts/dbg_ring_tile.py 3 > "$OUT/dbg.log" 2>&1; grep -v "^  dim\|amdgpu.ids" "$OUT/dbg.log" | tail -8
timeout 300 python scripts/ab_ring_tile.py 896 > "$OUT/ab.log" 2>&1; cat "$OUT/ab.log" | tail -5
for rt in ${RTS:-2}; do
DF3D_LIB=scratch/timing/libdf3d_hip_timing.so timeout 300 python scripts/probe_ring.py 896 bf16 $rt > "$OUT/probe_$rt.log" 2>&1; tail -30 "$OUT/probe_$rt.log"
done
timeout 900 python -m pytest tests/test_gpu_hourglass.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
