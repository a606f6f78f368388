#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/energy; mkdir -p "$OUT"; cd "$R"
{
for rep in 1 2; do
  unset DF3D_LIB; timeout 120 python tests/perf/probe_energy.py f16 896 6 2>&1 | tail -1
  for v in m512 m513 m639; do
    DF3D_LIB=$R/scratch/variants/libdf3d_hip_$v.so timeout 120 python tests/perf/probe_energy.py f16 896 6 2>&1 | tail -1
  done
done
} > "$OUT/energy2.txt" 2>&1
cat "$OUT/energy2.txt"
