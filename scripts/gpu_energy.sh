#!/bin/bash
# joules per 16-bit forward of the product library and of the ring kernel's ablation builds (scripts/build_variant.sh mN -DBR_ABLM=N, a9 -DBR_ABL=9):
# the measured price of each term of the 16-bit ring bottleneck -> gpurun_out/energy/energy.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/energy; mkdir -p "$OUT"; cd "$R"
ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -30 > "$OUT/hwmon_ls.txt"
{
for rep in 1 2; do
  unset DF3D_LIB; timeout 120 python tests/perf/probe_energy.py f16 896 ${SECS:-6} 2>&1 | tail -1
  for v in m8 m16 m4 m28 m2 m32 m64 m1 a9; do
    [ -f scratch/variants/libdf3d_hip_$v.so ] && DF3D_LIB=$R/scratch/variants/libdf3d_hip_$v.so timeout 120 python tests/perf/probe_energy.py f16 896 ${SECS:-6} 2>&1 | tail -1
  done
done
unset DF3D_LIB; timeout 120 python tests/perf/probe_energy.py f32 896 ${SECS:-6} 2>&1 | tail -1
} > "$OUT/energy.txt" 2>&1
cat "$OUT/energy.txt"
