#!/bin/bash
# Development builds of libdf3d_hip.so with extra -D switches, for same-box A/B runs through DF3D_LIB (scripts/abl_bench.sh):
#
#   bash scripts/build_variant.sh NAME [-DFLAG ...]        ->  scratch/variants/libdf3d_hip_NAME.so   (scratch/ is git-ignored)
#
# Only hourglass.hip is recompiled; every other object comes from the regular build (python -m deepfly3d_amd.build).
# Switches the kernels understand (none is defined in the product build):
#   -DDF3D_BT_TIMING        per-phase s_memtime sums of wave 0 of the ring kernels (scripts/probe_ring.py, probe_l1.py); the atomics cost ~17 %
#   -DBR_ABL=n              16-bit ring kernel: 1 no phase-2 MFMAs | 3 no phase-2 weight DMA | 4 phase 2 without waits / barriers |
#                           5 no x loads | 6 no bn1 arithmetic | 7, 8 phases 2(-3) without barriers | 9 no MFMAs at all
#   -DBR_ABL=10             16-bit ring kernel WITHOUT phase 1 (the t1 tile keeps whatever the LDS holds): what phases 2-3 cost alone
#   -DBR_ABLM=mask          16-bit ring kernel, combinable: 1 no MFMAs (phases 1 and 3; W2D's phase 2 keeps its own), 2 no weight DMA, 4 no x loads,
#                           8 no residual loads, 16 no output stores, 32 no W2 fragment reloads, 64 no t1 fragment reads, 512 no phase-2 MFMAs (W2D form), 128 empty workgroups
#                           (dispatch cost), 256 prologue only
#   -DBR_RET=n              the workgroup returns at checkpoint n (1 before phase 2's barrier .. 6 after the second half's K loop): wall-clock decomposition
#   -DBR_P2_DEPTH=n         t1 fragment groups requested n ahead in phase 2 (default 1);  -DBR_ST_POLICY=0..3  output stores plain / sc1 / nt / sc0 sc1
#   -DHG_NT_STORES=1        streaming stores in the stack heads and the 16-bit layer1 kernel too (measured: no change)
#   -DBR_FORCE_LDS=90000    ring kernels at ONE workgroup per CU
#   -DBR_SETPRIO            s_setprio around the phase-2 MFMAs
#   -DC1_ABLM=mask          fp32 conv1 kernel: 1 no MFMAs, 2 no weight DMA, 4 no x loads, 8 no t1 stores
#   -DBRF_ABLM=mask         fp32 / f32s ring (tail) kernel: 1 no phase-2 MFMAs, 2 no weight DMA after the prologue, 4 a quarter of phase 2's MFMAs and fragment reads
#   -DBRF_NO_T1DMA          fp32 tail kernels without their t1 halo DMA;  -DBRF_NO_LATE_RES  without the residual tiles requested in the epilogue
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scratch/variants
objs=""
for f in deepfly3d_amd/csrc/*.hip; do
  b=$(basename $f)
  [ $b = hourglass.hip ] && continue
  [ -f deepfly3d_amd/csrc/_obj/$b.o ] || { echo "run python -m deepfly3d_amd.build first"; exit 1; }
  objs="$objs deepfly3d_amd/csrc/_obj/$b.o"
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast "$@" -Iinclude -c deepfly3d_amd/csrc/hourglass.hip -o scratch/variants/hourglass_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scratch/variants/libdf3d_hip_$name.so $objs scratch/variants/hourglass_$name.o
rm -f scratch/variants/hourglass_$name.o
echo scratch/variants/libdf3d_hip_$name.so
