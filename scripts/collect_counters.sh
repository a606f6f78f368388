#!/bin/bash
# MFMA / LDS utilisation counters of bench.py's kernels (run through gpurun from the repo root).  One rocprofv3
# --pmc pass per counter group, no other trace domain next to --pmc; summarised by scripts/summarize_counters.py.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
OUT=$R/gpurun_out/${TAG}_counters
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for dt in ${DTYPES:-f32 bf16 f16 f32s}; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" \
             "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
             "MfmaUtil" \
             "SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/${dt}_g$i" -o pmc -- python "$R/bench.py" --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-bf16-leg > "$OUT/${dt}_g$i.log" 2>&1
    find "$OUT/${dt}_g$i" -name "*kernel_trace.csv" -delete
  done
done
du -sh "$OUT"
