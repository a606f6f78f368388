#!/bin/bash
# PMC passes over the bf16 bench (one group per rocprofv3 run; --kernel-trace only, no other trace domain beside --pmc).
#   bash scripts/pmc_groups.sh <outdir-under-gpurun_out> [dtype]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1
DT=${2:-bf16}
ONLY=${3:-}   # optional: space-separated group numbers to run
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $i "; then continue; fi
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -o pmc -- python "$R/bench.py" --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-bf16-leg > "$OUT/g$i.log" 2>&1
  find "$OUT/g$i" -name "*kernel_trace.csv" -delete
done <<GROUPS
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU
GROUPS
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in sorted(glob.glob("$OUT/g*/*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|^void |hgk::", "", r["Kernel_Name"])
        name = name.split("(")[0][:60]
        key = (name, r.get("Grid_Size", "?"))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (key, r["Dispatch_Id"], f)
        if did not in seen:
            seen.add(did)
    for k in {d[0] for d in seen}:
        cnt[k] = max(cnt[k], sum(1 for d in seen if d[0] == k))
with open("$OUT/summary.txt", "w") as o:
    for key, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        if "at::" in key[0] or c.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
        o.write("%s grid=%s dispatches=%d\n" % (key[0], key[1], cnt[key]))
        for n, v in sorted(c.items()):
            o.write("    %-36s %.4g\n" % (n, v / max(cnt[key], 1)))
print(open("$OUT/summary.txt").read()[:6000])
PY
