#!/bin/bash
# GPU box: per-kernel times of the device JPEG decode (rocprofv3 kernel trace of scripts/probe_jpeg.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/jpeg_prof
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o jp --output-format csv -- python $GRAFT_REPO_ROOT/scripts/probe_jpeg.py ${1:-896} ${2:-5} 2>&1 | grep -v "^W\|rocprof" | tail -4
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "jpeg" in r["Name"] or "fill" in r["Name"].lower():
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
