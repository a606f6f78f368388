#!/bin/bash
# MFMA busy / clocks / waits of one dtype's kernels in one go: DT=f32 bash scripts/pmc_quick.sh
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o pmc -- python $R/bench.py --dtype ${DT:-f32} --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-legs > $OUT/g$i.log 2>&1
  find $OUT/g$i -name "*kernel_trace.csv" -delete
done
cd $R; python scripts/summarize_counters.py pmcq_${DT:-f32} $OUT/g1 $OUT/g2 $OUT/g3 > /dev/null 2>&1; cp profiles/pmcq_${DT:-f32}_counters.csv $OUT/; rm -f profiles/pmcq_*
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/pmcq_${DT:-f32}_counters.csv")):
    if float(r['GRBM_GUI_ACTIVE_per_launch'] or 0) < 1e6: continue
    f=lambda k: (('%.3f' % float(r[k])) if r.get(k) not in (None,'') else '-')
    print('  %-62s n=%-3s mfma_busy %s clk %s lds_conf %s wait %s' % (r['kernel'][:62], r['launches'], f('mfma_busy_frac'), f('clock_ghz'), f('lds_conflict_frac'), f('wait_frac')))
PY
