#!/bin/bash
# frames/s of bench.py against DF3D_CHAIN_VIEWS (chunk size of the full-resolution chains; 0 = whole batch per launch)
cd ${GRAFT_REPO_ROOT:-$PWD}
for dt in ${DTYPES:-f16 f32}; do
for cv in ${CVS:-0 8 16 24 32 64 128}; do
  DF3D_CHAIN_VIEWS=$cv python bench.py --dtype $dt --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$dt chain_views', $cv, 'frames/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))"
done; done
