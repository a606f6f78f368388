#!/bin/bash
# per-kernel table of bench.py for a list of library variants: LIBS="scratch/habl/libdf3d_hip_habl1.so ..." KERN=head
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
for lib in default ${LIBS:-}; do
  envs=""
  case $lib in
    default) unset DF3D_LIB;;
    *=*) unset DF3D_LIB; envs=$lib;;   # VAR=VALUE: the default library with that environment (e.g. DF3D_RING2=0)
    *) export DF3D_LIB=$R/$lib;;
  esac
  env $envs python bench.py --dtype ${DT:-f16} --steps 2 --warmup 1 --full --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        rows = [k for k in d['roofline']['kernels'] if '${KERN:-head}' in k['kernel']]
        print('$lib', round(d['ms_per_step'], 2), 'ms/step;', '; '.join(k['kernel'].split('<')[0] + '<' + k['kernel'].split(',')[-1] + ' %.0f us' % k['avg_us'] for k in rows))"
done
