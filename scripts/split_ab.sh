cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hourglass.py -m gpu -x -q -k "split" 2>&1 | tail -5
for sp in 0 1; do
DF3D_SPLIT1=$sp python bench.py --dtype f32 --steps 4 --warmup 1 --full --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('split1=$sp', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms/step')
        for k in d['roofline']['kernels'][:7]: print('  ', k['kernel'], k['launches'], round(k['avg_us'],1), 'us', round(k['tflops'],1), 'TF/s')"
done
