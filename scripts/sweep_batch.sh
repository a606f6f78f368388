cd $GRAFT_REPO_ROOT
for dt in f16 f32; do for f in 64 128 256 512; do
  s=$((1024/f))
  python bench.py --dtype $dt --frames-per-step $f --steps $s --warmup 1 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$dt frames/step', $f, 'steps', $s, 'frames/s', round(d['value'],1))"
done; done
