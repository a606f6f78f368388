cd $GRAFT_REPO_ROOT
for f in 1 2 4 8 16 32 128; do
  s=$((1024/f)); if [ $s -gt 256 ]; then s=256; fi
  python bench.py --dtype f16 --frames-per-step $f --steps $s --warmup 4 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('f16 frames/step', $f, 'views', 7*$f, 'frames/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))"
done
for f in 2 8 32; do
  s=$((512/f))
  python bench.py --dtype f32 --frames-per-step $f --steps $s --warmup 4 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('f32 frames/step', $f, 'frames/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3))"
done
