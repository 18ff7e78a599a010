# same-box A / B of the grid leg: the working tree against a checkout of an older commit under _ab_old/ (built there)
for r in 1 2; do
  for d in . _ab_old; do
    (cd $d && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$d', 'grid', d['grid']['value'], 'sec', d['grid']['seconds'], 'step ms', d['ms_per_step'])")
  done
done
