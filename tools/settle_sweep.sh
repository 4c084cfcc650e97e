#!/bin/bash
# Developer experiment (GPU box): the driver's call (--steps 20 --warmup 5) after different numbers of untimed
# settling steps; each setting twice, fresh process each time.
for rep in 1 2; do
  for st in 40 120 300 800; do
    python bench.py --steps 20 --warmup 5 --settle-steps $st --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('settle+warmup $st: value %.1f Gelem/s  ms/step %.4f  bwd %.4f fwd %.4f stat %.4f' % (d['value'], d['ms_per_step'], d['calls']['backward']['ms'], d['calls']['forward']['ms'], d['calls']['statistic']['ms']))"
  done
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('default 50+200: value %.1f Gelem/s  ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
