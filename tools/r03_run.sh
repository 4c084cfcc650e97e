#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_call.json 2> $OUT/bench_driver_call.err; echo rc=$?
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo rc=$?
python - <<'PY'
import json
for f in ('bench_driver_call', 'bench_n1'):
    d = json.loads(open('gpurun_out/r03final/%s.json' % f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
PY
