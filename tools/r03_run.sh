#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03native
mkdir -p $OUT
: > $OUT/native_ab.txt
for rep in 1 2 3; do
for a in "--act-shape 32,512,56,56 --shard-path --c10d-collectives" "--act-shape 32,512,56,56 --shard-path" "--act-shape 32,512,56,56"; do
  timeout -k 10 300 python bench.py $a --steps 200 --warmup 50 --no-cpu-baseline 2>$OUT/err.txt > $OUT/line.json || { tail -5 $OUT/err.txt; exit 1; }
  python -c "
import sys, json
d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1])
print('$a', '| ms/step', d['ms_per_step'], '| collectives', (d['config'].get('collectives') or 'none (unsharded quantizer)')[:24])
" >> $OUT/native_ab.txt
done
done
cat $OUT/native_ab.txt
