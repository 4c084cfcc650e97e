#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03graph
mkdir -p $OUT
timeout -k 10 300 python bench.py --act-shape 32,512,56,56 --shard-path --graph --graph-timeout 0.002 --steps 100 --warmup 30 --no-cpu-baseline 2>$OUT/err.txt > $OUT/line.json; echo "rc=$?"
python -c "
import sys, json
d = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1])
print('| value', d['value'], '| ms/step', d['ms_per_step'], '| launch', d.get('launch'), '| eager', d.get('eager'), '| hipgraph', d.get('hipgraph'))
"
