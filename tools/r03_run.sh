#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03pt
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/tests_full.txt 2>&1; tail -3 $OUT/tests_full.txt
grep -q "failed\|error" $OUT/tests_full.txt && exit 1
for sh in "256,64,56,56 bf16" "1024,16,32,32 f32"; do set -- $sh; echo "== shape $1 $2"; timeout -k 10 200 python tools/onepass_ab.py --shape $1 --dtype $2 --rounds 6 2>&1 | grep absmax; done
