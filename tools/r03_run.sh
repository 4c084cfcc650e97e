#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
: > $OUT/fuzz_soak.txt
for seed in 20261005 31337; do
  BVQ_FUZZ_CASES=1300 BVQ_FUZZ_SEED=$seed timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 | sed "s/^/seed $seed, 1300 cases: /" >> $OUT/fuzz_soak.txt || { cat $OUT/fuzz_soak.txt; exit 1; }
done
cat $OUT/fuzz_soak.txt
