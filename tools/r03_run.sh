#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
timeout -k 10 900 python tools/microbench.py > $OUT/microbench.txt 2> $OUT/microbench.err || { tail -5 $OUT/microbench.err; exit 1; }
cat $OUT/microbench.txt
