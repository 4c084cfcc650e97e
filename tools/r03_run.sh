#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
bash tools/profile.sh $OUT/prof > $OUT/profile.log 2>&1 || { tail -5 $OUT/profile.log; exit 1; }
head -12 $OUT/prof/summary.md | cut -c1-160
python tools/bench_workloads.py > $OUT/bench_workloads.md 2> $OUT/bench_workloads.err || { tail -5 $OUT/bench_workloads.err; exit 1; }
tail -12 $OUT/bench_workloads.md | cut -c1-260
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_call.json 2> $OUT/bench_driver_call.err; cut -c1-200 $OUT/bench_driver_call.json
timeout -k 10 600 python tools/microbench.py > $OUT/microbench.txt 2>/dev/null; grep -c TB $OUT/microbench.txt
