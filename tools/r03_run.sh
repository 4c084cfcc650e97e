#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
bash tools/profile.sh $OUT/prof > $OUT/profile.log 2>&1 || { tail -5 $OUT/profile.log; exit 1; }
cat $OUT/prof/summary.md | head -40
python tools/bench_workloads.py > $OUT/bench_workloads.md 2> $OUT/bench_workloads.err || { tail -5 $OUT/bench_workloads.err; exit 1; }
cat $OUT/bench_workloads.md | tail -14
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_call.json 2> $OUT/bench_driver_call.err; cut -c1-300 $OUT/bench_driver_call.json
