#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03cols
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/tests_full.txt 2>&1; tail -3 $OUT/tests_full.txt
grep -q "failed\|error" $OUT/tests_full.txt && exit 1
SH="802816,512,1,bf16 802816,512,1,f16 802816,512,1,f32 65536,4096,1,bf16 65536,4096,1,f16 65536,4096,1,f32 1024,2048,49,bf16 1024,1024,196,f16 16384,8192,1,bf16"
for lib in build/variants/libbvq_head.so "" build/variants/libbvq_head.so ""; do
  echo "== ${lib:-this build}"; BREVITAS_AMD_LIB=$lib timeout -k 10 200 python tools/cols_bench.py $SH 2>&1 | grep -v "amdgpu.ids\|^libbvq\|^build\|^/" || exit 1
done > $OUT/final_ab.txt
cat $OUT/final_ab.txt
python tools/moments_bench.py > $OUT/moments.txt 2>&1 || true
