#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/tests_full.txt 2>&1; tail -3 $OUT/tests_full.txt
grep -q "failed\|error" $OUT/tests_full.txt && { grep -n "Error\|^E " $OUT/tests_full.txt | head; exit 1; }
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile.sh $OUT/prof > $OUT/profile.log 2>&1 || { tail -5 $OUT/profile.log; exit 1; }
head -14 $OUT/prof/summary.md | cut -c1-200
SH="802816,512,1,bf16 802816,512,1,f16 802816,512,1,f32 65536,4096,1,bf16 65536,4096,1,f16 65536,4096,1,f32 1024,2048,49,bf16 1024,1024,196,f16 16384,8192,1,bf16"
for lib in build/variants/libbvq_head.so ""; do echo "== ${lib:-final build}"; BREVITAS_AMD_LIB=$lib timeout -k 10 200 python tools/cols_bench.py $SH 2>&1 | grep -v "amdgpu.ids\|^libbvq\|^build\|^/"; done > $OUT/cols_final.txt
cat $OUT/cols_final.txt
