#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03list
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_shared_quant_golden.py tests/test_gpu_modules.py tests/test_gpu_cpp_autograd.py -m gpu -x -q 2>&1 | tail -40 > $OUT/tests.txt; cat $OUT/tests.txt
