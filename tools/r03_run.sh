#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
mkdir -p gpurun_out/r03x
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03x/tests_full.txt 2>&1; tail -5 gpurun_out/r03x/tests_full.txt
