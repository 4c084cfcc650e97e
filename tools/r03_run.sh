out=gpurun_out/r03u
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_affine_golden.py tests/test_gpu_fixed_point.py tests/test_gpu_modules.py tests/test_gpu_configs.py -q -m gpu -x > $out/tests.txt 2>&1 || { tail -50 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
