out=gpurun_out/r03i
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_cpp_autograd.py tests/test_gpu_two_ranks.py tests/test_gpu_distributed.py tests/test_gpu_modules.py tests/test_gpu_graphs.py -q -m gpu -x > $out/tests.txt 2>&1
rc=$?
tail -30 $out/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/r03_strong_probe.sh r03i_strong
