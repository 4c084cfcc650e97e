out=gpurun_out/r03o
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -q -m gpu > $out/gpu_tests.txt 2>&1
rc=$?
tail -8 $out/gpu_tests.txt
exit $rc
