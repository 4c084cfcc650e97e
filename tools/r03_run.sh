out=gpurun_out/r03r
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -q -m gpu > $out/gpu_tests.txt 2>&1
rc=$?
tail -6 $out/gpu_tests.txt
[ $rc -ne 0 ] && exit $rc
python tools/bench_workloads.py > $out/bench_workloads.md 2> $out/bench_workloads.err
cat $out/bench_workloads.md
