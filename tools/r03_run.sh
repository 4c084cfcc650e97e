out=gpurun_out/r03j
mkdir -p $out
bash tools/profile.sh $PWD/gpurun_out/r03j/prof > $out/profile.log 2>&1
echo "profile rc=$?"
python tools/bench_workloads.py > $out/bench_workloads.md 2> $out/bench_workloads.err
echo "workloads rc=$?"
cat $out/bench_workloads.md
python tools/qconv_parts.py > $out/qconv_parts.txt 2>&1
python tools/host_cost.py > $out/host_cost.txt 2>&1
tail -20 $out/qconv_parts.txt
tail -25 $out/host_cost.txt
head -60 gpurun_out/r03j/prof/summary.md
