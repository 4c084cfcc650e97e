set -e
out=gpurun_out/r03b
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_onepass.py -x -q -m gpu > $out/onepass_tests.txt 2>&1 || { tail -40 $out/onepass_tests.txt; exit 1; }
tail -3 $out/onepass_tests.txt
for v in "1 1" "1 0" "0 0" "1 1" "1 0" "0 0"; do set -- $v
  BREVITAS_AMD_ONEPASS=$1 BREVITAS_AMD_ONEPASS_BWD=$2 python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $out/bench_$1$2_$RANDOM.json 2>> $out/bench.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$out/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d['value'], 'Gelem/s', d['ms_per_step'], 'ms', {k: v['ms'] for k, v in d['calls'].items()})
PY
