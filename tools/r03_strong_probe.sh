#!/bin/bash
# what one rank of the strong-scaled split holds at N = 2 / 4 / 8 (128 / 64 / 32 rows of [256,512,56,56]): the
# batch-sharded code path with ONE rank (RCCL world 1) next to the unsharded path on the same shard, plus a kernel trace
# of the 32-row step.  One GPU; the real link latency of N ranks is not in these numbers.
set -e
out=gpurun_out/${1:-r03_strong}
mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_n1_steps20.json 2> $out/bench_n1_steps20.err
echo "n1 done"
for rows in 128 64 32; do
  python bench.py --steps 100 --warmup 20 --shard-path --no-cpu-baseline --act-shape $rows,512,56,56 > $out/shard_$rows.json 2> $out/shard_$rows.err
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --act-shape $rows,512,56,56 > $out/plain_$rows.json 2> $out/plain_$rows.err
  echo "rows $rows done"
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof32 -o shard32 -- python3 bench.py --steps 100 --warmup 20 --shard-path --no-cpu-baseline --act-shape 32,512,56,56 > $out/prof32.json 2> $out/prof32.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$out/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], 'Gelem/s', d['ms_per_step'], 'ms', d.get('calls'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
ls $out/prof32
