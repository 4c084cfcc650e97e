#!/bin/bash
# scratch GPU script of round 3 (one box per call)
set -o pipefail
OUT=gpurun_out/r03final
mkdir -p $OUT
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --share-device --steps 20 --warmup 5 > $OUT/rehearsal_n2.json 2> $OUT/rehearsal_n2.err; echo "rc=$?"
tail -5 $OUT/rehearsal_n2.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03final/rehearsal_n2.json').read().strip().splitlines()[-1])
keep = {k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'scaling', 'speedup_vs_n1', 'launch') if k in d}
keep['n1'] = d.get('n1'); keep['weak'] = d.get('weak'); keep['config'] = {k: d['config'][k] for k in ('tensors_per_gpu', 'parallelism', 'rccl_ranks', 'sharded_autograd_node', 'collectives')}
print(json.dumps(keep, indent=1))
PY
