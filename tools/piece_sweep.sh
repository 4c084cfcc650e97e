#!/bin/bash
# Developer experiment (GPU box): piece size of long rows in the QUANTIZER kernels (BVQ_QUANT_PIECE_CHUNKS = 16-byte
# chunks per lane and piece) -- the per-tensor activation (config 3) and the [8192,8192] weight (config 5).
for rep in 1 2; do
for pc in 8 7 5 9; do
  for wl in act_per_tensor_bf16 weight_linear_int4; do
    echo "## BVQ_QUANT_PIECE_CHUNKS=$pc $wl"
    BVQ_QUANT_PIECE_CHUNKS=$pc python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  value %.1f Gelem/s  ms/step %.4f  replay_us %s  calls %s' % (d['value'], d['ms_per_step'], d.get('us_per_step_graph_replay'), {k:(v['ms'],v['algorithmic_GBps']) for k,v in d['calls'].items()}))"
  done
  BVQ_QUANT_PIECE_CHUNKS=$pc python tools/pt_bench.py bf16,f16 2>&1 | grep -v amdgpu
done
done
