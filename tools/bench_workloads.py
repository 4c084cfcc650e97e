"""Developer tool (GPU box): run every `bench.py --workload` in its own process and print the table kept as
profiles/rNN/bench_workloads.md.    python tools/bench_workloads.py > gpurun_out/bench_workloads.md"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [('act_per_channel_bf16', []), ('act_per_tensor_bf16', []), ('act_per_channel_f32', []), ('weight_conv_int8', ['--steps', '2000', '--warmup', '200']),
        ('weight_linear_int4', ['--steps', '1000', '--warmup', '100']),
        # config 4 on ONE GPU of the eight it names: that GPU's 128 rows of the batch of 1024 (as in rounds 1 and 2)
        ('qconv_layer3', ['--steps', '1000', '--warmup', '100', '--act-shape', '128,1024,14,14', '--graph-replay']),
        ('qlinear_8192', ['--steps', '500', '--warmup', '50', '--graph-replay']), ('act_per_channel_bf16', ['--shard-path']),
        # one rank's shard of the 8-way strong split, collectives issued by a one-rank RCCL group, eager and replayed
        ('act_per_channel_bf16', ['--act-shape', '32,512,56,56', '--shard-path', '--graph-replay']),
        ('act_per_channel_bf16', ['--steps', '20', '--warmup', '5']),
        # the same call WITHOUT the 35 untimed clock-settling steps bench.py runs before the warm-up by default
        ('act_per_channel_bf16', ['--steps', '20', '--warmup', '5', '--no-settle'])]


def main():
    print('# bench.py --workload ... on one MI355X (one box, one process per line; defaults 50 + 200 steps unless stated)\n')
    print('Configs 2, 4, 5 of BASELINE.json are host-bound in eager mode (python + autograd around a few launches per quantizer): '
          'their `ms/step` is host time;\n`graph replay` is the same step captured into a HIP graph (us per replay).  `calls`: HIP-event '
          'brackets around the C-ABI calls inside the timed region\n(a bracket includes the call\'s launch-bound helpers), algorithmic '
          'bytes over time.\n')
    print('| workload | tensors per GPU | Gelem/s | ms/step | graph replay us | roofline.frac (backward call) | calls |')
    print('|---|---|---|---|---|---|---|')
    for name, extra in RUNS:
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', name, '--no-cpu-baseline'] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            print('| `%s %s` | failed: %s |' % (name, ' '.join(extra), r.stderr.strip().splitlines()[-1:] or r.returncode))
            continue
        calls = ', '.join('%s %.0f us (%.1f TB/s)' % (k, v['ms'] * 1e3, v['algorithmic_GBps'] / 1e3) for k, v in d['calls'].items())
        print('| `%s%s` | %s | %.1f | %.4f | %s | %s | %s |' % (
            name, (' ' + ' '.join(extra)) if extra else '', d['config']['tensors_per_gpu'], d['value'], d['ms_per_step'],
            d.get('us_per_step_graph_replay', ''), d['roofline']['frac'], calls), flush=True)


if __name__ == '__main__':
    main()
