"""`python bench.py --gpus N` started bare (no torch.distributed environment) must start its own N ranks, relay rank 0's
ONE JSON line and exit with the workers' status -- the way the driver starts the multi-GPU scaling run.  Checked here with
the dry mode (`--backend gloo --device cpu`: CPU tensors through the package's pure-torch CPU route, no kernels, not a
measurement): the launcher, the rendezvous, the strong-scaled split and the exchange protocol of the sharded quantizer."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bare_gpus_2_prints_one_json_line_with_the_strong_split():
    r = _bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--device', 'cpu', '--act-shape',
               '8,16,7,7')
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['scaling'] == 'strong'
    assert out['config']['rccl_ranks'] == 2                 # counted by an all-reduce of ones
    assert out['config']['global_activation'] == [8, 16, 7, 7]
    assert out['config']['tensors_per_gpu'] == [[4, 16, 7, 7]]  # the global batch cut into 8 / 2 rows per rank
    assert out['n1']['tensors_per_gpu'] == [[8, 16, 7, 7]]      # rank 0 alone on the whole tensor, same run
    assert abs(out['speedup_vs_n1'] - out['value'] / out['n1']['value']) < 0.51  # values are rounded to 3 decimals
    assert out['weak']['scaling'] == 'weak' and out['weak']['tensors_per_gpu'] == [[8, 16, 7, 7]]
    assert 'DRY RUN' in out['data']
    for key in ('metric', 'value', 'unit', 'ms_per_step', 'higher_is_better', 'vs_baseline', 'dtype', 'roofline'):
        assert key in out


def test_a_failing_rank_makes_the_launcher_exit_non_zero():
    # one row for two ranks: every worker refuses, the launcher must pass the failure on and print no result line
    r = _bench('--gpus', '2', '--steps', '1', '--warmup', '0', '--backend', 'gloo', '--device', 'cpu', '--act-shape',
               '1,4,2,2')
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_the_launcher_kills_workers_that_do_not_finish():
    r = _bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--device', 'cpu', '--act-shape',
               '8,16,7,7', '--launch-timeout', '0.2')
    assert r.returncode == 124
    assert not r.stdout.strip()


def test_a_rank_0_that_dies_in_a_late_phase_leaves_its_line_behind():
    """before every late, optional phase (direct RCCL calls, HIP graph) rank 0 hands the line as it stands to a helper it
    forked before touching the GPU; if the process then dies, the helper prints that line -- marked -- and the launcher
    passes it on (--die-in-late-phase: rank 0 aborts right after handing its complete measurements over)"""
    r = _bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--device', 'cpu', '--act-shape',
               '8,16,7,7', '--die-in-late-phase')
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert 'late_phase' in out and out['n_gpus'] == 2 and out['scaling'] == 'strong'
    assert out['n1']['tensors_per_gpu'] == [[8, 16, 7, 7]] and out['weak']['scaling'] == 'weak'
