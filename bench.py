#!/usr/bin/env python
"""Headline benchmark: fused int8 fake-quant fwd+bwd, per-channel, [256,512,56,56] bf16 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: the forward of a stats-scaled per-channel activation quantizer (AbsMax statistic over (N,H,W),
scale = max(stat, 1e-10)/128, quantize-dequantize; RescalingIntQuant with RuntimeStatsScaling in
training mode) followed by its full autograd backward (clamp mask, scale-gradient
reduction, deposit on the arg-max elements).  Nothing is skipped or cached between steps.

N > 1 (one rank per GPU: started bare, this script launches torch.distributed.run itself as a child process before it
touches the GPU; under torch.distributed.run it is a rank): the batch is sharded over the ranks and the only exchanges
are the all-reduce of the scale statistic (RCCL, <= 4 KB) in forward and the all-gather of the scale-gradient sums in
backward (brevitas_amd.distributed).  The judged line is STRONG scaling, the split SURVEY 8e names: the global
[256,512,56,56] cut into 256/N rows per rank, value = elements quantized by all ranks / max-over-ranks time.  The same
run and line also hold "n1" (rank 0 alone on the whole tensor, before the sharded job starts), "speedup_vs_n1" and
"weak" (every rank its own whole tensor).  All of that is measured on torch.distributed's collectives; then the group
gets a communicator on RCCL's C API (checked against torch.distributed on every rank, under a watchdog) and the strong
split runs once more with the C++ autograd node issuing its two collectives directly on the compute stream -- the line
takes the faster of the two and names it (config.collectives); last, the step is captured into a HIP graph (kernels and
the two direct RCCL calls), checked bit for bit against the eager step and replayed ("launch", "eager", "hipgraph").
Each late phase runs under a watchdog and after the line so far has been handed to a helper process (LineGuard).

Other workloads (--workload): the remaining BASELINE.json configs, same JSON shape --
  weight_conv_int8    config 2: Int8WeightPerChannelFloat on a [512,512,3,3] conv weight (us per step; eager and
                      HIP-graph replay)
  act_per_tensor_bf16 config 3: Int8ActPerTensorFloat (MAX statistic) on [256,512,56,56] bf16
  qconv_layer3        config 4: ResNet-50 layer3 bottleneck quantizers -- Int8 per-tensor activation
                      [1024,1024,14,14] bf16 sharded 128 per GPU + the three Int8 per-channel weights, replicated
  qlinear_8192        config 5: Int4WeightPerChannelFloat on Linear[8192,8192] + Int8ActPerTensorFloat on
                      [8192,8192] bf16 rows per GPU, per-tensor statistic all-reduced over the ranks
  weight_linear_int4  config 5's weight alone (eager and HIP-graph replay)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (the backward) priced in algorithmic HBM bytes per launch over its
                  HIP-event duration measured inside the timed region, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (a port of the reference algorithm, oracle/) timed on this host's cores
                  on a bounded sample of the same workload.  Baseline only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# The first ~10 ms of load after an idle GPU run 5-10 % slow (profiles/r02/transient.txt: the step time
# rises over steps 2..7 and settles by step ~15, with the reported sclk still ramping): steps before the W
# requested warm-up steps, untimed, until at least this many steps have run.
SETTLE_STEPS = 40

ACT_SHAPE = (256, 512, 56, 56)
WORKLOADS = {
    # name: (kind, dtype, description)
    'act_per_channel_bf16': ('act_pc', torch.bfloat16,
                             'Int8 per-channel act fake-quant fwd+bwd, AbsMax stats, [256,512,56,56] bf16'),
    'act_per_tensor_bf16': ('act_pt', torch.bfloat16,
                            'Int8ActPerTensorFloat(MAX stats) fwd+bwd, [256,512,56,56] bf16'),
    'act_per_channel_f32': ('act_pc', torch.float32,
                            'Int8 per-channel act fake-quant fwd+bwd, AbsMax stats, [256,512,56,56] f32'),
    'weight_conv_int8': ('weight_conv', torch.float32,
                         'Int8WeightPerChannelFloat fwd+bwd on a [512,512,3,3] f32 conv weight'),
    'weight_linear_int4': ('weight_linear', torch.bfloat16,
                           'Int4WeightPerChannelFloat fwd+bwd on a [8192,8192] bf16 linear weight'),
    'qconv_layer3': ('qconv', torch.bfloat16,
                     'ResNet-50 layer3 bottleneck quantizers: Int8ActPerTensorFloat(MAX) on [128,1024,14,14] bf16 per GPU '
                     '(batch 1024 over 8) + Int8WeightPerChannelFloat on [256,1024,1,1], [256,256,3,3], [1024,256,1,1]'),
    'qlinear_8192': ('qlinear', torch.bfloat16,
                     'Linear[8192,8192]: Int4WeightPerChannelFloat (replicated) + Int8ActPerTensorFloat(MAX) on '
                     '[8192,8192] bf16 activation rows per GPU, statistic all-reduced'),
}


def build_quantizer(channels, per_channel, device, group=None):
    from brevitas_amd.core.bit_width import BitWidthConst
    from brevitas_amd.core.function_wrapper import OverOutputChannelView, OverTensorView, RoundSte, TensorClamp
    from brevitas_amd.core.quant import IntQuant, RescalingIntQuant
    from brevitas_amd.core.restrict_val import FloatRestrictValue
    from brevitas_amd.core.scaling import IntScaling, RuntimeStatsScaling
    from brevitas_amd.core.stats import AbsMax
    from brevitas_amd.core.zero_point import ZeroZeroPoint
    if per_channel:
        view, stats, shape = OverOutputChannelView((1, 0, 2, 3)), AbsMax(1), (1, channels, 1, 1)
    else:
        view, stats, shape = OverTensorView(), AbsMax(), ()
    q = RescalingIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        RuntimeStatsScaling(stats, view, FloatRestrictValue(), shape, affine_rescaling=False,
                            scaling_stats_momentum=0.1, scaling_min_val=1e-10),
        IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8)).to(device)
    q.train()
    if group is not None:
        from brevitas_amd.distributed import shard_over_batch
        shard_over_batch(q, group)
    return q


class Job:
    """one workload instance on one rank: step(), the elements a step quantizes, its algorithmic bytes"""

    def __init__(self, kind, dtype, device, group, rank, act_shape=None):
        import brevitas_amd.quant as Q
        torch.manual_seed(123456 + rank)
        self.kind, self.dtype = kind, dtype
        self.acts, self.weights = [], []   # (x, g, quantizer)
        b = torch.tensor([], dtype=dtype).element_size()
        if kind in ('act_pc', 'act_pt'):
            shape = act_shape or ACT_SHAPE
            x = torch.randn(shape, device=device, dtype=dtype).requires_grad_(True)
            g = torch.randn(shape, device=device, dtype=dtype)
            self.acts.append((x, g, build_quantizer(shape[1], kind == 'act_pc', device, group)))
        elif kind in ('weight_conv', 'weight_linear'):
            if act_shape is not None:
                raise ValueError('a weight workload has no activation to reshape')
            shape, scale, bits = ((512, 512, 3, 3), 0.02, 8) if kind == 'weight_conv' else ((8192, 8192), 0.01, 4)
            w = torch.nn.Parameter((torch.randn(shape, device=device) * scale).to(dtype))
            g = torch.randn(shape, device=device, dtype=dtype)
            self.weights.append((w, g, Q.Int8WeightPerChannelFloat(w, bit_width=bits).to(device)))
        elif kind == 'qconv':
            shape = act_shape or (128, 1024, 14, 14)
            x = torch.randn(shape, device=device, dtype=dtype).requires_grad_(True)
            g = torch.randn(shape, device=device, dtype=dtype)
            self.acts.append((x, g, build_quantizer(shape[1], False, device, group)))
            torch.manual_seed(123456)  # replicated weights: the same on every rank
            for ws in ((256, 1024, 1, 1), (256, 256, 3, 3), (1024, 256, 1, 1)):
                w = torch.nn.Parameter((torch.randn(ws, device=device) * 0.02).to(dtype))
                self.weights.append((w, torch.randn(ws, device=device, dtype=dtype),
                                     Q.Int8WeightPerChannelFloat(w).to(device)))
        elif kind == 'qlinear':
            shape = act_shape or (8192, 8192)
            x = torch.randn(shape, device=device, dtype=dtype).requires_grad_(True)
            g = torch.randn(shape, device=device, dtype=dtype)
            self.acts.append((x, g, build_quantizer(shape[1], False, device, group)))
            torch.manual_seed(123456)
            w = torch.nn.Parameter((torch.randn(8192, 8192, device=device) * 0.01).to(dtype))
            self.weights.append((w, torch.randn(8192, 8192, device=device, dtype=dtype),
                                 Q.Int4WeightPerChannelFloat(w).to(device)))
        else:
            raise ValueError(kind)
        self.n_elem = sum(t[0].numel() for t in self.acts + self.weights)
        self.n_act = sum(t[0].numel() for t in self.acts)
        self.bytes_per_elem = 6 * b  # stat read + fwd read/write + bwd 2 reads / 1 write
        self.elsize = b

    def step(self):
        for x, g, q in self.acts + self.weights:
            x.grad = None
            y = q(x)[0]
            y.backward(g)


class KernelTimer:
    """HIP-event brackets around the named C-ABI calls, on the stream they are launched on"""

    def __init__(self, *names):
        self.pairs = {n: [] for n in names}
        self.enabled = False

    def before(self, name):
        if self.enabled and name in self.pairs:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.pairs[name].append([ev, None])

    def after(self, name):
        p = self.pairs.get(name)
        if self.enabled and p and p[-1][1] is None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            p[-1][1] = ev

    def mean_ms(self, name):
        ts = [a.elapsed_time(b) for a, b in self.pairs[name] if b is not None]
        return sum(ts) / len(ts) if ts else None

    def reset(self):
        for v in self.pairs.values():
            del v[:]


def usable_cpus():
    """host threads this process may really use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            quota, period = fh.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(shape, dtype, per_channel, budget_s=12.0):
    """time the oracle (oracle/, a port of the reference algorithm) on a bounded sample"""
    os.environ.setdefault('OMP_NUM_THREADS', str(usable_cpus()))  # before libgomp starts its pool
    import oracle as O
    O.build()
    n_full, c, h, w = shape
    n = 8  # 1/32 of the batch: Gelem/s is size independent to first order
    code = {torch.float32: O.F32, torch.bfloat16: O.BF16, torch.float16: O.F16}[dtype]
    g = torch.Generator().manual_seed(123456)
    x = torch.randn(n, c, h, w, generator=g).to(dtype)
    gr = torch.randn(n, c, h, w, generator=g).to(dtype)
    xn, _ = O.from_torch(x.reshape(-1))
    gn, _ = O.from_torch(gr.reshape(-1))
    if per_channel:
        d = O.make_desc(n, c, h * w, code, code, code, O.F32, scale_per_channel=True, qmin=-128.0, qmax=127.0)
    else:
        d = O.make_desc(1, 1, n * c * h * w, code, code, code, O.F32, scale_per_channel=False, qmin=-128.0,
                        qmax=127.0)
    O.step_stats_scaled(d, xn, gn, 1e-10, 128.0)  # warm-up
    # a container may see more cores than it may use: time one pass per candidate thread count, keep the best
    best = None
    for nt in sorted({usable_cpus(), 64, 32, 16, 8}, reverse=True):
        if nt > usable_cpus():
            continue
        O.set_num_threads(nt)
        O.step_stats_scaled(d, xn, gn, 1e-10, 128.0)
        t0 = time.perf_counter()
        O.step_stats_scaled(d, xn, gn, 1e-10, 128.0)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    O.set_num_threads(best[1])
    t0 = time.perf_counter()
    iters = 0
    while True:
        O.step_stats_scaled(d, xn, gn, 1e-10, 128.0)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 400:
            break
    elems = n * c * h * w
    return {
        'value': round(elems * iters / el / 1e9, 4), 'unit': 'Gelem/s', 'cores': O.num_threads(), 'kind': 'port',
        'sample': '[%d,%d,%d,%d] %s (1/%d of the batch), %d iterations in %.1f s of the C oracle '
                  '(oracle/bvq_oracle.c, OpenMP)' % (n, c, h, w, str(dtype).replace('torch.', ''), n_full // n,
                                                      iters, el)}


def timed_run(job, steps, warmup, settle, world, device, timer=None, stride=8, on_gpu=True):
    """settle + warm-up steps untimed, then `steps` steps between barriers -> max-over-ranks seconds"""

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(settle + warmup):
        job.step()
    barrier()
    # the per-call HIP events are recorded on every `stride`-th step only: each record is a barrier packet
    # in the queue and costs the GPU ~5 us of idle time (profiles/r01_gap_analysis.txt)
    t0 = time.perf_counter()
    for i in range(steps):
        if timer is not None:
            timer.enabled = i % stride == 0
        job.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if timer is not None:
        timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def graph_replay_us(job, iters=200):
    """the same step captured into a HIP graph (every launch goes to the caller's stream, nothing is read back):
    us per replay"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            job.step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):  # (the warm-up's stream: its arrival buffer exists, see _native.arrival_buffer)
        job.step()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def graphed_run(job, steps, warmup, world, device, group=None):
    """The step captured ONCE into a HIP graph (kernels and the two collectives alike: every launch goes to the capture
    stream, nothing is read back) and replayed: warm-up replays untimed, then `steps` replays between barriers.
    Before anything is timed the replay is checked against the eager step: the gradient it leaves in x.grad must equal
    the eager one bit for bit on EVERY rank, or the graph is not used.
    A sharded step is captured only when its collectives are direct RCCL calls on the capture stream
    (brevitas_amd.distributed.enable_native_collectives): torch.distributed's watchdog thread may query an event
    recorded in the capturing stream, which HIP answers with hipErrorCapturedEvent and the thread with terminate() --
    seen once in a few runs on this image.
    -> (max-over-ranks seconds or None, note)"""
    import torch.distributed as dist
    if group is not None:
        from brevitas_amd.core.quant import _fused
        fast = _fused._fast_module()
        if not fast or not fast.rccl_comm_active(group.group_name):
            return None, ('not captured: the collectives of this group go through torch.distributed, whose watchdog thread '
                          'may query a captured event and abort the process')

    def agree(flag):  # 1.0 only if every rank says so
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            job.step()
    torch.cuda.current_stream().wait_stream(s)
    sync()
    want = [t[0].grad.clone() for t in job.acts + job.weights]
    g = torch.cuda.CUDAGraph()
    ok, note = True, None
    try:
        with torch.cuda.graph(g, stream=s):  # (the warm-up's stream: its arrival buffer exists)
            job.step()
    except Exception as e:
        ok, note = False, 'capture failed: %s: %s' % (type(e).__name__, str(e).splitlines()[0][:160])
    if not agree(ok):
        return None, note or 'capture failed on another rank'
    g.replay()
    sync()
    same = all(torch.equal(t[0].grad.view(torch.int16 if t[0].grad.element_size() == 2 else torch.int32),
                           w.view(torch.int16 if w.element_size() == 2 else torch.int32))
               for t, w in zip(job.acts + job.weights, want))
    if not agree(same):
        return None, 'replay does not reproduce the eager gradients'
    for _ in range(warmup):
        g.replay()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, 'replay verified against the eager step (gradients bit for bit on every rank)'


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='act_per_channel_bf16', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--settle-steps', type=int, default=None,
                    help='developer option: untimed steps before the warm-up so that settle + warm-up reach this many '
                         '(default SETTLE_STEPS)')
    ap.add_argument('--no-settle', action='store_true',
                    help='developer option: do not run the untimed clock-settling steps before the warm-up')
    ap.add_argument('--timer-stride', type=int, default=8,
                    help='record the per-call HIP events on every n-th timed step (1: every step)')
    ap.add_argument('--graph-replay', action='store_true',
                    help='developer option (one GPU): also capture the step into a HIP graph and time its replay')
    ap.add_argument('--shard-path', action='store_true',
                    help='developer option: run the batch-sharded code path (RCCL collectives included) even with one '
                         'rank, to measure its fixed per-step overhead on a single GPU')
    ap.add_argument('--act-shape', default=None,
                    help='developer option: the GLOBAL activation shape instead of the workload\'s own, e.g. 32,512,56,56 '
                         '(what one rank of an 8-way strong split holds)')
    ap.add_argument('--backend', default='nccl', choices=('nccl', 'gloo'),
                    help='torch.distributed backend; gloo goes with --device cpu (a dry run of the launcher and the '
                         'exchange protocol on CPU tensors -- never a measurement)')
    ap.add_argument('--device', default='cuda', choices=('cuda', 'cpu'))
    ap.add_argument('--no-n1', action='store_true',
                    help='N > 1: skip rank 0\'s single-GPU run of the whole tensor (no speedup_vs_n1 in the line)')
    ap.add_argument('--no-weak', action='store_true', help='N > 1: skip the weak-scaled side measurement')
    ap.add_argument('--share-device', action='store_true',
                    help='developer option: every rank on device 0, exchanging over gloo -- a rehearsal of the N > 1 control '
                         'flow (rank 0 alone first, strong split, weak) on a one-GPU box; its numbers mean nothing')
    ap.add_argument('--c10d-collectives', action='store_true',
                    help='N > 1: keep the sharded step\'s collectives on torch.distributed (default: RCCL\'s C API from the '
                         'C++ node when its start-up check against c10d passes)')
    ap.add_argument('--graph', action='store_true',
                    help='--shard-path on one GPU: also measure the step replayed from a HIP graph and report the faster of '
                         'the two (N > 1 does so by default once the direct RCCL calls are in use)')
    ap.add_argument('--no-graph', action='store_true',
                    help='N > 1: do not measure the step replayed from a HIP graph')
    ap.add_argument('--die-in-late-phase', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--native-timeout', type=float, default=120.0,
                    help='N > 1: seconds the re-run on direct RCCL calls may take before the line is printed without it')
    ap.add_argument('--graph-timeout', type=float, default=120.0,
                    help='N > 1: seconds the HIP-graph measurement may take before the eager line is printed without it')
    ap.add_argument('--launch-timeout', type=float, default=900.0,
                    help='bare --gpus N: seconds the launcher waits for its workers before it kills them')
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class LineGuard:
    """N > 1, rank 0: a forked helper (before this process touches the GPU; it runs nothing but read / write / _exit)
    that holds the result descriptor and the read end of a pipe.  Rank 0 hands it the line as it stands before every
    late, optional phase (the re-run on direct RCCL calls, the HIP graph) and tells it when the final line is out.  If
    rank 0's end of the pipe closes without that word -- the process died inside such a phase -- the helper prints the
    last line it was handed: the measurements that were complete are not lost with the process."""

    def __init__(self, fd):
        r, w = os.pipe()
        if os.fork() == 0:
            try:
                os.close(w)
                buf = b''
                while True:
                    chunk = os.read(r, 65536)
                    if not chunk:
                        break
                    buf += chunk
                msgs = buf.split(b'\n')
                lines = [m for m in msgs if m.startswith(b'{')]
                if b'FINAL' not in msgs and lines:
                    os.write(fd, lines[-1] + b'\n')
            finally:
                os._exit(0)
        os.close(r)
        self.w = w

    def provisional(self, out, phase):
        d = dict(out)
        d['late_phase'] = ('the process ended inside the %s phase; everything in this line was measured before it' % phase)
        d.setdefault('launch', 'eager')
        os.write(self.w, (json.dumps(d) + '\n').encode())

    def final(self):
        os.write(self.w, b'FINAL\n')


def launch_workers(args, argv):
    """`python bench.py --gpus N` with no torch.distributed environment: start the N ranks the way the driver's own
    multi-GPU command does (python -m torch.distributed.run, one process per GPU) as a CHILD process group, pass
    rank 0's JSON line through, exit with the workers' status.  This process has not touched the GPU (it must not:
    a GPU-initialised process may not start other programs on the box) and never does."""
    import signal
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)   # exactly the process group started above
            except ProcessLookupError:
                break
            try:
                proc.communicate(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        sys.stderr.write('bench.py: the %d workers did not finish within %.0f s; killed\n' % (args.gpus, args.launch_timeout))
        return 124
    lines = [ln for ln in out.decode(errors='replace').splitlines() if ln.startswith('{') and '"metric"' in ln]
    if proc.returncode != 0:
        sys.stderr.write('bench.py: torch.distributed.run exited with status %d\n' % proc.returncode)
        if len(lines) == 1 and '"late_phase"' in lines[0]:
            # rank 0 died inside a late, optional phase: its helper has printed the line of the completed measurements
            sys.stdout.write(lines[0] + '\n')
            sys.stdout.flush()
            return 0
        return proc.returncode or 1
    if len(lines) != 1:
        sys.stderr.write('bench.py: expected one JSON line from rank 0, got %d\n' % len(lines))
        return 1
    sys.stdout.write(lines[0] + '\n')
    sys.stdout.flush()
    return 0


class Measurement:
    """one timed job: what the report needs of it once its tensors are freed"""

    def __init__(self, job, elapsed, steps, world, timer=None):
        self.elapsed, self.steps = elapsed, steps
        self.ms_per_step = elapsed / steps * 1e3
        self.n_elem, self.n_act, self.elsize = job.n_elem, job.n_act, job.elsize
        self.bytes_per_elem = job.bytes_per_elem
        self.shapes = [list(t[0].shape) for t in job.acts + job.weights]
        self.bwd_elems = job.acts[0][0].numel() if job.acts else job.weights[0][0].numel()
        # weights are replicated: every rank quantizes its own copy, counted once per rank (that is the work done)
        self.value = job.n_elem * world * steps / elapsed / 1e9
        self.calls = {}
        if timer is not None:
            b = job.elsize
            for key, name, passes in (('statistic', 'bvq_stats', 1), ('forward', 'bvq_fakequant_fwd', 2),
                                      ('statistic+forward', 'bvq_stats_fakequant_fwd', 3),
                                      ('backward', 'bvq_fakequant_bwd', 3)):
                ms = timer.mean_ms(name)
                if ms:
                    self.calls[key] = {'ms': round(ms, 4),
                                       'algorithmic_GBps': round(passes * b * self.bwd_elems / (ms * 1e-3) / 1e9, 1)}
            self.bwd_ms = timer.mean_ms('bvq_fakequant_bwd')
        else:
            self.bwd_ms = None

    def brief(self, scaling, **extra):
        d = {'value': round(self.value, 3), 'unit': 'Gelem/s', 'ms_per_step': round(self.ms_per_step, 4),
             'scaling': scaling, 'tensors_per_gpu': self.shapes}
        d.update(extra)
        return d


def probe_sharded_node(group, device):
    """One tiny step of the batch-sharded quantizer through the C++ autograd node (which issues its collectives
    through c10d itself) before anything is timed: if it raises on ANY rank, every rank switches the sharded node off
    for this run and the sharded steps take the Python Function (same kernels, same results, more host time).
    -> True (node in use) / False (switched off) / None (node not built)"""
    import brevitas_amd.config as config
    from brevitas_amd.core.quant import _fused
    if not _fused._fast_module() or not config.CPP_AUTOGRAD_SHARDED:
        return None
    ok = 1.0
    try:
        q = build_quantizer(8, True, device, group)
        x = torch.randn(4, 8, 14, 16, device=device, dtype=torch.bfloat16).requires_grad_(True)
        y = q(x)[0]
        y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001
        sys.stderr.write('bench.py: the sharded C++ node failed its probe (%s); using the Python Function\n' % (exc,))
        ok = 0.0
    flag = torch.tensor([ok], device=device)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=group)
    if float(flag.item()) < 1.0:
        config.CPP_AUTOGRAD_SHARDED = False
        return False
    return True


def library_digest():
    """digest of the sources the loaded libbvq.so was built from (brevitas_amd/csrc/build.py's stamp) or None"""
    try:
        from brevitas_amd.csrc import build as bvq_build
        return bvq_build.source_digest()
    except Exception:  # noqa: BLE001
        return None


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare `python bench.py --gpus N`: this process only starts the N ranks and relays rank 0's line
        sys.exit(launch_workers(args, argv))

    # stdout carries exactly ONE line, the JSON result: everything else this process or its libraries print
    # (RCCL's version banner goes to stdout) is sent to stderr until then
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    guard = LineGuard(result_fd) if world > 1 and rank == 0 else None   # (forks: before anything touches the GPU)
    kind, dtype, descr = WORKLOADS[args.workload]
    on_gpu = args.device == 'cuda'
    full_shape = tuple(int(v) for v in args.act_shape.split(',')) if args.act_shape else None
    # The CPU baseline runs FIRST, before this process touches the GPU: it may have to (re)build the
    # oracle with `make`, and a GPU-initialised process must not spawn other programs on the box.
    baseline = None
    if rank == 0 and world == 1 and on_gpu and not args.no_cpu_baseline and kind in ('act_pc', 'act_pt'):
        baseline = cpu_baseline(ACT_SHAPE, dtype, kind == 'act_pc')
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('bench.py --gpus %d found WORLD_SIZE=%d in its environment' % (args.gpus, world))
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a ROCm device (the product path has no CPU fallback for device tensors)')
        if args.share_device:
            if args.backend != 'gloo':
                raise SystemExit('--share-device goes with --backend gloo (RCCL refuses two ranks on one device)')
            local_rank = 0
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    else:
        if args.backend != 'gloo' and (world > 1 or args.shard_path):
            raise SystemExit('--device cpu goes with --backend gloo')
        device = torch.device('cpu')
        if full_shape is None:
            raise SystemExit('--device cpu is a dry run of the launcher / exchange protocol: give a small --act-shape')
    group = None
    rccl_ranks = 1
    if world > 1 or args.shard_path:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        kw = {'device_id': device} if on_gpu and args.backend == 'nccl' else {}
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(free_port()))
            dist.init_process_group(args.backend, rank=0, world_size=1, timeout=datetime.timedelta(seconds=600), **kw)
        else:
            dist.init_process_group(args.backend, timeout=datetime.timedelta(seconds=600), **kw)
        group = dist.group.WORLD
        # the number of ranks the collective library really joined: an all-reduce of ones
        ones = torch.ones(1, device=device, dtype=torch.float32)
        dist.all_reduce(ones, group=group)
        rccl_ranks = int(ones.item())

    def sync():
        if world > 1:
            torch.distributed.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def free():
        if on_gpu:
            torch.cuda.empty_cache()

    timer = None
    cpp_sharded = None
    native_coll = False
    if on_gpu:
        from brevitas_amd import _native as nat
        timer = KernelTimer('bvq_fakequant_bwd', 'bvq_fakequant_fwd', 'bvq_stats', 'bvq_stats_fakequant_fwd')
        if group is not None:
            cpp_sharded = probe_sharded_node(group, device)
            if cpp_sharded and not args.c10d_collectives and world == 1:
                # the node's two collectives through RCCL's C API on the compute stream (checked against c10d first).
                # With N > 1 this happens AFTER the whole run has been measured on torch.distributed's collectives
                # (below): a second communicator between N ranks has never been set up on this pool.
                from brevitas_amd.distributed import enable_native_collectives
                native_coll = enable_native_collectives(group)
    settle_target = SETTLE_STEPS if args.settle_steps is None else args.settle_steps
    settle = 0 if args.no_settle else max(0, settle_target - args.warmup)
    has_act = kind in ('act_pc', 'act_pt', 'qconv', 'qlinear')
    # the GLOBAL activation of the workload: the named tensor itself, or (config 4) batch 1024
    base = {'qconv': (1024, 1024, 14, 14), 'qlinear': (8192, 8192)}.get(kind, ACT_SHAPE)
    if not has_act:
        base = None
    elif full_shape is not None:
        base = full_shape

    # ---- N > 1: rank 0 alone on the whole tensor first (the N = 1 figure of the same run, same box) ---------------
    n1 = None
    if world > 1 and has_act and not args.no_n1:
        if rank == 0:
            j = Job(kind, dtype, device, None, 0, act_shape=base)
            e = timed_run(j, args.steps, args.warmup, settle, 1, device, None, on_gpu=on_gpu)
            n1 = Measurement(j, e, args.steps, 1)
            del j
            free()
        sync()

    # ---- the judged measurement ------------------------------------------------------------------------------------
    # N > 1: STRONG scaling (north_star): the global tensor cut into 1/N of its rows per rank.  N = 1: the
    # whole tensor.  Workloads without an activation (weights) are replicas: every rank quantizes its own copy.
    scaling = 'strong' if has_act else 'weak'
    if has_act and world > 1:
        rows = base[0] // world
        if rows < 1:
            raise SystemExit('%d ranks for a batch of %d' % (world, base[0]))
        shard_shape = (rows,) + tuple(base[1:])
    else:
        shard_shape = base
    if timer is not None:
        nat.set_kernel_timer(timer)
    job = Job(kind, dtype, device, group, rank, act_shape=shard_shape)
    # (a step whose calls are bracketed runs through the Python Functions so that the brackets see the C-ABI calls; one
    #  rank's shard of a strong-scaled split is host-bound there -- 0.27 against 0.16 ms, profiles/r03_strong_scaling.md --
    #  so with N > 1 fewer steps are bracketed)
    stride = args.timer_stride if world == 1 else max(args.timer_stride, 16)
    elapsed = timed_run(job, args.steps, args.warmup, settle, world, device, timer, stride, on_gpu=on_gpu)
    if timer is not None:
        nat.set_kernel_timer(None)
    main_m = Measurement(job, elapsed, args.steps, world, timer)
    replay_us = None
    graph_error = None
    if world == 1 and on_gpu and (kind in ('weight_conv', 'weight_linear') or args.graph_replay) \
            and (group is None or native_coll):   # (c10d collectives are never captured: see graphed_run)
        try:
            replay_us = graph_replay_us(job)
        except Exception as e:  # developer option on routes that may not capture (collectives): say so, keep the line
            if not args.graph_replay:
                raise
            replay_us = None
            graph_error = '%s: %s' % (type(e).__name__, str(e).splitlines()[0][:200])
    del job
    free()

    # ---- N > 1: the weak-scaled side measurement (every rank holds the whole named tensor) --------------------------
    weak = None
    if world > 1 and has_act and not args.no_weak:
        wshape = base if kind != 'qconv' else (base[0] // 8,) + tuple(base[1:])   # config 4: 128 per GPU
        j = Job(kind, dtype, device, group, rank, act_shape=wshape)
        e = timed_run(j, args.steps, args.warmup, 0, world, device, None, on_gpu=on_gpu)
        weak = Measurement(j, e, args.steps, world)
        del j
        free()

    out = None
    if rank == 0:
        m = main_m
        b = m.elsize
        headline = kind == 'act_pc' and (base == ACT_SHAPE)
        # algorithmic bytes of the backward kernel per launch: read g + read x + write dx
        bwd_bytes = 3 * b * m.bwd_elems
        achieved = bwd_bytes / (m.bwd_ms * 1e-3) / 1e9 if m.bwd_ms else None
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if on_gpu and world == 1 and os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            digest = library_digest()
            if tj.get('_source_digest') is not None and tj.get('_source_digest') == digest:
                traffic = tj.get(args.workload, {}).get('bvq_fakequant_bwd')
                if traffic is not None:
                    traffic_src = 'profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier ' \
                                  'run of this command on the same library sources (digest %s; %s), not measured by ' \
                                  'this run' % (digest[:12], tj.get('_source', 'see profiles/README.md'))
            else:
                traffic_src = 'null: profiles/traffic.json was collected on other library sources (digest %s, this ' \
                              'library %s)' % (str(tj.get('_source_digest'))[:12], str(digest)[:12])
        dt_name = str(dtype).replace('torch.', '').replace('bfloat16', 'bf16').replace('float32', 'f32')
        metric = 'Gelem/s fused int8 fake-quant fwd+bwd, per-channel, [256,512,56,56]' if headline else \
            'Gelem/s fake-quant fwd+bwd, ' + args.workload
        if world > 1 and has_act:
            par = ('dp%d, strong scaling: the global %s activation batch-sharded into %d rows per rank over %d %s ranks, '
                   'all-reduce(MAX) of the statistic + all-gather of the scale-gradient sums; weights replicated'
                   % (world, list(base), shard_shape[0], world, 'RCCL' if args.backend == 'nccl' else args.backend))
        elif world > 1:
            par = 'replicas only: %d ranks each quantize their own copy of the weight (no exchange)' % world
        else:
            par = 'single GPU'
        out = {
            'metric': metric,
            'value': round(m.value, 3), 'unit': 'Gelem/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'settle_steps': settle, 'ms_per_step': round(m.ms_per_step, 4),
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': dt_name,
            'data': 'synthetic' if on_gpu else 'synthetic; DRY RUN on CPU tensors over %s (launcher / exchange-protocol '
                                               'check through the pure-torch CPU route: not a measurement)' % args.backend,
            'config': {'workload': descr, 'name': args.workload,
                       'global_activation': list(base) if has_act else None,
                       'tensors_per_gpu': m.shapes,
                       'quantizer': 'RescalingIntQuant(IntQuant, RuntimeStatsScaling(AbsMax) / StatsFromParameterScaling'
                                    '(AbsMax), IntScaling, ZeroZeroPoint, BitWidthConst), training mode',
                       'parallelism': par,
                       'rccl_ranks': rccl_ranks,
                       'sharded_autograd_node': {True: 'C++ (collectives issued through c10d from the node)', False: 'Python Function (the C++ node failed its start-up probe)', None: None}[cpp_sharded] if group is not None else None,
                       'collectives': (('RCCL C API from the C++ node, on the compute stream (checked against c10d at start-up)'
                                        if native_coll else 'torch.distributed (c10d)') if group is not None else None),
                       'algorithmic_bytes_per_elem': m.bytes_per_elem},
            'hbm_frac_whole_step': round(m.bytes_per_elem * m.n_elem * world / (m.ms_per_step * 1e-3) / 1e9
                                         / (HBM_PEAK_GBS * world), 4),
            'roofline': {'kernel': 'fakequant_bwd_kernel (bvq_fakequant_bwd)', 'bound': 'hbm',
                         'achieved': round(achieved, 1) if achieved else None, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': bwd_bytes,
                         'avg_launch_ms': round(m.bwd_ms, 4) if m.bwd_ms else None},
        }
        if n1 is not None:
            out['n1'] = n1.brief('single GPU, rank 0 alone on the whole tensor, same run')
            out['speedup_vs_n1'] = round(m.value / n1.value, 3)
        if weak is not None:
            out['weak'] = weak.brief('weak')
        if replay_us is not None:
            out['us_per_step_eager'] = round(m.ms_per_step * 1e3, 2)
            out['us_per_step_graph_replay'] = round(replay_us, 2)
        if graph_error is not None:
            out['graph_replay_error'] = graph_error
        # the streaming calls of the step, same method (each bracket includes its launch-bound helpers); the
        # one-launch forward (statistic + quantizer) moves the statistic's read through the Infinity Cache
        out['calls'] = m.calls
        if baseline is not None:
            out['cpu_baseline'] = baseline

    import threading
    emit_lock = threading.Lock()
    emitted = []

    def emit():
        with emit_lock:
            if rank == 0 and not emitted:
                emitted.append(True)
                sys.stdout.flush()
                os.write(result_fd, (json.dumps(out) + '\n').encode())
                if guard is not None:
                    guard.final()

    if args.die_in_late_phase and world > 1:   # test hook (tests/test_bench_launcher.py): a rank 0 that dies after its
        if guard is not None:                  # measurements are complete leaves its line behind through the helper
            guard.provisional(out, 'test')
            os.abort()
        time.sleep(5.0)                        # (the other ranks: torch.distributed.run ends them)

    # ---- N > 1: the strong split once more with the two collectives as direct RCCL calls -----------------------------------
    # Everything above ran on torch.distributed's collectives and is complete.  Now the group gets a communicator of its
    # own (checked against torch.distributed on every rank) and the strong job runs again; the line takes the faster of
    # the two and says which (`config.collectives`, the other one under "c10d" / "native_collectives").  If the set-up
    # fails its check, raises, or nothing comes back within --native-timeout seconds, the line is printed as it stands.
    if on_gpu and group is not None and world > 1 and cpp_sharded and has_act and scaling == 'strong' \
            and not args.c10d_collectives:
        def abandon_native():
            if out is not None:
                out['native_collectives'] = {'error': 'no result within %g s: abandoned, the run on torch.distributed\'s '
                                                      'collectives stands' % args.native_timeout}
                out['launch'] = 'eager'
            emit()
            os._exit(0)
        if guard is not None:
            guard.provisional(out, 'direct-RCCL-collectives')
        watchdog = threading.Timer(args.native_timeout, abandon_native)
        watchdog.daemon = True
        watchdog.start()
        n_m, note, failed = None, None, False
        try:
            from brevitas_amd.distributed import enable_native_collectives
            native_coll = enable_native_collectives(group)
            if native_coll:
                job = Job(kind, dtype, device, group, rank, act_shape=shard_shape)
                e = timed_run(job, args.steps, args.warmup, 0, world, device, None, on_gpu=True)
                n_m = Measurement(job, e, args.steps, world)
                del job
                free()
            else:
                note = 'set-up or its check against torch.distributed failed on some rank'
        except Exception as exc:
            failed, n_m = True, None
            note = 'failed: %s: %s' % (type(exc).__name__, str(exc).splitlines()[0][:160])
        watchdog.cancel()
        if out is not None:
            if n_m is not None:
                c10d = {'value': out['value'], 'ms_per_step': out['ms_per_step']}
                native = {'value': round(n_m.value, 3), 'ms_per_step': round(n_m.ms_per_step, 4)}
                if n_m.value > out['value']:
                    out['value'], out['ms_per_step'] = native['value'], native['ms_per_step']
                    out['hbm_frac_whole_step'] = round(n_m.bytes_per_elem * n_m.n_elem * world / (n_m.ms_per_step * 1e-3)
                                                       / 1e9 / (HBM_PEAK_GBS * world), 4)
                    if n1 is not None:
                        out['speedup_vs_n1'] = round(n_m.value / n1.value, 3)
                    out['config']['collectives'] = ('RCCL C API from the C++ node, on the compute stream (checked against '
                                                    'c10d at start-up); the same steps on torch.distributed\'s collectives '
                                                    'are under "c10d"')
                    out['c10d'] = c10d
                else:
                    out['native_collectives'] = native
            else:
                out['native_collectives'] = {'error': note}
        if failed:  # the communicator may be unusable: no teardown
            if out is not None:
                out['launch'] = 'eager'
            emit()
            os._exit(0)

    # ---- the sharded step replayed from a HIP graph (N > 1; --shard-path on one GPU) -------------------------------------
    # One rank's shard of a strong split is a launch-bound step: ~120 us of kernels behind ~200 us of host time
    # (profiles/r03_strong_scaling.md).  Captured once -- kernels, the all-reduce and the all-gather alike -- and
    # replayed, the host leaves the critical path.  Everything above is measured eagerly and is complete at this point:
    # if the capture raises, the replay does not reproduce the eager gradients, or nothing comes back within
    # --graph-timeout seconds, the eager line is printed as it stands; if the process dies inside the capture -- N ranks
    # of RCCL in a captured graph have never run on this one-GPU-per-box pool -- rank 0's helper prints it (LineGuard).
    # Only direct RCCL calls are ever captured (graphed_run).
    if on_gpu and group is not None and has_act and scaling == 'strong' and native_coll \
            and (args.graph or (world > 1 and not args.no_graph)):
        def abandon():
            if out is not None:
                out['hipgraph'] = {'error': 'no result within %g s: abandoned, the eager measurement stands' % args.graph_timeout}
                out['launch'] = 'eager'
            emit()
            os._exit(0)
        if guard is not None:
            guard.provisional(out, 'HIP-graph')
        watchdog = threading.Timer(args.graph_timeout, abandon)
        watchdog.daemon = True
        watchdog.start()
        failed = False
        try:
            job = Job(kind, dtype, device, group, rank, act_shape=shard_shape)
            g_elapsed, note = graphed_run(job, args.steps, args.warmup, world, device, group)
            g_m = Measurement(job, g_elapsed, args.steps, world) if g_elapsed else None
            del job
        except Exception as e:
            failed, g_m = True, None
            note = 'failed: %s: %s' % (type(e).__name__, str(e).splitlines()[0][:160])
        watchdog.cancel()
        if out is not None:
            eager = {'value': out['value'], 'ms_per_step': out['ms_per_step']}
            if g_m is not None:
                out['hipgraph'] = {'value': round(g_m.value, 3), 'ms_per_step': round(g_m.ms_per_step, 4), 'note': note}
                if g_m.value > out['value']:
                    out['value'], out['ms_per_step'] = round(g_m.value, 3), round(g_m.ms_per_step, 4)
                    out['hbm_frac_whole_step'] = round(g_m.bytes_per_elem * g_m.n_elem * world / (g_m.ms_per_step * 1e-3)
                                                       / 1e9 / (HBM_PEAK_GBS * world), 4)
                    if n1 is not None:
                        out['speedup_vs_n1'] = round(g_m.value / n1.value, 3)
                    out['launch'] = ('hipgraph: the step (kernels + both collectives) captured once and replayed; '
                                     'the same steps issued eagerly are under "eager"')
                    out['eager'] = eager
            else:
                out['hipgraph'] = {'error': note}
        if failed:  # a failed capture may have left the stream or the communicator unusable: no teardown
            emit()
            os._exit(0)
    if out is not None and 'launch' not in out:
        out['launch'] = 'eager'
    emit()
    if group is not None:
        if native_coll:
            from brevitas_amd.distributed import disable_native_collectives
            disable_native_collectives(group)
        # the line is out: a teardown that does not come back (a rank gone after a failed capture) must not hold the job
        guard = threading.Timer(30.0, lambda: os._exit(0))
        guard.daemon = True
        guard.start()
        torch.distributed.destroy_process_group()
        guard.cancel()


if __name__ == '__main__':
    main()
