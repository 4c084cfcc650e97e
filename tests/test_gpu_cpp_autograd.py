"""The weight quantizer's autograd node in C++ (brevitas_amd/csrc/bvq_autograd.cpp) against the Python
torch.autograd.Function: same kernels, so y / scale / dx must be identical bit for bit -- on the direct route and on
every case the node hands back to Python (a gradient through `scale`, a strided gradient, only `scale` used)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _bits(t):
    return t.detach().contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32)


def _step(q, w, g, h=None, strided=False, use_y=True):
    w.grad = None
    y, scale, _, _ = q(w)
    gg = g.transpose(0, 1).contiguous().transpose(0, 1) if strided else g
    loss = (y * gg).sum() if use_y else 0.0
    if h is not None:
        loss = loss + (scale.reshape(-1).float() * h).sum()
    loss.backward()
    return y.detach().clone(), scale.detach().clone(), w.grad.detach().clone()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(64, 32, 3, 3), (128, 256), (16, 8, 1, 1)], ids=lambda s: 'x'.join(map(str, s)))
def test_cpp_node_equals_python_function(dtype, shape, monkeypatch):
    import brevitas_amd.quant as Q
    from brevitas_amd.core.quant import _fused
    assert _fused._fast_module(), 'brevitas_amd/_bvq_autograd.so is not built (python -c "import __graft_entry__ as g; g.build()")'
    torch.manual_seed(123456)
    w = torch.nn.Parameter((torch.randn(shape, device=DEV) * 0.05).to(dtype))
    g = torch.randn(shape, device=DEV).to(dtype)
    h = torch.randn(shape[0], device=DEV)
    q = Q.Int8WeightPerChannelFloat(w).to(DEV)
    calls = {'n': 0}
    real = _fused.fast_stats_fakequant

    def counted(*a, **k):
        r = real(*a, **k)
        calls['n'] += r is not None
        return r
    cases = [dict(), dict(h=h), dict(strided=True), dict(h=h, use_y=False)]
    monkeypatch.setattr(_fused, 'fast_stats_fakequant', counted)
    fast = [_step(q, w, g, **c) for c in cases]
    assert calls['n'] == len(cases), 'the C++ node did not take these steps'
    with torch.no_grad():
        y_ng = q(w)[0]
    monkeypatch.setattr(_fused, 'fast_stats_fakequant', lambda *a, **k: None)
    slow = [_step(q, w, g, **c) for c in cases]
    for (ya, sa, da), (yb, sb, db), c in zip(fast, slow, cases):
        assert torch.equal(_bits(ya), _bits(yb)) and torch.equal(_bits(sa), _bits(sb)), c
        assert torch.equal(_bits(da), _bits(db)), c
    assert torch.equal(_bits(y_ng), _bits(slow[0][0]))


def test_cpp_node_step_captures_into_a_hip_graph():
    import brevitas_amd.quant as Q
    from brevitas_amd.core.quant import _fused
    assert _fused._fast_module()
    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(64, 32, 3, 3, device=DEV) * 0.05)
    g = torch.randn(64, 32, 3, 3, device=DEV)
    q = Q.Int8WeightPerChannelFloat(w).to(DEV)

    def step():
        w.grad = None
        y = q(w)[0]
        y.backward(g)
        return y
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    want_y, want_dx = step().detach().clone(), w.grad.detach().clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = step()
        dx = w.grad
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want_y) and torch.equal(dx, want_dx)


def _act_steps(q, x, g, steps, cases):
    out = []
    for _ in range(steps):
        for c in cases:
            xi = x.clone().requires_grad_(True)
            y, scale = q(xi)[:2]
            gg = g.transpose(0, 1).contiguous().transpose(0, 1) if c.get('strided') else g
            loss = (y * gg).sum() if c.get('use_y', True) else 0.0
            if c.get('h') is not None:
                loss = loss + (scale.reshape(-1).float() * c['h'][:scale.numel()]).sum()
            loss.backward()
            out.append((y.detach().clone(), scale.detach().clone(), xi.grad.detach().clone(),
                        q.scaling_impl.runtime_stats.running_stats.detach().clone()))
    return out


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('per_channel', [True, False], ids=['per_channel', 'per_tensor'])
@pytest.mark.parametrize('shape', [(24, 48, 28, 28), (6, 10, 56, 56), (4, 16, 7, 9)], ids=lambda s: 'x'.join(map(str, s)))
def test_cpp_activation_node_equals_python_function(dtype, per_channel, shape, monkeypatch):
    """the stats-scaled ACTIVATION quantizer (statistic kernel + quantizer kernel, running statistic folded in) through
    the C++ node against the Python Function: y, scale, dx and the running statistic bit for bit over several steps, on
    the direct route and on what the node hands back (gradient through `scale`, strided gradient, only `scale` used)"""
    from bench import build_quantizer
    from brevitas_amd.core.quant import _fused
    assert _fused._fast_module(), 'brevitas_amd/_bvq_autograd.so is not built'
    torch.manual_seed(123456)
    x = torch.randn(shape, device=DEV).to(dtype)
    g = torch.randn(shape, device=DEV).to(dtype)
    h = torch.randn(shape[1], device=DEV)
    cases = [dict(), dict(h=h), dict(strided=True), dict(h=h, use_y=False)]
    calls = {'n': 0}
    real = _fused.fast_act_stats_fakequant

    def counted(*a, **k):
        r = real(*a, **k)
        calls['n'] += r is not None
        return r
    monkeypatch.setattr(_fused, 'fast_act_stats_fakequant', counted)
    fast = _act_steps(build_quantizer(shape[1], per_channel, torch.device(DEV)), x, g, 2, cases)
    assert calls['n'] == 2 * len(cases), 'the C++ node did not take these steps'
    monkeypatch.setattr(_fused, 'fast_act_stats_fakequant', lambda *a, **k: None)
    slow = _act_steps(build_quantizer(shape[1], per_channel, torch.device(DEV)), x, g, 2, cases)
    for i, (a, b) in enumerate(zip(fast, slow)):
        for ta, tb, what in zip(a, b, ('y', 'scale', 'dx', 'running')):
            assert torch.equal(_bits(ta), _bits(tb)), (i, what)


def test_cpp_activation_node_sharded_world_of_one(monkeypatch):
    """the batch-sharded branch of the node (float32 statistic, all-reduce, scale launch, message, all-gather,
    unpack + deposit) with a one-rank group: equals the unsharded quantizer bit for bit"""
    import torch.distributed as dist
    from bench import build_quantizer
    from brevitas_amd.core.quant import _fused
    import os
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
    sock.close()
    torch.cuda.set_device(0)
    # twice: the second default group has the first one's name and another communicator -- the node must follow it
    for per_channel in (True, False):
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
        try:
            _sharded_world_of_one(monkeypatch, dist, build_quantizer, _fused, per_channel)
        finally:
            dist.destroy_process_group()


def _sharded_world_of_one(monkeypatch, dist, build_quantizer, _fused, per_channel=True):
    torch.manual_seed(5)
    x = torch.randn(8, 32, 28, 28, device=DEV, dtype=torch.bfloat16)
    g = torch.randn_like(x)
    calls = {'n': 0}
    real = _fused.fast_act_stats_fakequant

    def counted(*a, **k):
        r = real(*a, **k)
        calls['n'] += r is not None
        return r
    monkeypatch.setattr(_fused, 'fast_act_stats_fakequant', counted)
    x[1, 3, 2, 2] = 9.0
    x[5, 7, 1, 1] = -9.0   # two ties of the whole-tensor maximum: they share its gradient
    sharded = _act_steps(build_quantizer(32, per_channel, torch.device(DEV), dist.group.WORLD), x, g, 2, [dict()])
    plain = _act_steps(build_quantizer(32, per_channel, torch.device(DEV)), x, g, 2, [dict()])
    assert calls['n'] == 4
    # the Python Function on the same sharded quantizer (the node switched off): the same bits
    monkeypatch.setattr(_fused, 'fast_act_stats_fakequant', lambda *a, **k: None)
    python = _act_steps(build_quantizer(32, per_channel, torch.device(DEV), dist.group.WORLD), x, g, 2, [dict()])
    monkeypatch.setattr(_fused, 'fast_act_stats_fakequant', counted)
    for a, b in zip(sharded, python):
        for ta, tb, what in zip(a, b, ('y', 'scale', 'dx', 'running')):
            assert torch.equal(_bits(ta), _bits(tb)), ('node vs Python Function', what)
    for a, b in zip(sharded, plain):
        for ta, tb, what in zip(a, b, ('y', 'scale', 'dx', 'running')):
            assert torch.equal(_bits(ta), _bits(tb)), what
