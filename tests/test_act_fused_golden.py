"""Fused activation + quantizer: oracle on CPU, HIP path on GPU, both against the
reference's FusedActivationQuantProxy sequence  tensor_quant(relu(x))  (tests/golden/act_fused.npz),
plus the externally scaled bias quantizers."""
import numpy as np
import pytest
import torch

import golden_util as G
from test_oracle_golden import layout

CASES = G.load('act_fused')
DEV = 'cuda:0'
SUM_RTOL = {'f32': 2e-5, 'bf16': 2.0 ** -6}


def sel(graph):
    cs = [c for c in CASES if c['graph'] == graph]
    return pytest.mark.parametrize('c', cs, ids=G.ids(cs, ['tag', 'dtype', 'signed', 'step']))


def _argmax_positions(xr, chdim):
    from test_gpu_modules import argmax_positions
    return argmax_positions(xr, chdim)


# ---- oracle (CPU) ----------------------------------------------------------------------------------

@sel('relu_runtime_stats')
def test_oracle_relu_runtime_stats(oracle, c):
    """statistic, scale, y and dx of  RescalingIntQuant(relu(x))  with the pre-op folded in"""
    O = oracle
    x = c.arr('x').reshape(-1)
    dt = c.dt('x')
    pc = c['channels']
    outer, ch, inner = (4, pc, 25) if pc else (1, 1, x.size)
    stat = O.stats(O.STAT_ABSMAX, x, dt, outer, ch, inner, pre_op=O.PRE_RELU)
    # scale = clamp_min(stat, 1e-10) / int_threshold in the dtype torch gives it
    xt = c.torch('x')
    st = torch.from_numpy(stat).to(xt.dtype)
    st = torch.clamp_min(st.reshape((1, pc, 1, 1)) if pc else st.reshape(()), 1e-10)
    int_thr = torch.tensor(128.0 if c['signed'] else 255.0)
    scale = st / int_thr
    want_scale = c.torch('scale')
    assert scale.dtype == want_scale.dtype and torch.equal(scale, want_scale)
    sn, sdt = O.from_torch(scale.reshape(-1))
    qmin, qmax = (-128.0, 127.0) if c['signed'] else (0.0, 255.0)
    d = O.make_desc(outer, ch, inner, dt, c.dt('y'), sdt, O.F32, scale_per_channel=pc is not None, qmin=qmin,
                    qmax=qmax, pre_op=O.PRE_RELU)
    zp = np.zeros(1, np.float32)
    y, _ = O.fakequant_fwd(d, x, sn, zp)
    assert G.same_bits(y, c.arr('y').reshape(-1), c['dtypes']['y'])
    dx, ds, _ = O.fakequant_bwd(d, c.arr('g').reshape(-1), x, sn, zp)
    want = c.arr('dx').reshape(-1)
    bad = np.nonzero(dx != want)[0]
    dep = _argmax_positions(torch.relu(xt.float()), 1 if pc else None)
    assert set(bad.tolist()) <= dep, (bad.tolist(), sorted(dep))


@sel('relu_parameter_scale')
def test_oracle_relu_parameter_scale(oracle, c):
    O = oracle
    x = c.arr('x').reshape(-1)
    sn, sdt = O.from_torch(c.torch('scale').reshape(-1))
    qmin, qmax = (-128.0, 127.0) if c['signed'] else (0.0, 255.0)
    d = O.make_desc(1, 1, x.size, c.dt('x'), c.dt('y'), sdt, O.F32, qmin=qmin, qmax=qmax, pre_op=O.PRE_RELU)
    zp = np.zeros(1, np.float32)
    y, _ = O.fakequant_fwd(d, x, sn, zp)
    assert G.same_bits(y, c.arr('y').reshape(-1), c['dtypes']['y'])
    dx, ds, _ = O.fakequant_bwd(d, c.arr('g').reshape(-1), x, sn, zp)
    assert G.same_bits(dx, c.arr('dx').reshape(-1), c['dtypes']['dx'])


# ---- HIP path (GPU) --------------------------------------------------------------------------------

def _quant(c, scaling):
    from test_gpu_modules import mods
    m = mods()
    return m['RescalingIntQuant'](
        m['IntQuant'](narrow_range=False, signed=c['signed'], float_to_int_impl=m['RoundSte'](),
                      tensor_clamp_impl=m['TensorClamp']()),
        scaling, m['IntScaling'](signed=c['signed'], narrow_range=False), m['ZeroZeroPoint'](), m['BitWidthConst'](8))


@pytest.fixture
def cpu_scalar_semantics(monkeypatch):
    import brevitas_amd.config as config
    monkeypatch.setattr(config, 'SCALAR_OPERAND_MODE', 'cpu')


@pytest.mark.gpu
@sel('relu_runtime_stats')
@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'generic'])
def test_gpu_relu_runtime_stats(c, fused, monkeypatch, cpu_scalar_semantics):
    import brevitas_amd.config as config
    from brevitas_amd.proxy import FusedActivationQuantProxy
    from test_gpu_modules import _act_parts, assert_bits, assert_dx, mods
    monkeypatch.setattr(config, 'FUSED_PATHS', fused)
    if not fused and c['channels'] is None and c['dtype'] != 'f32':
        pytest.skip("0-dim float32 scale next to a bf16 tensor: the op-by-op chain runs torch's device kernels, "
                    "which round that scalar to bf16 first; the golden vectors hold the CPU kernels' behaviour")
    m = mods()
    view, stats, shape = _act_parts(c['channels'])
    q = _quant(c, m['RuntimeStatsScaling'](stats, view, m['FloatRestrictValue'](), shape, False, 0.1, 1e-10))
    proxy = FusedActivationQuantProxy(torch.nn.ReLU(), q).to(DEV)
    proxy.train()
    series = [k for k in CASES if k['graph'] == c['graph'] and
              (k['tag'], k['dtype'], k['signed']) == (c['tag'], c['dtype'], c['signed'])]
    for prev in series:
        if prev['step'] >= c['step']:
            break
        proxy(prev.torch('x', DEV))
    x = c.torch('x', DEV).requires_grad_(True)
    y, scale, zp, bw = proxy(x)
    assert_bits(y, c, 'y')
    assert_bits(scale, c, 'scale')
    assert_bits(q.scaling_impl.runtime_stats.running_stats, c, 'running_stats')
    y.backward(c.torch('g', DEV))
    dep = _argmax_positions(torch.relu(x.detach().float()), 1 if c['channels'] else None)
    assert_dx(x.grad, c, x, dep)


@pytest.mark.gpu
@sel('relu_parameter_scale')
def test_gpu_relu_parameter_scale(c, cpu_scalar_semantics):
    from brevitas_amd.proxy import FusedActivationQuantProxy
    from test_gpu_modules import assert_bits, assert_dx, mods
    m = mods()
    q = _quant(c, m['ParameterScaling'](2.5, None, m['FloatRestrictValue'](), 1e-10))
    proxy = FusedActivationQuantProxy(torch.nn.ReLU(), q).to(DEV)
    x = c.torch('x', DEV).requires_grad_(True)
    y, scale, zp, bw = proxy(x)
    assert_bits(y, c, 'y')
    assert_bits(scale, c, 'scale')
    y.backward(c.torch('g', DEV))
    assert_dx(x.grad, c, x)
    want = c.f32('dvalue').reshape(-1)
    got = q.scaling_impl.value.grad.detach().float().cpu().numpy().reshape(-1)
    mag = float(np.abs(c.f32('g')).sum()) * 255 * 2
    np.testing.assert_allclose(got, want, rtol=0, atol=SUM_RTOL[c['dtypes']['y']] * mag)


@pytest.mark.gpu
def test_gpu_prescaled_quantizers():
    """B/core/quant/int.py:17-91 -- doctest and a per-channel bias-like case with gradients"""
    from brevitas_amd.core.bit_width import BitWidthConst
    from brevitas_amd.core.function_wrapper import Identity, RoundSte, TensorClamp
    from brevitas_amd.core.quant import IntQuant, PrescaledRestrictIntQuant, PrescaledRestrictIntQuantWithInputBitWidth
    from test_gpu_modules import assert_bits
    c = [k for k in CASES if k['graph'] == 'prescaled_input_bit_width_doctest'][0]
    q = PrescaledRestrictIntQuantWithInputBitWidth(IntQuant(narrow_range=True, signed=True), Identity()).to(DEV)
    y, scale, zp, bw = q(c.torch('x', DEV), torch.tensor(0.01, device=DEV), torch.tensor(4., device=DEV))
    assert_bits(y, c, 'y')
    assert torch.allclose(y.cpu(), torch.tensor([0.04, -0.05, 0.07, -0.07]), atol=5e-5)
    assert float(zp) == 0.0 and float(bw) == 4.0
    for c in [k for k in CASES if k['graph'] == 'prescaled_bias']:
        q = PrescaledRestrictIntQuant(IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(),
                                               tensor_clamp_impl=TensorClamp()), BitWidthConst(8)).to(DEV)
        b = c.torch('x', DEV).requires_grad_(True)
        s = c.torch('scale', DEV).requires_grad_(True)
        y, so, zp, bw = q(b, s)
        assert_bits(y, c, 'y')
        y.backward(c.torch('g', DEV))
        assert_bits(b.grad, c, 'dx')
        # per-element scale: each dscale is a single term, so even the "reduced" gradient is exact
        assert_bits(s.grad, c, 'dscale')
