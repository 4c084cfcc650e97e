"""Percentile statistics: oracle on CPU and the HIP radix select on GPU against the
reference's AbsPercentile / NegativePercentileOrZero / PercentileInterval outputs
(tests/golden/percentile.npz, which includes the reference's own tests/brevitas/core/test_stats.py
vectors) and its default Int8ActPerTensorFloat graph.  Values are exact selections: bit-exact."""
import math

import numpy as np
import pytest
import torch

import golden_util as G

CASES = G.load('percentile')
DEV = 'cuda:0'


def sel(stat):
    cs = [c for c in CASES if c['stat'] == stat]
    return pytest.mark.parametrize('c', cs, ids=G.ids(cs, ['tag', 'dtype', 'q', 'dim']))


def layout_of(x, dim):
    if dim is None:
        return 1, 1, x.numel()
    return 1, x.shape[0], x.shape[1]


# ---- oracle --------------------------------------------------------------------------------------------

@sel('abs_percentile')
def test_oracle_abs_percentile(oracle, c):
    x = c.torch('x')
    outer, ch, inner = layout_of(x, c['dim'])
    k = int(math.floor(.01 * c['q'] * (x.numel() if c['dim'] is None else x.shape[1]) + 0.5))
    xn, dt = oracle.from_torch(x.reshape(-1))
    got = oracle.kth_value(xn, dt, outer, ch, inner, k, True)
    assert G.bits_equal(got, c.f32('out').reshape(-1))


@sel('neg_percentile')
def test_oracle_neg_percentile(oracle, c):
    x = c.torch('x')
    outer, ch, inner = layout_of(x, c['dim'])
    k = int(math.ceil(.01 * c['q'] * (x.numel() if c['dim'] is None else x.shape[1])))
    xn, dt = oracle.from_torch(x.reshape(-1))
    got = np.minimum(oracle.kth_value(xn, dt, outer, ch, inner, k, False), np.float32(0.0))
    assert np.array_equal(got, c.f32('out').reshape(-1))


@sel('interval')
def test_oracle_interval(oracle, c):
    x = c.torch('x')
    outer, ch, inner = layout_of(x, c['dim'])
    n = x.numel() if c['dim'] is None else x.shape[1]
    xn, dt = oracle.from_torch(x.reshape(-1))
    lo = oracle.kth_value(xn, dt, outer, ch, inner, int(math.ceil(.01 * c['low_q'] * n)), False)
    hi = oracle.kth_value(xn, dt, outer, ch, inner, int(math.floor(.01 * c['high_q'] * n + 0.5)), False)
    got = torch.abs(torch.from_numpy(hi).to(x.dtype) - torch.from_numpy(lo).to(x.dtype)).float().numpy()
    assert G.bits_equal(got, c.f32('out').reshape(-1))


# ---- HIP path ------------------------------------------------------------------------------------------

@pytest.mark.gpu
@sel('abs_percentile')
def test_gpu_abs_percentile(c):
    from brevitas_amd.core.stats import AbsPercentile
    from test_gpu_modules import assert_bits
    x = c.torch('x', DEV).requires_grad_(c.has('dx'))
    out = AbsPercentile(c['q'], c['dim'])(x)
    assert_bits(out, c, 'out')
    if c.has('dx'):
        gout = c.torch('gout', DEV)
        out.backward(gout)
        dx = x.grad.float().cpu().reshape(x.shape)
        want = c.f32('dx').reshape(x.shape)
        # one element per selected value receives sgn(x) * gout.  Which of several elements with the same
        # |x| gets it is implementation-defined in torch (and +a / -a ties flip the sign), so compare the
        # magnitude of the row sums with the reference and check where the gradient sits.
        rows = dx.reshape(1, -1) if c['dim'] is None else dx
        wrows = torch.from_numpy(want).reshape(rows.shape)
        assert torch.equal(rows.sum(dim=1).abs(), wrows.sum(dim=1).abs())
        assert int((rows != 0).sum(dim=1).max()) <= 1
        nz = rows != 0
        xa = x.detach().float().cpu().abs().reshape(rows.shape)
        o = out.detach().float().cpu().reshape(-1, 1).expand_as(rows)
        assert torch.equal(xa[nz], o[nz])  # the gradient sits on an element attaining the percentile


@pytest.mark.gpu
@sel('neg_percentile')
def test_gpu_neg_percentile(c):
    from brevitas_amd.core.stats import NegativePercentileOrZero
    from test_gpu_modules import assert_bits
    assert_bits(NegativePercentileOrZero(c['q'], c['dim'])(c.torch('x', DEV)), c, 'out')


@pytest.mark.gpu
@sel('interval')
def test_gpu_interval(c):
    from brevitas_amd.core.stats import PercentileInterval
    from test_gpu_modules import assert_bits
    assert_bits(PercentileInterval(c['low_q'], c['high_q'], c['dim'])(c.torch('x', DEV)), c, 'out')


@pytest.mark.gpu
@pytest.mark.parametrize('dn', ['f32', 'bf16'])
def test_gpu_default_int8_act_per_tensor_float(dn, monkeypatch):
    """the reference's default activation quantizer (AbsPercentile 99.999 -> learned scale), 2 collection
    steps then the hand-over"""
    import brevitas_amd.config as config
    from brevitas_amd.core.scaling import ParameterFromRuntimeStatsScaling
    from brevitas_amd.core.stats import AbsPercentile
    from test_gpu_modules import assert_bits, mods
    monkeypatch.setattr(config, 'SCALAR_OPERAND_MODE', 'cpu')
    m = mods()
    q = m['RescalingIntQuant'](
        m['IntQuant'](narrow_range=False, signed=True, float_to_int_impl=m['RoundSte'](),
                      tensor_clamp_impl=m['TensorClamp']()),
        ParameterFromRuntimeStatsScaling(2, AbsPercentile(99.999, None), m['OverTensorView'](), (),
                                         m['FloatRestrictValue'](), 0.1, 1e-10),
        m['IntScaling'](signed=True, narrow_range=False), m['ZeroZeroPoint'](), m['BitWidthConst'](8)).to(DEV)
    q.train()
    for c in [k for k in CASES if k['stat'] == 'int8_act_per_tensor_float' and k['dtype'] == dn]:
        y, scale, zp, bw = q(c.torch('x', DEV))
        assert_bits(scale, c, 'scale')
        assert_bits(y, c, 'y')
        assert_bits(q.scaling_impl.buffer, c, 'buffer')
        assert_bits(q.scaling_impl.value, c, 'value')


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
def test_gpu_percentile_large_vs_torch(dtype):
    """4e7 elements (past the Infinity Cache): the radix select equals torch.kthvalue exactly, per tensor
    and per channel, for |x| and for x"""
    from brevitas_amd import _native as nat
    torch.manual_seed(123456)
    x = torch.randn(64, 128, 70, 70, device=DEV, dtype=dtype)
    n = x.numel()
    for q in (99.999, 50.0):
        k = int(math.floor(.01 * q * n + 0.5))
        got = nat.kth_value(x.reshape(-1), k, 1, 1, n, True)
        want = x.abs().float().reshape(-1).kthvalue(k).values
        assert float(got.float()) == float(want)
    per = 64 * 70 * 70
    k = int(math.floor(.01 * 99.9 * per + 0.5))
    got = nat.kth_value(x.reshape(-1), k, 64, 128, 70 * 70, False)
    want = x.float().permute(1, 0, 2, 3).reshape(128, -1).kthvalue(k, dim=1).values
    assert torch.equal(got.float(), want)
