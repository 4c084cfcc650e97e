"""GPU parity of the function / module layer of seam 1 and of the statistics modules, including
autograd, against the reference's golden vectors (restating tests/brevitas/function/
test_autograd_ste_ops.py:26-190: straight-through backward gives x.grad == grad exactly; clamp
variants pass the gradient to x only; abs_binary_sign_grad has subgradient 1 at 0)."""
import numpy as np
import pytest
import torch

import golden_util as G
from test_gpu_modules import assert_bits, to_np

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

STE = [c for c in G.load('ste_ops') if c['op'] != 'int_range']


@pytest.mark.parametrize('c', STE, ids=G.ids(STE, ['op', 'dtype', 'bounds']))
def test_ops_ste_functions_with_autograd(c):
    from brevitas_amd.function import ops as F
    from brevitas_amd.function import ops_ste as S
    op, dn = c['op'], c['dtype']
    x = c.torch('x', DEV).requires_grad_(op != 'tensor_clamp_ste_')
    if op in ('round_ste', 'floor_ste', 'ceil_ste', 'round_to_zero_ste', 'dpu_round_ste', 'binary_sign_ste',
              'ternary_sign_ste', 'abs_binary_sign_grad'):
        y = getattr(S, op)(x)
    elif op == 'scalar_clamp_ste':
        y = S.scalar_clamp_ste(x, c['lo'], c['hi'])
    elif op == 'scalar_clamp_min_ste':
        y = S.scalar_clamp_min_ste(x, c['lo'])
    elif op == 'tensor_clamp_ste':
        lo, hi = c.torch('lo', DEV).requires_grad_(True), c.torch('hi', DEV).requires_grad_(True)
        y = S.tensor_clamp_ste(x, lo, hi)
    elif op == 'tensor_clamp':
        y = F.tensor_clamp(x, c.torch('lo', DEV), c.torch('hi', DEV))
    elif op == 'tensor_clamp_ste_':
        xin = x.clone()
        y = S.tensor_clamp_ste_(xin, c.torch('lo', DEV), c.torch('hi', DEV))
        assert y.data_ptr() == xin.data_ptr()  # in place
    assert_bits(y, c, 'y')
    if c.has('dx'):
        y.backward(c.torch('g', DEV))
        assert_bits(x.grad, c, 'dx')
        if op == 'tensor_clamp_ste':
            assert lo.grad is None and hi.grad is None  # gradient to x only


def test_plain_ops_doctests():
    """B/function/ops.py:27-29,47-49,67-69,94-96 on the device"""
    from brevitas_amd.function import ops as F
    t = lambda v: torch.tensor(v, device=DEV)  # noqa: E731
    assert F.binary_sign(t([2.1, -0.3, 0.0])).tolist() == [1.0, -1.0, 1.0]
    assert F.round_to_zero(t([-1.5, -0.5, 0.5, 1.5])).tolist() == [-1.0, -0.0, 0.0, 1.0]
    assert F.dpu_round(t([-1.5, -0.5, 0.5, 1.5])).tolist() == [-1.0, -0.0, 0.0, 2.0]
    assert torch.allclose(F.tensor_clamp(t([1.7, -0.5, 0.1]), t(0.0), t(1.0)).cpu(), torch.tensor([1.0, 0.0, 0.1]))
    x = t([1.7, -1.7]).requires_grad_(True)
    assert F.round_to_zero(x).sum().backward() is None and x.grad.tolist() == [0.0, 0.0]
    xin = t([3.0, -3.0, 0.5])
    assert F.tensor_clamp_(xin, t(-2.0), t(2.0)) is xin and xin.tolist() == [2.0, -2.0, 0.5]


STATS = G.load('stats')


@pytest.mark.parametrize('c', STATS, ids=G.ids(STATS, ['stat', 'dtype', 'tag', 'chdim']))
def test_stats_modules_with_autograd(c):
    """AbsMax / AbsMinMax modules behind the reference's views, forward and backward"""
    from brevitas_amd.core.function_wrapper import OverOutputChannelView, OverTensorView
    from brevitas_amd.core.stats import AbsMax, AbsMinMax
    x = c.torch('x', DEV).requires_grad_(True)
    chdim = c['chdim']
    if chdim is None:
        view, dim = OverTensorView(), None
    elif chdim == 0:
        view, dim = OverOutputChannelView(None), 1
    else:
        perm = (1, 0) + tuple(range(2, x.dim()))
        view, dim = OverOutputChannelView(perm), 1
    mod = AbsMax(dim) if c['stat'] == 'absmax' else AbsMinMax(dim)
    out = mod(view(x))
    assert_bits(out, c, 'out')
    if np.isnan(c.f32('out')).any():
        return
    out.backward(c.torch('gout', DEV))
    want, got = c.f32('dx').reshape(-1), x.grad.float().cpu().numpy().reshape(-1)
    if c['stat'] == 'absmax' and chdim in (None, 0):
        assert_bits(x.grad, c, 'dx')  # no permuted copy in between: even the zero signs match
    else:
        assert np.array_equal(got, want)  # values (autograd of the permute drops the sign of zeros)


def test_parameter_list_stats_concatenates_tracked_weights():
    """two layers sharing one weight quantizer: statistics over the concatenation
    (B/core/stats/stats_wrapper.py:83-114), generic route"""
    from test_gpu_modules import mods
    m = mods()
    torch.manual_seed(123456)
    w1 = torch.nn.Parameter(torch.randn(6, 4, 3, 3, device=DEV) * 0.1)
    w2 = torch.nn.Parameter(torch.randn(6, 2, 3, 3, device=DEV) * 0.3)
    q = m['RescalingIntQuant'](
        m['IntQuant'](narrow_range=True, signed=True, float_to_int_impl=m['RoundSte'](),
                      tensor_clamp_impl=m['TensorClampSte']()),
        m['StatsFromParameterScaling'](m['AbsMax'](1), m['OverOutputChannelView'](None), 1, [w1, w2],
                                       m['FloatRestrictValue'](), (6, 1, 1, 1), False, 1e-10),
        m['IntScaling'](True, True), m['ZeroZeroPoint'](), m['BitWidthConst'](8)).to(DEV)
    y, scale, zp, bw = q(w1)
    want_stat = torch.maximum(w1.detach().reshape(6, -1).abs().amax(1), w2.detach().reshape(6, -1).abs().amax(1))
    # (a tensor divisor: dividing by a python scalar multiplies by its reciprocal on the device)
    assert torch.equal(scale.reshape(-1), want_stat / torch.tensor(127.0, device=DEV))
    ref = torch.round(w1.detach() / scale) * scale
    assert torch.equal(y, ref)
    y.sum().backward()
    assert w1.grad is not None and w2.grad is not None  # the statistic's gradient reaches both
