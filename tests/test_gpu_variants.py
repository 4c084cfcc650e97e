"""Quantizer variants of the same elementwise family: binary, clamped binary,
ternary, decoupled and truncating quantizers against the reference (tests/golden/variants.npz).
They run op by op on the HIP-backed straight-through ops: y and dx bit-exact."""
import numpy as np
import pytest
import torch

import golden_util as G
from test_gpu_modules import assert_bits

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = G.load('variants')


def pick(name):
    return [c for c in CASES if c['quant'] == name]


@pytest.mark.parametrize('c', pick('binary') + pick('clamped_binary') + pick('ternary'),
                         ids=lambda c: '%s-%s' % (c['quant'], c['dtype']))
def test_sign_quantizers(c):
    from brevitas_amd.core.quant import BinaryQuant, ClampedBinaryQuant, TernaryQuant
    from brevitas_amd.core.scaling import ParameterScaling
    q = {'binary': lambda: BinaryQuant(ParameterScaling(0.7)),
         'clamped_binary': lambda: ClampedBinaryQuant(ParameterScaling(0.7)),
         'ternary': lambda: TernaryQuant(ParameterScaling(0.9), 0.5)}[c['quant']]().to(DEV)
    x = c.torch('x', DEV).requires_grad_(True)
    y, scale, zp, bw = q(x)
    assert_bits(y, c, 'y')
    assert_bits(scale, c, 'scale')
    assert float(zp) == float(c.f32('zp')) and float(bw) == float(c.f32('bit_width'))
    y.backward(c.torch('g', DEV))
    if c['dtype'] == 'f32':
        assert_bits(x.grad, c, 'dx')
    else:
        # dx = g * scale with a 0-dim float32 scale next to a bf16 tensor: torch's device kernels round
        # the scalar to bf16 first, the CPU kernels (golden) do not -- one bf16 ulp apart at most
        got, ref = x.grad.float().cpu(), c.torch('dx').float()
        assert bool(((got - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-30).all())
    want = c.f32('dvalue').reshape(-1)
    got = q.scaling_impl.value.grad.float().cpu().numpy().reshape(-1)
    np.testing.assert_allclose(got, want, rtol=2e-2 if c['dtype'] == 'bf16' else 1e-5, atol=1e-6)


@pytest.mark.parametrize('c', pick('decoupled'), ids=lambda c: c['dtype'])
def test_decoupled(c):
    from brevitas_amd.core.quant import DecoupledIntQuant
    dq = DecoupledIntQuant(narrow_range=True, signed=True).to(DEV)
    x = c.torch('x', DEV).requires_grad_(True)
    t = lambda v: torch.tensor(v, device=DEV)  # noqa: E731
    y = dq(c.torch('pre_scale', DEV), t(0.), c.torch('scale', DEV), t(0.), t(4.), x)
    if c['dtype'] == 'f32':
        assert_bits(y, c, 'y')
        y.backward(c.torch('g', DEV))
        assert_bits(x.grad, c, 'dx')
    else:  # 0-dim float32 scales next to a bf16 tensor: torch's device kernels round them to bf16 first
        assert torch.allclose(y.float().cpu(), c.torch('y').float(), atol=0.02)


@pytest.mark.parametrize('c', pick('trunc'), ids=lambda c: '%s-%s' % (c['round'], c['dtype']))
def test_trunc(c):
    from brevitas_amd.core.bit_width import BitWidthConst
    from brevitas_amd.core.function_wrapper import FloorSte, RoundSte
    from brevitas_amd.core.quant import TruncIntQuant
    tq = TruncIntQuant({'floor': FloorSte, 'round': RoundSte}[c['round']](), BitWidthConst(5)).to(DEV)
    x = c.torch('x', DEV).requires_grad_(True)
    y, scale, zp, bw = tq(x, torch.tensor(0.05, device=DEV), torch.tensor(0., device=DEV), torch.tensor(8., device=DEV))
    assert float(bw) == 5.0
    if c['dtype'] == 'f32':
        assert_bits(y, c, 'y')
        y.backward(c.torch('g', DEV))
        assert_bits(x.grad, c, 'dx')
    else:
        assert torch.allclose(y.float().cpu(), c.torch('y').float(), atol=0.06)


def test_doctests():
    from brevitas_amd.core.quant import DecoupledIntQuant, TernaryQuant
    from brevitas_amd.core.scaling import ConstScaling
    t = lambda v: torch.tensor(v, device=DEV)  # noqa: E731
    c = pick('decoupled_doctest')[0]
    y = DecoupledIntQuant(narrow_range=True, signed=True).to(DEV)(t(0.02), t(0.), t(0.01), t(0.), t(4.), c.torch('x', DEV))
    assert_bits(y, c, 'y')
    assert torch.allclose(y.cpu(), torch.tensor([0.02, -0.03, 0.07, -0.07]), atol=5e-5)
    c = pick('ternary_doctest')[0]
    y, scale, zp, bw = TernaryQuant(ConstScaling(1.0), 0.5).to(DEV)(c.torch('x', DEV))
    assert y.tolist() == [0.0, -1.0, 1.0] and float(scale) == 1.0 and float(zp) == 0.0 and float(bw) == 2.0
