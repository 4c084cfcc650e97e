"""Learned bit widths (B/core/bit_width/parameter.py, const.py:43-66) against the
reference's resolved graphs (tests/golden/learned_bw.npz; float32, bfloat16, float16).  The bit width is a
tensor in the autograd graph: the fused quantizer kernels read the integer range from device memory and
return the range's own gradient (bvq_fakequant_fwd_bounds / bvq_fakequant_bwd_bounds), so the quantizer is
still one forward and one backward pass: y, scale, bit width and dx bit-exact, the reduced gradients
(bit-width offset, learned scale) within the rounding of a reduced sum.  `generic`: the op-by-op route."""
import numpy as np
import pytest
import torch

import golden_util as G
from test_gpu_modules import assert_bits
from test_gpu_shifted import _dx_check

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = G.load('learned_bw')


@pytest.fixture(autouse=True)
def cpu_scalar_semantics(monkeypatch):
    import brevitas_amd.config as config
    monkeypatch.setattr(config, 'SCALAR_OPERAND_MODE', 'cpu')


def _close(got, c, name, rel):
    got = float(got)
    want = float(c.f32(name))
    assert abs(got - want) <= rel * max(1.0, abs(want)), (name, got, want)


def _quant(graph, weight=None, bits=4):
    from brevitas_amd.core.bit_width import BitWidthParameter
    from brevitas_amd.core.function_wrapper import OverOutputChannelView, RoundSte, TensorClamp, TensorClampSte
    from brevitas_amd.core.quant import IntQuant, RescalingIntQuant
    from brevitas_amd.core.restrict_val import FloatRestrictValue
    from brevitas_amd.core.scaling import IntScaling, ParameterScaling, StatsFromParameterScaling
    from brevitas_amd.core.stats import AbsMax
    from brevitas_amd.core.zero_point import ZeroZeroPoint
    if graph == 'weight':
        return RescalingIntQuant(
            IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
            StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, [weight], FloatRestrictValue(),
                                      (6, 1, 1, 1), False, 1e-10),
            IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthParameter(4)).to(DEV)
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        ParameterScaling(1.5, None, FloatRestrictValue(), 1e-10),
        IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthParameter(bits)).to(DEV)


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'generic'])
@pytest.mark.parametrize('c', [k for k in CASES if k['graph'] == 'weight'], ids=lambda c: c['dtype'])
def test_weight_learned_bit_width(c, fused, monkeypatch):
    import brevitas_amd.config as config
    monkeypatch.setattr(config, 'FUSED_PATHS', fused)
    w = torch.nn.Parameter(c.torch('x', DEV))
    q = _quant('weight', w)
    y, scale, zp, bw = q(w)
    assert bw.requires_grad
    assert_bits(bw, c, 'bit_width')
    assert_bits(scale, c, 'scale')
    if c['dtype'] == 'f32' or fused:
        assert_bits(y, c, 'y')
    y.backward(c.torch('g', DEV))
    if c['dtype'] == 'f32' or fused:
        _dx_check(w.grad, c, 6)
    _close(q.msb_clamp_bit_width_impl.bit_width_offset.grad, c, 'doffset', 2e-3 if c['dtype'] == 'f32' else 6e-2)


@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'generic'])
@pytest.mark.parametrize('c', [k for k in CASES if k['graph'] == 'act'],
                         ids=lambda c: '%d-%s' % (c['bits'], c['dtype']))
def test_act_learned_bit_width(c, fused, monkeypatch):
    import brevitas_amd.config as config
    monkeypatch.setattr(config, 'FUSED_PATHS', fused)
    q = _quant('act', bits=c['bits'])
    x = c.torch('x', DEV).requires_grad_(True)
    y, scale, zp, bw = q(x)
    assert_bits(bw, c, 'bit_width')
    assert_bits(scale, c, 'scale')
    if c['dtype'] == 'f32' or fused:
        assert_bits(y, c, 'y')
    y.backward(c.torch('g', DEV))
    if c['dtype'] == 'f32' or fused:
        assert_bits(x.grad, c, 'dx')
    tol = 2e-3 if c['dtype'] == 'f32' else 6e-2
    _close(q.msb_clamp_bit_width_impl.bit_width_offset.grad, c, 'doffset', tol)
    _close(q.scaling_impl.value.grad, c, 'dvalue', tol)


def test_bit_width_modules():
    from brevitas_amd.core.bit_width import BitWidthParameter, MsbClampBitWidth, RemoveBitwidthParameter
    c = [k for k in CASES if k['graph'] == 'bit_width_parameter'][0]
    bw = BitWidthParameter(5, min_bit_width=3).to(DEV)
    assert_bits(bw.bit_width_offset, c, 'offset')
    out = bw()
    out.backward()
    assert_bits(out, c, 'out')
    assert_bits(bw.bit_width_offset.grad, c, 'doffset')
    for c in [k for k in CASES if k['graph'] == 'msb_clamp']:
        rm = RemoveBitwidthParameter(c['remove']).to(DEV)
        msb = MsbClampBitWidth(rm, 2, 16).to(DEV)
        assert_bits(rm.bit_width_coeff, c, 'coeff')
        inp = torch.tensor(9.0, device=DEV, requires_grad=True)
        out = msb(inp)
        out.backward()
        assert_bits(out, c, 'out')
        assert_bits(inp.grad, c, 'dinp')
        assert np.allclose(rm.bit_width_coeff.grad.cpu().numpy(), c.f32('dcoeff'), rtol=1e-5)
    with pytest.raises(RuntimeError):
        BitWidthParameter(1)
    with pytest.raises(RuntimeError):
        BitWidthParameter(4, min_bit_width=6)
