"""Asymmetric ("shifted") quantizers: integer zero-point from statistics.
brevitas_amd.quant.ShiftedUint8{Weight,Act}* against the reference's resolved graphs
(tests/golden/shifted.npz): scale, zero-point, y bit-exact; dx bit-exact away from the elements that
receive a statistic's gradient (the tensor min / max of each channel, or the percentile elements)."""
import numpy as np
import pytest
import torch

import golden_util as G
from test_gpu_modules import assert_bits, to_np

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = G.load('shifted')


@pytest.fixture(autouse=True)
def cpu_scalar_semantics(monkeypatch):
    import brevitas_amd.config as config
    monkeypatch.setattr(config, 'SCALAR_OPERAND_MODE', 'cpu')


def _dx_check(dx, c, max_deposits):
    got, want = to_np(dx).reshape(-1), c.arr('dx').reshape(-1)
    gf = dx.detach().float().cpu().numpy().reshape(-1)
    wf = c.f32('dx').reshape(-1)
    # (a NaN gradient -- log2 of an all-zero statistic, as in the reference -- equals a NaN gradient)
    bad = np.nonzero((got != want) & ~(np.isnan(gf) & np.isnan(wf)))[0]
    assert bad.size <= max_deposits, (bad.size, max_deposits)
    scale = max(1.0, float(np.abs(wf).max()))
    tol = {'f32': 2e-4, 'bf16': 2.0 ** -5, 'f16': 2.0 ** -8}[c['dtypes']['dx']] * 64 * scale
    assert np.all(np.abs(gf[bad] - wf[bad]) <= tol), (gf[bad], wf[bad])


@pytest.mark.parametrize('c', [k for k in CASES if k['graph'] == 'shifted_weight'],
                         ids=lambda c: '%s-%s' % (c['tag'], c['dtype']))
def test_shifted_weight(c):
    import brevitas_amd.quant as Q
    w = torch.nn.Parameter(c.torch('x', DEV))
    build = Q.ShiftedUint8WeightPerChannelFloat if c['tag'] == 'per_channel' else Q.ShiftedUint8WeightPerTensorFloat
    q = build(w).to(DEV)
    y, scale, zp, bw = q(w)
    assert_bits(scale, c, 'scale')
    assert_bits(zp, c, 'zp')
    assert_bits(y, c, 'y')
    assert float(zp.min()) >= 0 and float(zp.max()) <= 255
    y.backward(c.torch('g', DEV))
    channels = w.shape[0] if c['tag'] == 'per_channel' else 1
    _dx_check(w.grad, c, 2 * channels + 2)  # the min and the max element of every channel


@pytest.mark.parametrize('dn', ['f32', 'bf16'])
def test_shifted_act(dn):
    import brevitas_amd.quant as Q
    q = Q.ShiftedUint8ActPerTensorFloat(collect_stats_steps=2).to(DEV)
    q.train()
    for c in [k for k in CASES if k['graph'] == 'shifted_act' and k['dtype'] == dn]:
        x = c.torch('x', DEV).requires_grad_(True)
        q.zero_grad()
        y, scale, zp, bw = q(x)
        assert_bits(scale, c, 'scale')
        assert_bits(zp, c, 'zp')
        assert_bits(y, c, 'y')
        assert_bits(q.zero_point_impl.buffer, c, 'zp_buffer')
        assert_bits(q.zero_point_impl.value, c, 'zp_value')
        assert_bits(q.scaling_impl.value, c, 'scale_value')
        y.backward(c.torch('g', DEV))
        _dx_check(x.grad, c, 4)  # collection phase: three percentile elements receive a gradient
    want_keys = [k for k in CASES if k['graph'] == 'shifted_act_state_dict' and k['dtype'] == dn][0]['keys']
    assert sorted(q.state_dict().keys()) == want_keys
