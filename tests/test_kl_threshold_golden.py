"""KLMinimizerThreshold (B/core/stats/stats_op.py:280-350) against the reference's own module
(tests/golden/kl_threshold.npz: inputs, the reference's histogram and its selected threshold).  On CPU tensors the
module takes the pure-torch route for the histogram; the threshold search is the module's own restatement of the
reference's loop -- pinned here bit for bit on every case (signed / unsigned, 4 / 6 / 8 bits, Gaussian, heavy-tailed,
post-ReLU and sparse inputs)."""
import pytest
import torch

import golden_util as G

CASES = G.load('kl_threshold')


@pytest.mark.parametrize('c', CASES, ids=lambda c: c['tag'])
def test_kl_threshold_matches_reference_on_cpu(c):
    from brevitas_amd.core.bit_width import BitWidthConst
    from brevitas_amd.core.stats import KLMinimizerThreshold
    m = KLMinimizerThreshold(c['signed'], BitWidthConst(c['bits']))
    x = c.torch('x')
    assert torch.equal(m._histogram(x, x.abs().max()), c.torch('hist'))
    out = m(x)
    assert out.dtype == torch.float32 and out.dim() == 0
    assert float(out) == float(c.f32('out')), (float(out), float(c.f32('out')))


@pytest.mark.gpu
@pytest.mark.parametrize('c', CASES, ids=lambda c: c['tag'])
def test_kl_threshold_on_the_device(c):
    """device route: abs-max and histogram kernels (one read of x each, range read from device memory), search on the
    host.  The histogram equals torch.histc's on the same device; the threshold is the reference's."""
    from brevitas_amd import _native as nat
    from brevitas_amd.core.bit_width import BitWidthConst
    from brevitas_amd.core.stats import KLMinimizerThreshold
    dev = 'cuda:0'
    x = c.torch('x', dev)
    a = x.abs().max()
    got = nat.histc(x, a, 1001)
    want = torch.histc(x, bins=1001, min=-float(a), max=float(a)).int()
    assert int(got.sum()) == x.numel()
    assert int((got.cpu() - want.cpu()).abs().sum()) <= 4      # edge elements: float32 op order of the bin formula
    m = KLMinimizerThreshold(c['signed'], BitWidthConst(c['bits'])).to(dev)
    out = m(x)
    assert out.is_cuda and abs(float(out) - float(c.f32('out'))) <= 1e-6 * abs(float(c.f32('out')))
    # a view into the middle of a buffer (not 16-byte aligned): the module must serve it like the reference does
    buf = torch.empty(x.numel() + 4, device=dev)
    xv = buf[1:1 + x.numel()]
    xv.copy_(x.reshape(-1))
    assert xv.data_ptr() % 16 != 0
    out_v = m(xv.reshape(x.shape))
    assert abs(float(out_v) - float(out)) <= 1e-6 * abs(float(out))
    for dt in (torch.bfloat16, torch.float16):                 # 16-bit inputs: same kernel family
        xh = x.to(dt)
        ah = xh.abs().max()
        gh = nat.histc(xh, ah, 1001)
        wh = torch.histc(xh.float(), bins=1001, min=-float(ah), max=float(ah)).int()
        assert int(gh.sum()) == x.numel() and int((gh.cpu() - wh.cpu()).abs().sum()) <= 8
