"""Moment statistics on the device (bvq_abs_moments / bvq_abs_affine_bwd; AbsAve, MeanSigmaStd,
MeanLearnedSigmaStd): the kernel's sums against the oracle's double sums (1e-6 relative: float32
partials), the modules against the reference's outputs and gradients (tests/golden/moments.npz) within
the rounding of the reference's dtype."""
import numpy as np
import pytest
import torch

import golden_util as G
from test_moments_golden import TOL, TOL_CHAIN, layout

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = G.load('moments')


@pytest.mark.parametrize('c', CASES, ids=G.ids(CASES, ['stat', 'tag', 'dtype']))
def test_moment_modules_match_reference(c):
    import oracle as O
    from brevitas_amd import _native as nat
    from brevitas_amd.core.stats import AbsAve, MeanSigmaStd
    outer, ch, inner = layout(c)
    x = c.torch('x', DEV)
    n = outer * inner
    mean, var = moments_from_device(nat, x, outer, ch, inner)
    want_sums = O.abs_moments(c.arr('x').reshape(-1), c.dt('x'), outer, ch, inner)
    wmean = want_sums[:ch] / n
    assert np.all(np.abs(mean - wmean) <= 1e-6 * np.abs(wmean))
    if n > 1:
        wvar = (want_sums[ch:] - want_sums[:ch] * wmean) / (n - 1)
        assert np.all(np.abs(var - wvar) <= 1e-5 * np.abs(wvar) + 1e-12)
    mod = AbsAve(c['dim']) if c['stat'] == 'abs_ave' else MeanSigmaStd(3.0, c['dim'])
    xi = x.clone().requires_grad_(True)
    out = mod.to(DEV)(xi)
    want = c.f32('out').reshape(-1)
    assert tuple(out.shape) == tuple(c.arr('out').shape)
    tol = (TOL if c['stat'] == 'abs_ave' else TOL_CHAIN)[c['dtype']]
    got = out.detach().float().cpu().numpy().reshape(-1)
    assert np.all(np.abs(got - want) <= tol * np.abs(want) + 1e-30), (got, want)
    out.backward(c.torch('g', DEV))
    dx, wdx = xi.grad.float().cpu().numpy().reshape(-1), c.f32('dx').reshape(-1)
    # the reference rounds the gradient of every op to the tensor dtype; one rounding here
    scale = np.abs(wdx).max()
    assert np.all(np.abs(dx - wdx) <= 4 * tol * scale), np.abs(dx - wdx).max() / scale
    assert np.all(dx[c.f32('x').reshape(-1) == 0] == 0)  # sgn(0) = 0


def moments_from_device(nat, x, outer, ch, inner):
    """(mean |x|, unbiased var |x|) per channel from bvq_abs_moments' shifted sums, in float64"""
    s = nat.abs_moments(x.reshape(-1), outer, ch, inner).double().cpu().numpy()
    n = outer * inner
    d1, d2, p = s[:ch], s[ch:2 * ch], s[2 * ch:]
    mean = p + d1 / n
    var = (d2 - d1 * d1 / n) / (n - 1) if n > 1 else np.full_like(mean, np.nan)
    return mean, var


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_variance_of_a_large_offset_tensor(dtype):
    """|x| = 100 + 0.01 randn: mean^2 / var = 10^8, where sum x^2 - (sum |x|)^2 / n of float32 sums cancels to
    garbage; the shifted sums keep the variance (yardstick: float64 two-pass on the same values)"""
    from brevitas_amd import _native as nat
    from brevitas_amd.core.stats import MeanSigmaStd
    torch.manual_seed(5)
    x = (100.0 + 0.01 * torch.randn(6, 40, 3000, device=DEV)) * torch.where(torch.rand(6, 40, 3000, device=DEV) < 0.5, -1.0, 1.0)
    if dtype is torch.bfloat16:
        x = (64.0 + torch.randn(6, 40, 3000, device=DEV)).to(dtype)  # bf16 keeps 8 bits: values 62 .. 66 in steps of 0.25/0.5
    else:
        x = x.to(dtype)
    a = x.double().abs()
    for outer, ch, inner, dims in ((6, 40, 3000, (0, 2)), (1, 1, x.numel(), None)):
        mean, var = moments_from_device(nat, x, outer, ch, inner)
        wm = (a.mean(dim=dims) if dims else a.mean()).cpu().numpy().reshape(-1)
        wv = (a.var(dim=dims) if dims else a.var()).cpu().numpy().reshape(-1)
        assert np.all(np.abs(mean - wm) <= 1e-6 * wm)
        assert np.all(np.abs(var - wv) <= 2e-4 * wv), (var, wv)
    out = MeanSigmaStd(3.0, None).to(DEV)(x)
    want = a.mean() + 3.0 * torch.sqrt(a.var() + 1e-8)
    tol = 2e-6 if dtype is torch.float32 else 2.0 ** -6
    assert abs(float(out) - float(want)) <= tol * float(want)
    # a non-finite first element: no shift, and the statistic is what torch gives (inf / nan)
    y = x.clone().float()
    y.view(-1)[0] = float('inf')
    assert np.isinf(moments_from_device(nat, y, 1, 1, y.numel())[0][0])


def test_full_size_moments_and_learned_sigma():
    from brevitas_amd import _native as nat
    from brevitas_amd.core.stats import MeanLearnedSigmaStd
    torch.manual_seed(123456)
    x = torch.randn(64, 512, 56, 56, device=DEV, dtype=torch.bfloat16)
    n = x.numel()
    sums = nat.abs_moments(x.reshape(-1), 1, 1, n)
    mean, var = moments_from_device(nat, x, 1, 1, n)
    a = x.double().abs()
    assert abs(mean[0] - float(a.mean())) <= 1e-6 * float(a.mean()) and abs(var[0] - float(a.var())) <= 1e-5 * float(a.var())
    pm, pv = moments_from_device(nat, x, 64, 512, 56 * 56)
    r1 = a.mean(dim=(0, 2, 3)).cpu().numpy()
    assert np.all(np.abs(pm - r1) <= 1e-6 * r1)
    # determinism
    assert torch.equal(nat.abs_moments(x.reshape(-1), 1, 1, n), sums)
    # learned sigma: a parameter in the graph, state-dict key `sigma` (+ the `learned_sigma` retro key)
    m = MeanLearnedSigmaStd(2.0, (), None).to(DEV)
    xs = x[:1].float().reshape(-1).requires_grad_(True)
    out = m(xs)
    a = xs.detach().abs()
    want = a.mean() + 2.0 * torch.sqrt(a.var() + 1e-8)
    assert abs(float(out) - float(want)) <= 1e-5 * float(want)
    out.backward()
    assert abs(float(m.sigma.grad) - float(torch.sqrt(a.var() + 1e-8))) <= 1e-5
    ref = xs.detach().clone().requires_grad_(True)
    (ref.abs().mean() + 2.0 * torch.sqrt(ref.abs().var() + 1e-8)).backward()
    assert torch.allclose(xs.grad, ref.grad, rtol=1e-4, atol=1e-9)
    m2 = MeanLearnedSigmaStd(1.0, (), None)
    m2.load_state_dict({'learned_sigma': torch.tensor(4.0)})
    assert float(m2.sigma) == 4.0 and list(m2.state_dict().keys()) == ['sigma']


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(300, 512, 1), (257, 64, 1), (65, 1000, 1), (33, 16, 49), (5000, 8, 2)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_column_mapped_moments_equal_row_mapped(dtype, shape):
    """channel-last layouts (short `inner`): bvq_abs_moments and bvq_abs_affine_bwd on the column-mapped units against
    the row-mapped route (a misaligned copy of the same tensor takes it): sums within float32 accumulation error of
    the summed magnitudes, pivots and dx bit for bit"""
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3 + 1).to(dtype).reshape(-1)
    buf = torch.empty(x.numel() + 8, dtype=dtype, device=DEV)
    xm = buf[1:1 + x.numel()]          # 2 or 4 bytes past a 16-byte boundary: the row-mapped route
    xm.copy_(x)
    sc = nat.abs_moments(x, outer, ch, inner).double()
    sr = nat.abs_moments(xm, outer, ch, inner).double()
    assert torch.equal(sc[2 * ch:], sr[2 * ch:])                    # the channels' pivots
    p = sc[2 * ch:].reshape(1, ch, 1)
    d = (x.double().reshape(outer, ch, inner).abs() - p)
    mag1, mag2 = d.abs().sum(dim=(0, 2)), (d * d).sum(dim=(0, 2))
    eps = 2.0 ** -22
    assert torch.all((sc[:ch] - sr[:ch]).abs() <= eps * 8 * mag1 + 1e-30)
    assert torch.all((sc[ch:2 * ch] - sr[ch:2 * ch]).abs() <= eps * 8 * mag2 + 1e-30)
    assert torch.all((sc[:ch] - d.sum(dim=(0, 2))).abs() <= eps * 8 * mag1 + 1e-30)   # and against float64
    a = torch.randn(ch, device=DEV)
    b = torch.randn(ch, device=DEV)
    dc = nat.abs_affine_bwd(x, a, b, outer, ch, inner)
    dr = nat.abs_affine_bwd(xm, a, b, outer, ch, inner)
    assert torch.equal(dc.view(torch.int16 if dc.element_size() == 2 else torch.int32),
                       dr.view(torch.int16 if dr.element_size() == 2 else torch.int32))
