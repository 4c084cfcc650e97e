"""Statistics reduced over dim 0 of [rows, channels] tensors -- the channel axis LAST ([tokens, hidden], flattened NHWC),
the layout the column-mapped kernels serve -- against the reference's outputs (tests/golden/channel_last.npz, generated
by tests/golden/make_golden.py from /root/reference): AbsPercentile / NegativePercentileOrZero / PercentileInterval
(B/core/stats/stats_op.py:41-126; exact selections: bit-exact) and AbsAve / MeanSigmaStd (:188-279; sums: within the
rounding of the reference's dtype).  CPU: the oracle's k-th value in this layout, and the package's CPU route; device
(marked gpu): the select on the transposed copy (bvq_kth_hist), the column-mapped tie scan of its backward, the
column-mapped moment kernels."""
import math

import numpy as np
import pytest
import torch

import golden_util as G
from test_moments_golden import TOL, TOL_CHAIN

CASES = G.load('channel_last')
DEV = 'cuda:0'


def sel(*stats):
    cs = [c for c in CASES if c['stat'] in stats]
    return pytest.mark.parametrize('c', cs, ids=G.ids(cs, ['stat', 'tag', 'dtype', 'q']))


def _module(c):
    from brevitas_amd.core.stats import AbsAve, AbsPercentile, MeanSigmaStd, NegativePercentileOrZero, PercentileInterval
    return {'abs_percentile': lambda: AbsPercentile(c['q'], 0), 'neg_percentile': lambda: NegativePercentileOrZero(c['q'], 0),
            'interval': lambda: PercentileInterval(c['low_q'], c['high_q'], 0), 'abs_ave': lambda: AbsAve(0),
            'mean_sigma_std': lambda: MeanSigmaStd(c['sigma'], 0)}[c['stat']]()


@sel('abs_percentile', 'neg_percentile')
def test_oracle_kth_value_channel_last(oracle, c):
    x = c.torch('x')
    rows, ch = x.shape
    xn, dt = oracle.from_torch(x.reshape(-1))
    if c['stat'] == 'abs_percentile':
        got = oracle.kth_value(xn, dt, rows, ch, 1, int(math.floor(.01 * c['q'] * rows + 0.5)), True)
        assert G.bits_equal(got, c.f32('out').reshape(-1))
    else:
        got = np.minimum(oracle.kth_value(xn, dt, rows, ch, 1, int(math.ceil(.01 * c['q'] * rows)), False), np.float32(0.0))
        assert np.array_equal(got, c.f32('out').reshape(-1))


@sel('abs_percentile', 'neg_percentile', 'interval', 'abs_ave', 'mean_sigma_std')
def test_cpu_route_channel_last(c):
    out = _module(c)(c.torch('x'))
    want = c.torch('out')
    assert out.shape == want.shape
    assert torch.equal(out.float().nan_to_num(nan=-1.0), want.float().nan_to_num(nan=-1.0))


@pytest.mark.gpu
@sel('abs_percentile', 'neg_percentile', 'interval')
def test_gpu_percentile_channel_last(c):
    from test_gpu_modules import assert_bits
    x = c.torch('x', DEV).requires_grad_(c.has('dx'))
    out = _module(c)(x)
    assert_bits(out, c, 'out')
    if c.has('dx'):
        out.backward(c.torch('gout', DEV))
        dx = x.grad.float().cpu()
        want = torch.from_numpy(c.f32('dx').reshape(tuple(x.shape)))
        # one element per column receives sgn(x) * gout; which of several elements with the same |x| gets it is
        # implementation-defined in torch.kthvalue, so compare the column sums' magnitudes and check where the gradient sits
        assert torch.equal(dx.sum(dim=0).abs().nan_to_num(nan=-1.0), want.sum(dim=0).abs().nan_to_num(nan=-1.0))
        assert int((dx != 0).sum(dim=0).max()) <= 1
        nz = dx != 0
        xa = x.detach().float().cpu().abs()
        o = out.detach().float().cpu().reshape(1, -1).expand_as(xa)
        assert torch.equal(xa[nz], o[nz])


@pytest.mark.gpu
@sel('abs_ave', 'mean_sigma_std')
def test_gpu_moments_channel_last(c):
    x = c.torch('x', DEV).requires_grad_(True)
    out = _module(c).to(DEV)(x)
    want = c.f32('out').reshape(-1)
    tol = dict(TOL if c['stat'] == 'abs_ave' else TOL_CHAIN, f16=(2.0 ** -9 if c['stat'] == 'abs_ave' else 2.0 ** -7))[c['dtype']]
    got = out.detach().float().cpu().numpy().reshape(-1)
    assert np.all(np.abs(got - want) <= tol * np.abs(want) + 1e-30), (got, want)
    out.backward(c.torch('gout', DEV))
    dx, wdx = x.grad.float().cpu().numpy().reshape(-1), c.f32('dx').reshape(-1)
    # (the all-zero channel of a float16 MeanSigmaStd: its epsilon underflows and the reference's own gradient is NaN there)
    ok = np.isfinite(wdx)
    scale = np.abs(wdx[ok]).max()
    assert np.all(np.abs(dx[ok] - wdx[ok]) <= 4 * tol * scale), np.abs(dx[ok] - wdx[ok]).max() / scale
    assert np.all(dx[(c.f32('x').reshape(-1) == 0) & ok] == 0)  # sgn(0) = 0


@pytest.mark.gpu
def test_channel_last_select_takes_the_transposed_copy_and_matches_kthvalue():
    """a larger channel-last tensor: the device select (transposed copy) equals torch.kthvalue per column, and its
    workspace request covers the copy"""
    from brevitas_amd import _native as nat
    torch.manual_seed(5)
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(4099, 520, device=DEV).to(dt)
        for k in (1, 37, 4099 - 8, 4099):
            got = nat.kth_value(x.reshape(-1), k, 4099, 520, 1, True)
            want = x.float().abs().kthvalue(k, dim=0).values.to(dt)
            assert torch.equal(got, want), (dt, k)
            got = nat.kth_value(x.reshape(-1), k, 4099, 520, 1, False)
            assert torch.equal(got, x.float().kthvalue(k, dim=0).values.to(dt)), (dt, k)
        wsb = nat.lib.bvq_kth_workspace_bytes(nat.dtype_code(dt), 4099, 520, 1)
        assert wsb > x.numel() * x.element_size()
