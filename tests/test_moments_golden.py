"""Moment statistics (AbsAve, MeanSigmaStd; B/core/stats/stats_op.py:186-240): the oracle's double
sums of |x| and x^2 reproduce the reference's outputs (tests/golden/moments.npz, generated from the
reference's modules) within the rounding of the reference's own dtype -- sums are order-dependent, so
this row is pinned by tolerance, not bit for bit (tolerances below)."""
import numpy as np
import pytest

import golden_util as G

CASES = G.load('moments')
# relative tolerance of a statistic against the reference's value in its dtype
TOL = {'f32': 2e-6, 'bf16': 2.0 ** -8}
# MeanSigmaStd in bf16 rounds var, sqrt, sigma * std and the sum to bf16 one after the other
TOL_CHAIN = {'f32': 4e-6, 'bf16': 2.0 ** -6}


def layout(c):
    shape, dim = c['shape'], c['dim']
    if dim is None:
        return 1, 1, int(np.prod(shape))
    return (1, shape[0], shape[1]) if dim == 1 else (shape[0], shape[1], 1)


def stat_from_sums(c, sums, ch, n):
    mean = sums[:ch] / n
    if c['stat'] == 'abs_ave':
        return mean
    var = (sums[ch:] - sums[:ch] * mean) / (n - 1)
    return mean + 3.0 * np.sqrt(var + 1e-8)


@pytest.fixture(scope='module')
def orc():
    import oracle
    oracle.build()
    return oracle


@pytest.mark.parametrize('c', CASES, ids=G.ids(CASES, ['stat', 'tag', 'dtype']))
def test_oracle_moments_match_reference(orc, c):
    outer, ch, inner = layout(c)
    sums = orc.abs_moments(c.arr('x').reshape(-1), c.dt('x'), outer, ch, inner)
    got = stat_from_sums(c, sums, ch, outer * inner)
    want = c.f32('out').reshape(-1).astype(np.float64)
    tol = (TOL if c['stat'] == 'abs_ave' else TOL_CHAIN)[c['dtype']]
    assert np.all(np.abs(got - want) <= tol * np.abs(want)), (got, want)
