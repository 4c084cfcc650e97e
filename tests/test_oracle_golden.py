"""Pins the CPU oracle (oracle/bvq_oracle.c) against the reference.

Two sources, both produced by the reference itself:
  * tests/golden/*.npz -- outputs of the reference's own modules/functions run on CPU by
    tests/golden/make_golden.py (inputs from its test seed 123456);
  * the known-answer examples the reference carries in its docstrings and tests
    (SURVEY 8c "Golden vectors"), restated here as literals with their file:line.

Bar: bit-exact for every elementwise result (integer codes, dequantized values, dx); the reduced
scale / zero-point gradients are order- and precision-dependent in the reference (a torch.sum of
ct-rounded products, itself rounded to ct) and are compared within a tolerance derived from the
magnitude of the summed terms.
"""
import numpy as np
import pytest

import golden_util as G

RM = {'round': 0, 'floor': 1, 'ceil': 2, 'rtz': 3, 'dpu': 4}
EPS = {'f32': 2.0 ** -20, 'bf16': 2.0 ** -6, 'f16': 2.0 ** -9}  # a few ulps of the summed magnitude


def int_range(signed, narrow, bw):
    """B/function/ops.py:132-191"""
    if signed:
        return float(-(2 ** (bw - 1)) + (1 if narrow else 0)), float(2 ** (bw - 1) - 1)
    return 0.0, float(2 ** bw - 1 - (1 if narrow else 0))


def layout(shape, chdim):
    if chdim is None:
        return 1, 1, int(np.prod(shape))
    return int(np.prod(shape[:chdim])), int(shape[chdim]), int(np.prod(shape[chdim + 1:]))


def desc_for(orc, c, **over):
    outer, ch, inner = layout(c['shape'], c['chdim'])
    qmin, qmax = int_range(c['signed'], c['narrow'], c['bit_width'])
    kw = dict(scale_per_channel=c.arr('scale').size > 1, zp_per_channel=c.arr('zp').size > 1, qmin=qmin,
              qmax=qmax, round_mode=RM[c['round']], scalar_mode=orc.SCALAR_OPMATH,
              clamp_ste=c['clamp'] == 'ste')
    kw.update(over)
    return orc.make_desc(outer, ch, inner, c.dt('x'), c.dt('y'), c.dt('scale'), c.dt('zp'), **kw)


def grad_sum_tolerance(c, which):
    """magnitude of the terms the reference sums for dscale / dzp, times a few ulps of ct"""
    x = c.f32('x').astype(np.float64).reshape(c['shape'])
    g = c.f32('g').astype(np.float64).reshape(c['shape'])
    s = np.broadcast_to(c.f32('scale').astype(np.float64).reshape(
        c.arr('scale').shape if c.arr('scale').ndim else ()), x.shape)
    qmin, qmax = int_range(c['signed'], c['narrow'], c['bit_width'])
    zmax = float(np.max(np.abs(c.f32('zp'))))
    if which == 'dscale':
        with np.errstate(all='ignore'):
            terms = np.abs(g) * (max(abs(qmin), abs(qmax)) + zmax + np.abs(x / s))
    else:
        terms = 2 * np.abs(g * s)
    ref = c.f32(which)
    if ref.size > 1:
        axes = tuple(i for i in range(x.ndim) if i != c['chdim'])
        mag = terms.sum(axis=axes).reshape(-1)
    else:
        mag = np.array([terms.sum()])
    return EPS[c['dtypes']['y']] * mag + 1e-30


INT_QUANT = G.load('int_quant')


@pytest.mark.parametrize('c', INT_QUANT, ids=G.ids(INT_QUANT, ['x_dtype', 'layout', 'round', 'clamp', 'bit_width']))
def test_int_quant_forward_bit_exact(oracle, c):
    d = desc_for(oracle, c)
    args = (c.arr('x').reshape(-1), c.arr('scale').reshape(-1), c.arr('zp').reshape(-1))
    y, codes = oracle.fakequant_fwd(d, *args)
    assert G.same_bits(y, c.arr('y').reshape(-1), c['dtypes']['y']), G.mismatch_report(y, c.arr('y'), 0)
    d.out_kind = oracle.OUT_INT
    yi, _ = oracle.fakequant_fwd(d, *args)
    assert G.same_bits(yi, c.arr('y_int').reshape(-1), c['dtypes']['y_int'])
    # integer codes = the float-encoded integers of to_int
    want = c.f32('y_int').reshape(-1)
    fin = np.isfinite(want)
    assert np.array_equal(codes[fin], want[fin].astype(np.int32))
    qmin, qmax = int_range(c['signed'], c['narrow'], c['bit_width'])
    assert codes[fin].min() >= qmin and codes[fin].max() <= qmax


@pytest.mark.parametrize('c', INT_QUANT, ids=G.ids(INT_QUANT, ['x_dtype', 'layout', 'round', 'clamp', 'bit_width']))
def test_int_quant_backward(oracle, c):
    d = desc_for(oracle, c)
    dx, ds, dz = oracle.fakequant_bwd(d, c.arr('g').reshape(-1), c.arr('x').reshape(-1),
                                      c.arr('scale').reshape(-1), c.arr('zp').reshape(-1))
    assert G.same_bits(dx, c.arr('dx').reshape(-1), c['dtypes']['dx']), G.mismatch_report(dx, c.arr('dx'), 0)
    for got, name in ((ds, 'dscale'), (dz, 'dzp')):
        want = c.f32(name).reshape(-1).astype(np.float64)
        got = got.astype(np.float64)
        if got.size != want.size:  # gradient of a 0-dim operand next to a per-channel one
            got = np.array([got.sum()])
        tol = grad_sum_tolerance(c, name)
        both_nan = np.isnan(got) & np.isnan(want)
        both_inf = np.isinf(got) & np.isinf(want)
        ok = both_nan | both_inf | ~np.isfinite(tol) | (np.abs(got - want) <= tol)
        assert ok.all(), (name, got, want, tol)


STATS = G.load('stats')


@pytest.mark.parametrize('c', STATS, ids=G.ids(STATS, ['stat', 'dtype', 'tag', 'chdim']))
def test_stats(oracle, c):
    outer, ch, inner = layout(c['shape'], c['chdim'])
    dt = c.dt('x')
    x = c.arr('x').reshape(-1)
    if c['stat'] == 'absmax':
        out = oracle.stats(oracle.STAT_ABSMAX, x, dt, outer, ch, inner)
        want = c.f32('out').reshape(-1)
        assert G.bits_equal(out, want)
        if not np.isnan(want).any():
            stat = c.arr('out').reshape(-1)
            dx = oracle.absmax_bwd(x, stat, c.arr('gout').reshape(-1), dt, outer, ch, inner)
            assert G.same_bits(dx, c.arr('dx').reshape(-1), c['dtypes']['dx']), \
                G.mismatch_report(dx, c.arr('dx'), 0)
    else:
        out = oracle.stats(oracle.STAT_MINMAX, x, dt, outer, ch, inner)
        # AbsMinMax.forward: abs(max - min) in the tensor dtype (B/core/stats/stats_op.py:152-158)
        import oracle as O
        diff = out[:ch] - out[ch:]
        store = np.abs(diff).astype(np.float32)
        if dt == O.F32:
            got = store
        else:
            import torch
            tdt = torch.bfloat16 if dt == O.BF16 else torch.float16
            got = torch.from_numpy(store).to(tdt).float().numpy()
        assert G.bits_equal(got, c.f32('out').reshape(-1))


STE = G.load('ste_ops')
UNARY = {'round_ste': 0, 'floor_ste': 1, 'ceil_ste': 2, 'round_to_zero_ste': 3, 'dpu_round_ste': 4,
         'binary_sign_ste': 5, 'ternary_sign_ste': 6, 'abs_binary_sign_grad': 7}


@pytest.mark.parametrize('c', STE, ids=G.ids(STE, ['op', 'dtype', 'bounds']))
def test_ste_ops(oracle, c):
    op = c['op']
    if op == 'int_range':
        qmin, qmax = int_range(c['signed'], c['narrow'], c['bit_width'])
        assert float(c.arr('min_int')) == qmin and float(c.arr('max_int')) == qmax
        return
    dt = c.dt('x')
    dn = c['dtype']
    x = c.arr('x')
    if op in UNARY:
        y = oracle.unary(UNARY[op], x, dt)
        assert G.same_bits(y, c.arr('y'), dn), G.mismatch_report(y, c.arr('y'), dn)
        if op == 'abs_binary_sign_grad':
            dx = oracle.abs_binary_sign_grad_bwd(c.arr('g'), x, dt)
            assert G.same_bits(dx, c.arr('dx'), dn)
        else:  # straight-through: x.grad == grad exactly (tests/brevitas/function/test_autograd_ste_ops.py:53-63)
            assert G.same_bits(c.arr('dx'), c.arr('g'), dn)
    elif op == 'scalar_clamp_ste':
        y = oracle.scalar_clamp(x, dt, c['lo'], c['hi'])
        assert G.same_bits(y, c.arr('y'), dn)
        assert G.same_bits(c.arr('dx'), c.arr('g'), dn)
    elif op == 'scalar_clamp_min_ste':
        y = oracle.scalar_clamp(x, dt, c['lo'], None)
        assert G.same_bits(y, c.arr('y'), dn)
        assert G.same_bits(c.arr('dx'), c.arr('g'), dn)
    elif op in ('tensor_clamp_ste', 'tensor_clamp', 'tensor_clamp_ste_'):
        lo, hi = c.arr('lo').reshape(-1), c.arr('hi').reshape(-1)
        y = oracle.tensor_clamp(x, lo, hi, dt)
        assert G.same_bits(y, c.arr('y'), dn), G.mismatch_report(y, c.arr('y'), dn)
        if op == 'tensor_clamp_ste':
            assert G.same_bits(c.arr('dx'), c.arr('g'), dn)
        elif op == 'tensor_clamp':
            dx = oracle.tensor_clamp_bwd(c.arr('g'), x, lo, hi, dt)
            assert G.same_bits(dx, c.arr('dx'), dn)
    else:
        raise AssertionError(op)


# ---- known-answer vectors the reference itself carries ----------------------------------------------

def _f32(v):
    return np.asarray(v, dtype=np.float32)


def test_doctest_int_quant(oracle):
    """B/core/quant/int_base.py:32-38: IntQuant(narrow_range=True, signed=True), scale .01, 4 bit"""
    d = oracle.make_desc(1, 1, 4, oracle.F32, oracle.F32, oracle.F32, qmin=-7, qmax=7, clamp_ste=False)
    y, codes = oracle.fakequant_fwd(d, _f32([0.042, -0.053, 0.31, -0.44]), _f32([0.01]), _f32([0.0]))
    np.testing.assert_allclose(y, [0.04, -0.05, 0.07, -0.07], rtol=0, atol=5e-5)  # printed to 4 decimals
    assert codes.tolist() == [4, -5, 7, -7]


def test_doctest_rescaling_int_quant(oracle):
    """B/core/quant/int.py:113-134: ConstScaling(0.1), narrow signed 4 bit -> scale 0.1/7"""
    scale = _f32([0.1]) / _f32([7.0])
    assert abs(float(scale[0]) - 0.0143) < 5e-5
    d = oracle.make_desc(1, 1, 4, oracle.F32, oracle.F32, oracle.F32, qmin=-7, qmax=7)
    y, _ = oracle.fakequant_fwd(d, _f32([0.042, -0.053, 0.31, -0.44]), scale, _f32([0.0]))
    np.testing.assert_allclose(y, [0.0429, -0.0571, 0.1000, -0.1000], rtol=0, atol=5e-5)
    c = [k for k in G.load('quant_graphs') if k['graph'] == 'const_scale_doctest'][0]
    assert G.bits_equal(y, c.arr('y')) and G.bits_equal(scale.reshape(()), c.arr('scale'))


def test_doctest_elementwise(oracle):
    """B/function/ops_ste.py:56-63 (round_ste), B/function/ops.py:27-29,47-49,67-69,94-96"""
    O = oracle
    assert O.unary(O.OP_ROUND, _f32([1.7, -1.7]), O.F32).tolist() == [2.0, -2.0]
    assert O.unary(O.OP_BINARY_SIGN, _f32([2.1, -0.3, 0.0]), O.F32).tolist() == [1.0, -1.0, 1.0]
    rtz = O.unary(O.OP_ROUND_TO_ZERO, _f32([-1.5, -0.5, 0.5, 1.5]), O.F32)
    assert rtz.tolist() == [-1.0, -0.0, 0.0, 1.0] and np.signbit(rtz[1]) and not np.signbit(rtz[2])
    dpu = O.unary(O.OP_DPU_ROUND, _f32([-1.5, -0.5, 0.5, 1.5]), O.F32)
    assert dpu.tolist() == [-1.0, -0.0, 0.0, 2.0] and np.signbit(dpu[1])
    tc = O.tensor_clamp(_f32([1.7, -0.5, 0.1]), _f32([0.0]), _f32([1.0]), O.F32)
    np.testing.assert_array_equal(tc, _f32([1.0, 0.0, 0.1]))
    assert O.unary(O.OP_CEIL, _f32([1.7, -1.7]), O.F32).tolist() == [2.0, -1.0]
    assert O.unary(O.OP_FLOOR, _f32([1.7, -1.7]), O.F32).tolist() == [1.0, -2.0]


@pytest.mark.parametrize('signed', [True, False])
@pytest.mark.parametrize('narrow', [True, False])
@pytest.mark.parametrize('bw', range(2, 9))
@pytest.mark.parametrize('scale', [0.001, 5.0])
@pytest.mark.parametrize('zp_mult', [0.0, 0.3, 0.7])
def test_int_quant_arange_round_trip(oracle, signed, narrow, bw, scale, zp_mult):
    """tests/brevitas/core/test_int_quant.py:44-59: every representable grid point round-trips"""
    qmin, qmax = int_range(signed, narrow, bw)
    zp = float(np.float32(zp_mult) * np.float32(2 ** (bw - 1) - 1))
    grid = np.arange(qmin, qmax + 1, dtype=np.float32)
    x = (np.float32(scale) * (grid - np.float32(zp))).astype(np.float32)
    d = oracle.make_desc(1, 1, x.size, oracle.F32, oracle.F32, oracle.F32, qmin=qmin, qmax=qmax)
    y, _ = oracle.fakequant_fwd(d, x, _f32([scale]), _f32([zp]))
    assert np.isclose(x, y, rtol=1e-5, atol=1e-8).all()
