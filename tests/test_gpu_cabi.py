"""GPU parity of the C-ABI library (libbvq.so) against the reference's golden vectors and the oracle.

Every call goes through brevitas_amd._native, i.e. through the extern "C" entry points of
include/bvq.h, on cuda:0.  Bar: bit-exact for codes, dequantized values and dx; tolerance (stated
below) for the reduced scale / zero-point gradients only.
"""
import numpy as np
import pytest
import torch

import golden_util as G
from test_oracle_golden import RM, desc_for, grad_sum_tolerance, int_range, layout

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def nat():
    from brevitas_amd import _native
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return _native


def to_np(t):
    """device tensor -> oracle-style numpy array (16-bit floats as uint16 patterns)"""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def ndesc(nat, od):
    """oracle descriptor -> native descriptor (same field layout)"""
    d = nat.QuantDesc()
    for f, _ in nat.QuantDesc._fields_:
        setattr(d, f, getattr(od, f))
    return d


INT_QUANT = G.load('int_quant')


@pytest.mark.parametrize('c', INT_QUANT, ids=G.ids(INT_QUANT, ['x_dtype', 'layout', 'round', 'clamp', 'bit_width']))
def test_int_quant_golden(nat, oracle, c):
    d = ndesc(nat, desc_for(oracle, c))
    x = c.torch('x', DEV).reshape(-1)
    scale = c.torch('scale', DEV).reshape(-1)
    zp = c.torch('zp', DEV).reshape(-1)
    g = c.torch('g', DEV).reshape(-1)
    y, codes = nat.fakequant_fwd(d, x, scale, zp, want_codes=True)
    assert G.same_bits(to_np(y), c.arr('y').reshape(-1), c['dtypes']['y']), \
        G.mismatch_report(to_np(y), c.arr('y'), 0)
    d.out_kind = nat.OUT_INT
    yi = nat.fakequant_fwd(d, x, scale, zp)
    assert G.same_bits(to_np(yi), c.arr('y_int').reshape(-1), c['dtypes']['y_int'])
    want = c.f32('y_int').reshape(-1)
    fin = np.isfinite(want)
    assert np.array_equal(to_np(codes)[fin], want[fin].astype(np.int32))
    d.out_kind = nat.OUT_DEQUANT
    dx, ds, dz = nat.fakequant_bwd(d, g, x, scale, zp, True, True)
    assert G.same_bits(to_np(dx), c.arr('dx').reshape(-1), c['dtypes']['dx']), \
        G.mismatch_report(to_np(dx), c.arr('dx'), 0)
    # dx must not depend on whether the sums are requested
    dx2, _, _ = nat.fakequant_bwd(d, g, x, scale, zp, False, False)
    assert torch.equal(dx.view(torch.int16 if dx.element_size() == 2 else torch.int32),
                       dx2.view(torch.int16 if dx.element_size() == 2 else torch.int32))
    for got, name in ((ds, 'dscale'), (dz, 'dzp')):
        want = c.f32(name).reshape(-1).astype(np.float64)
        got = to_np(got).astype(np.float64)
        if got.size != want.size:
            got = np.array([got.sum()])
        tol = grad_sum_tolerance(c, name)
        with np.errstate(all='ignore'):
            ok = (np.isnan(got) & np.isnan(want)) | (np.isinf(got) & np.isinf(want)) | ~np.isfinite(tol) | \
                (np.abs(got - want) <= tol)
        assert ok.all(), (name, got, want, tol)


STATS = G.load('stats')


@pytest.mark.parametrize('c', STATS, ids=G.ids(STATS, ['stat', 'dtype', 'tag', 'chdim']))
def test_stats_golden(nat, c):
    outer, ch, inner = layout(c['shape'], c['chdim'])
    x = c.torch('x', DEV).reshape(-1)
    if c['stat'] == 'absmax':
        out = nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner)
        assert G.same_bits(to_np(out), c.arr('out').reshape(-1), c['dtype'])
        out32 = nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner, out_f32=True)
        assert G.bits_equal(to_np(out32), c.f32('out').reshape(-1))
        if not np.isnan(c.f32('out')).any():
            gout = c.torch('gout', DEV).reshape(-1)
            dx = nat.stat_bwd(nat.MATCH_ABS, x, out, gout, outer, ch, inner)
            # bit-exact, including the sign of the zeros (0 * sgn(x) in the reference)
            assert G.same_bits(to_np(dx), c.arr('dx').reshape(-1), c['dtype']), \
                G.mismatch_report(to_np(dx), c.arr('dx'), 0)
            # additive mode on top of an existing gradient
            base = torch.randn(x.shape, device=DEV).to(x.dtype)
            acc = nat.stat_bwd(nat.MATCH_ABS, x, out, gout, outer, ch, inner, dx=base.clone())
            assert torch.equal(acc, (base + dx))
    else:
        out = nat.stats(nat.STAT_MINMAX, x, outer, ch, inner)
        got = torch.abs(out[:ch] - out[ch:])
        assert G.same_bits(to_np(got), c.arr('out').reshape(-1), c['dtype'])


STE = [c for c in G.load('ste_ops') if c['op'] != 'int_range']
UNARY = {'round_ste': 0, 'floor_ste': 1, 'ceil_ste': 2, 'round_to_zero_ste': 3, 'dpu_round_ste': 4,
         'binary_sign_ste': 5, 'ternary_sign_ste': 6, 'abs_binary_sign_grad': 7}


@pytest.mark.parametrize('c', STE, ids=G.ids(STE, ['op', 'dtype', 'bounds']))
def test_ste_forward_golden(nat, c):
    op, dn = c['op'], c['dtype']
    x = c.torch('x', DEV)
    if op in UNARY:
        y = nat.unary(UNARY[op], x)
        if op == 'abs_binary_sign_grad':
            dx = nat.abs_binary_sign_grad_bwd(c.torch('g', DEV), x)
            assert G.same_bits(to_np(dx), c.arr('dx'), dn)
    elif op == 'scalar_clamp_ste':
        y = nat.scalar_clamp(x, c['lo'], c['hi'])
    elif op == 'scalar_clamp_min_ste':
        y = nat.scalar_clamp(x, c['lo'], None)
    else:
        lo, hi = c.torch('lo', DEV), c.torch('hi', DEV)
        y = nat.tensor_clamp(x, lo, hi)
        if op == 'tensor_clamp':
            dx = nat.tensor_clamp_bwd(c.torch('g', DEV), x, lo, hi)
            assert G.same_bits(to_np(dx), c.arr('dx'), dn)
    assert G.same_bits(to_np(y), c.arr('y'), dn), G.mismatch_report(to_np(y), c.arr('y'), dn)


# ---- seeded random parity against the oracle at sizes the oracle finishes in seconds ----------------

CASES = [
    # (shape, chdim, x dtype, scale dtype, bit width, signed, narrow, clamp_ste, scalar_mode)
    ((4, 1024), None, torch.float32, torch.float32, 8, True, False, False, 0),          # config 1
    ((64, 64, 3, 3), 0, torch.float32, torch.float32, 8, True, True, True, 0),          # config 2 (scaled down)
    ((8, 32, 14, 14), 1, torch.bfloat16, torch.bfloat16, 8, True, False, False, 0),     # config 3 layout
    ((8, 32, 14, 14), None, torch.bfloat16, torch.float32, 8, True, False, False, 0),   # 0-dim f32 scale, CPU semantics
    ((8, 32, 14, 14), None, torch.bfloat16, torch.float32, 8, True, False, False, 1),   # 0-dim f32 scale, device semantics
    ((8, 32, 14, 14), 1, torch.bfloat16, torch.float32, 8, True, False, False, 0),      # promotes to f32
    ((96, 160), 0, torch.bfloat16, torch.bfloat16, 4, True, True, True, 0),             # config 5 (scaled down)
    ((7, 9, 5), 1, torch.float16, torch.float16, 6, False, False, False, 0),            # odd inner: scalar path
    ((1, 3, 1001), None, torch.float32, torch.float32, 3, False, True, False, 0),       # ragged end
    ((33, 17), 1, torch.bfloat16, torch.bfloat16, 8, True, False, False, 0),            # inner 1
]


@pytest.mark.parametrize('case', CASES, ids=[str(i) for i in range(len(CASES))])
def test_random_vs_oracle(nat, oracle, case):
    shape, chdim, xdt, sdt, bw, signed, narrow, ste, smode = case
    torch.manual_seed(123456)
    x = (torch.randn(shape) * 1.5).to(xdt)
    outer, ch, inner = layout(shape, chdim)
    if chdim is None:
        scale = torch.tensor([0.0123]).to(sdt)
    else:
        scale = (torch.rand(ch) * 0.03 + 0.004).to(sdt)
    zp = torch.zeros(1)
    ct = torch.result_type(x, scale.reshape([1] * 0) if chdim is None else scale)
    if chdim is None and xdt != torch.float32:
        ct = xdt  # a 0-dim operand does not promote a dimensioned tensor
    g = torch.randn(shape).to(ct)
    qmin, qmax = int_range(signed, narrow, bw)
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
    od = oracle.make_desc(outer, ch, inner, code[xdt], code[ct], code[sdt], 0, scale_per_channel=chdim is not None,
                          qmin=qmin, qmax=qmax, clamp_ste=ste, scalar_mode=smode)
    xn, _ = oracle.from_torch(x.reshape(-1))
    sn, _ = oracle.from_torch(scale)
    zn, _ = oracle.from_torch(zp)
    gn, _ = oracle.from_torch(g.reshape(-1))
    y_o, codes_o = oracle.fakequant_fwd(od, xn, sn, zn)
    dx_o, ds_o, dz_o = oracle.fakequant_bwd(od, gn, xn, sn, zn)
    d = ndesc(nat, od)
    xd, sd, zd, gd = x.to(DEV).reshape(-1), scale.to(DEV), zp.to(DEV), g.to(DEV).reshape(-1)
    y, codes = nat.fakequant_fwd(d, xd, sd, zd, want_codes=True)
    dx, ds, dz = nat.fakequant_bwd(d, gd, xd, sd, zd, True, True)
    names = {0: 'f32', 1: 'bf16', 2: 'f16'}
    assert np.array_equal(to_np(codes), codes_o)
    assert G.same_bits(to_np(y), y_o, names[code[ct]])
    assert G.same_bits(to_np(dx), dx_o, names[code[xdt]])
    # reduced gradients: same per-element terms, different summation order / accumulator width.
    # tolerance: 1e-5 of the summed magnitude (float32 accumulation of <= 1e5 terms per channel)
    mag_s = float(np.abs(to_np(g.float())).sum() * (max(abs(qmin), abs(qmax)) * 3))
    np.testing.assert_allclose(to_np(ds), ds_o, rtol=0, atol=1e-5 * mag_s + 1e-12, equal_nan=True)
    np.testing.assert_allclose(to_np(dz), dz_o, rtol=0, atol=1e-5 * mag_s + 1e-12, equal_nan=True)
    # run-to-run reproducible (fixed-order combine, no atomics)
    dx2, ds2, dz2 = nat.fakequant_bwd(d, gd, xd, sd, zd, True, True)
    assert torch.equal(ds.view(torch.int32), ds2.view(torch.int32))
    assert torch.equal(dz.view(torch.int32), dz2.view(torch.int32))
