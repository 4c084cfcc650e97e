"""float16 fast paths (DivF16 and DivF16R, brevitas_amd/csrc/bvq_fakequant.h): the arithmetic claims on every
float16 numerator x every float16 scale in [2^-14, 2^14], and the kernels on inputs full of tiny and subnormal
values (the quotients that must take the IEEE division) against the CPU oracle."""
import numpy as np
import pytest
import torch

from test_gpu_cabi import ndesc, to_np

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_f16_reciprocal_claim_all_scales():
    a = torch.arange(65536, dtype=torch.int32, device=DEV).to(torch.int16).view(torch.float16).float()
    finite = ~torch.isnan(a)
    scales = torch.arange(0x0400, 0x7400 + 1, dtype=torch.int32, device=DEV).to(torch.int16).view(torch.float16).float()
    scales = scales[(scales >= 2.0 ** -14) & (scales <= 2.0 ** 14)]
    assert scales.numel() == 28 * 1024 + 1, scales.numel()
    bad = 0
    for chunk in scales.split(512):
        s = chunk[:, None]
        q = a[None, :] * (1.0 / s)
        fast = q.half().view(torch.int16)
        ref = (a[None, :] / s).half().view(torch.int16)
        ab = q.view(torch.int32) & 0x7fffffff
        guard = (ab > 0) & (ab < 0x38810000)
        bad += int(((fast != ref) & finite[None, :] & ~guard).sum())
    assert bad == 0, bad


def test_f16_refined_division_is_the_ieee_quotient_for_every_pair():
    """DivF16R: product with the correctly rounded reciprocal + one exact remainder step + v_div_fixup == a / s in
    float32, bit for bit, for all 65536 float16 numerators (zeros, subnormals, infinities, NaNs included) x all
    28673 float16 scales in [2^-14, 2^14] -- computed by the library's own device code (bvq_selftest_div_f16r)"""
    from brevitas_amd import _native as nat
    a = torch.arange(65536, dtype=torch.int32, device=DEV).to(torch.int16).view(torch.float16).float().contiguous()
    scales = torch.arange(0x0400, 0x7400 + 1, dtype=torch.int32, device=DEV).to(torch.int16).view(torch.float16).float()
    scales = scales[(scales >= 2.0 ** -14) & (scales <= 2.0 ** 14)].contiguous()
    assert scales.numel() == 28 * 1024 + 1, scales.numel()
    bad, nan_bad = 0, 0
    for chunk in scales.split(1024):
        got = nat.selftest_div_f16r(a, chunk.contiguous())
        ref = a[None, :] / chunk[:, None]
        nan = torch.isnan(ref)
        nan_bad += int((torch.isnan(got) != nan).sum())
        bad += int(((got.view(torch.int32) != ref.view(torch.int32)) & ~nan).sum())
    assert bad == 0 and nan_bad == 0, (bad, nan_bad)
    # the device's own a / s against numpy's on a sample (the reference above must itself be the IEEE quotient)
    sub = scales[::997].cpu().numpy()
    an = a.cpu().numpy()
    with np.errstate(all='ignore'):
        want = an[None, :] / sub[:, None]
    got = nat.selftest_div_f16r(a, scales[::997].contiguous()).cpu().numpy()
    ok = (got.view(np.int32) == want.view(np.int32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all()


@pytest.mark.parametrize('layout', [(8, 16, 3136), (1, 1, 40000), (300, 64, 1), (64, 8, 49)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('zp_kind', ['zero', 'scalar'])
def test_f16_forward_and_backward_with_tiny_values(oracle, layout, zp_kind):
    from brevitas_amd import _native as nat
    O = oracle
    outer, ch, inner = layout
    pc = ch > 1
    g = torch.Generator().manual_seed(123456 + outer + inner)
    n = outer * ch * inner
    x = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g)
    # a third of the elements tiny enough that x / s, (g s) / s or (x / s) / s lands below 2^-14
    tiny = torch.rand(n, generator=g)
    x = torch.where(tiny < 0.2, x * 1e-6, x)
    gr = torch.where((tiny > 0.1) & (tiny < 0.4), gr * 3e-5, gr)
    x, gr = x.half(), gr.half()
    code = nat.dtype_code(torch.float16)
    lay = (outer, ch, inner) if pc else (1, 1, n)
    xd, gd = x.to(DEV), gr.to(DEV)
    stat = nat.stats(nat.STAT_ABSMAX, xd, *lay)
    scale = (stat.float().clamp_min(1e-3) / 127.0).half()
    zp = torch.zeros(1) if zp_kind == 'zero' else torch.tensor([2.0])
    od = O.make_desc(*lay, code, code, code, O.F32, scale_per_channel=pc, zp_per_channel=False, qmin=-127.0,
                     qmax=127.0, round_mode=0, clamp_ste=False, pre_op=0)
    d = ndesc(nat, od)
    xn, _ = O.from_torch(x)
    gn, _ = O.from_torch(gr)
    sn, _ = O.from_torch(scale.cpu())
    y_o, codes_o = O.fakequant_fwd(od, xn, sn, zp.numpy().astype(np.float32))
    y, codes = nat.fakequant_fwd(d, xd, scale, zp.to(DEV), want_codes=True)
    assert np.array_equal(to_np(y), y_o) and np.array_equal(to_np(codes), codes_o)
    assert np.array_equal(to_np(nat.fakequant_fwd(d, xd, scale, zp.to(DEV))), y_o)  # the route without codes
    if zp_kind == 'zero':
        one = nat.stats_fakequant_fwd(d, xd, 1e-3, 127.0, torch.float16)  # statistic + quantizer in one launch
        if one is not None:
            assert torch.equal(one[1], scale) and np.array_equal(to_np(one[2]), y_o)
    dx_o, ds_o, _ = O.fakequant_bwd(od, gn, xn, sn, zp.numpy().astype(np.float32))
    dx, ds, _ = nat.fakequant_bwd(d, gd, xd, scale, zp.to(DEV), True, False)[:3]
    assert np.array_equal(to_np(dx), dx_o)
    np.testing.assert_allclose(ds.cpu().numpy().reshape(-1), np.asarray(ds_o).reshape(-1), rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize('case', [('bf16', 300001), ('f16', 300001), ('f32', 70003), ('f32', 3136 * 6), ('bf16', 8192 * 5)],
                         ids=lambda c: '%s-%d' % c)
def test_long_rows_piece_rules_against_the_oracle(oracle, case):
    """per-tensor rows long enough for every piece rule of the quantizer kernels (7 KiB pieces of 16-bit rows with
    >= 64 pieces, 4 KiB float32 pieces, evenly cut rows of a few pieces, ragged last pieces): forward codes / values
    and dx bit-exact, the scale gradient within summation-order tolerance"""
    from brevitas_amd import _native as nat
    O = oracle
    name, n = case
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[name]
    g = torch.Generator().manual_seed(123456 + n)
    x = (torch.randn(n, generator=g) * 1.5).to(dt)
    gr = torch.randn(n, generator=g).to(dt)
    code = nat.dtype_code(dt)
    scale = torch.tensor([3.0 / 127.0]).to(dt)
    zp = torch.zeros(1)
    od = O.make_desc(1, 1, n, code, code, code, O.F32, scale_per_channel=False, zp_per_channel=False, qmin=-127.0,
                     qmax=127.0, round_mode=0, clamp_ste=False, pre_op=0)
    d = ndesc(nat, od)
    xn, _ = O.from_torch(x)
    gn, _ = O.from_torch(gr)
    sn, _ = O.from_torch(scale)
    y_o, codes_o = O.fakequant_fwd(od, xn, sn, zp.numpy().astype(np.float32))
    xd, gd, sd, zd = x.to(DEV), gr.to(DEV), scale.to(DEV), zp.to(DEV)
    y, codes = nat.fakequant_fwd(d, xd, sd, zd, want_codes=True)
    assert np.array_equal(to_np(y), y_o) and np.array_equal(to_np(codes), codes_o)
    dx_o, ds_o, _ = O.fakequant_bwd(od, gn, xn, sn, zp.numpy().astype(np.float32))
    dx, ds, _ = nat.fakequant_bwd(d, gd, xd, sd, zd, True, False)[:3]
    assert np.array_equal(to_np(dx), dx_o)
    np.testing.assert_allclose(ds.cpu().numpy().reshape(-1), np.asarray(ds_o).reshape(-1), rtol=2e-3, atol=1e-2)
    assert np.array_equal(to_np(nat.fakequant_bwd(d, gd, xd, sd, zd, False, False)[0]), dx_o)  # the dx-only kernel
