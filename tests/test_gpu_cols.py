"""Column-mapped kernels (channel axis last or nearly last: NHWC, [tokens, hidden], 7x7 maps) against the
row-mapped path they replace on those layouts (BVQ_COLS=0 selects it; it is the path pinned to the
reference's golden vectors) and against plain torch reductions: bit-identical statistics, outputs and
gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}

SHAPES = [  # (outer, channels, inner)
    (300, 512, 1),    # [tokens, hidden] / NHWC: one full strip per row
    (257, 64, 1),     # short rows: several rows per wave pass
    (130, 24, 1),     # 3 chunks per row (bf16): 21 rows per pass, idle lanes
    (65, 1000, 1),    # two strips, the second one partly empty
    (33, 16, 49),     # 7x7 maps
    (40, 6, 12),      # small inner, odd sizes
    (5000, 8, 2),     # many rows: several row blocks
]


def bits(t):
    return t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_cols_absmax(dn, shape):
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3).to(DT[dn])
    x[0, 1, 0] = float('nan') if dn == 'f32' else 7.0
    for pre in (0, 1):
        stat = nat.stats(nat.STAT_ABSMAX, x.reshape(-1), outer, ch, inner, pre_op=pre)
        src = torch.relu(x) if pre else x
        want = src.abs().amax(dim=(0, 2))
        assert torch.equal(bits(stat), bits(want)), pre
        st2, scale = nat.absmax_scale(x.reshape(-1), outer, ch, inner, 1e-10, 128.0, DT[dn], pre)
        assert torch.equal(bits(st2), bits(want))


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_cols_minmax(dn, shape):
    """AbsMinMax / NegativeMinOrZero statistics (asymmetric per-channel quantizers) on the same layouts: torch's
    amax / amin, the row-mapped route on a misaligned copy, NaN poisoning one channel only"""
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(123457)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3 + 0.5).to(DT[dn])
    x[:, 0] = x[:, 0].abs() + 1  # an all-positive and an all-negative channel
    x[:, 2] = -x[:, 2].abs() - 1
    for pre in (0, 1):
        src = torch.relu(x) if pre else x
        got = nat.stats(nat.STAT_MINMAX, x.reshape(-1), outer, ch, inner, pre_op=pre).view(2, ch)
        assert torch.equal(got[0], src.amax(dim=(0, 2))) and torch.equal(got[1], src.amin(dim=(0, 2))), pre
        row = nat.stats(nat.STAT_MINMAX, _misaligned(x), outer, ch, inner, pre_op=pre).view(2, ch)
        assert torch.equal(bits(got), bits(row)), pre
    x[outer // 2, 1, inner - 1] = float('nan')
    got = nat.stats(nat.STAT_MINMAX, x.reshape(-1), outer, ch, inner).view(2, ch)
    assert torch.isnan(got[:, 1]).all() and not torch.isnan(got[:, 0]).any() and not torch.isnan(got[:, 2:]).any()


def _misaligned(t):
    """the same values in a buffer that starts one element off a 16-byte boundary: the library then takes its
    row-mapped route (the column-mapped one needs aligned rows), which is the reference here"""
    buf = torch.empty(t.numel() + 16, dtype=t.dtype, device=t.device)
    view = buf[1:1 + t.numel()]
    view.copy_(t.reshape(-1))
    assert view.data_ptr() % 16 != 0
    return view


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_cols_forward_and_backward_equal_row_mapped(dn, shape):
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    dt = DT[dn]
    code = nat.dtype_code(dt)
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 2).to(dt).reshape(-1)
    g = torch.randn(outer, ch, inner, device=DEV).to(dt).reshape(-1)
    xm, gm = _misaligned(x), _misaligned(g)
    for pre, clamp_ste, rm, zpv in ((0, 0, 0, 0.0), (1, 0, 0, 0.0), (0, 1, 1, 0.0), (0, 0, 0, 3.0)):
        stat, scale = nat.absmax_scale(x, outer, ch, inner, 1e-10, 128.0, dt, pre)
        zp = torch.full((1,), zpv, device=DEV)
        d = nat.QuantDesc(outer, ch, inner, code, code, code, nat.F32, 1, 0, -128.0, 127.0, rm, 0, clamp_ste,
                          nat.OUT_DEQUANT, pre)
        y_c = nat.fakequant_fwd(d, x, scale, zp)
        y_r = nat.fakequant_fwd(d, xm, scale, zp)
        assert torch.equal(bits(y_c), bits(y_r)), ('y', pre, clamp_ste, rm, zpv)
        # backward through the general entry point (atomics for the arg-max positions)
        dx_c, ds_c, _, ties_c = nat.fakequant_bwd(d, g, x, scale, zp, True, False, tie_stat=stat)
        dx_r, ds_r, _, ties_r = nat.fakequant_bwd(d, gm, xm, scale, zp, True, False, tie_stat=stat)
        assert torch.equal(bits(dx_c), bits(dx_r)), ('dx', pre, clamp_ste, rm, zpv)
        assert torch.equal(ties_c[:ch], ties_r[:ch])
        ok = torch.isfinite(ds_r)
        assert torch.allclose(ds_c[ok], ds_r[ok], rtol=2e-5, atol=1e-4 * float(ds_r[ok].abs().max() + 1))
        # dx only
        dx_c2, _, _ = nat.fakequant_bwd(d, g, x, scale, zp, False, False)
        assert torch.equal(bits(dx_c2), bits(dx_r))
        # two-launch stats backward: deposit at the same element
        if zpv == 0.0:
            res = nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, dt, 128.0, dt, want_dscale=True)
            assert res is not None
            dxs, dss = res
            want = dx_r.clone()
            nat.stat_tie_apply_dscale(xm, stat, ds_r, dt, 128.0, dt, ties_r, want, outer, ch, inner, pre_op=pre)
            diff = (bits(dxs) != bits(want)).nonzero().reshape(-1)
            assert diff.numel() <= ch  # only the deposit elements may differ (reduced gradient, other order)
            if diff.numel():
                a, b = dxs[diff].float(), want[diff].float()
                # the deposit is dscale / 128: dscale sums differ by their (float32 / double) summation order
                okd = torch.isfinite(ds_r)
                deposit = float(ds_r[okd].abs().max()) / 128.0 if bool(okd.any()) else 0.0
                tol = {'f32': 1e-4, 'bf16': 2.0 ** -6, 'f16': 2.0 ** -8}[dn]
                assert bool(((a - b).abs() <= tol * (a.abs() + b.abs() + deposit + 1e-3)).all())


def test_cols_mixed_scales_take_the_exact_division_per_wave():
    """a float32-valued (non-bf16) scale in one channel: waves touching it divide exactly, the others use the
    reciprocal; both equal the row-mapped result"""
    from brevitas_amd import _native as nat
    torch.manual_seed(3)
    outer, ch = 64, 512
    x = torch.randn(outer, ch, device=DEV).to(torch.bfloat16).reshape(-1)
    scale = (torch.rand(ch, device=DEV) * 0.05 + 0.01)  # float32 scales: not bf16 values
    zp = torch.zeros(1, device=DEV)
    d = nat.QuantDesc(outer, ch, 1, nat.BF16, nat.BF16, nat.F32, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0, 0)
    # with a float32 scale tensor the compute type is float32 in the reference; here: same dtype contract -> skip
    d2 = nat.QuantDesc(outer, ch, 1, nat.BF16, nat.BF16, nat.BF16, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0, 0)
    sb = scale.to(torch.bfloat16)
    sb[5] = torch.tensor(3e-6).to(torch.bfloat16)  # below 2^-14: the reciprocal path does not cover it
    y_c = nat.fakequant_fwd(d2, x, sb, zp)
    y_r = nat.fakequant_fwd(d2, _misaligned(x), sb, zp)
    assert torch.equal(bits(y_c), bits(y_r))


@pytest.mark.parametrize('kind', ['stats_scaled', 'learned_scale'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_channels_last_per_channel_quantizer(kind, dtype):
    """a per-channel (dim 1) quantizer on a dense channels_last activation runs on its memory as it lies
    ([N*H*W, C], column-mapped kernels) and returns channels_last tensors: same values as for the NCHW tensor"""
    import brevitas_amd.quant as Q
    from brevitas_amd.core.bit_width import BitWidthConst
    from brevitas_amd.core.function_wrapper import RoundSte, TensorClamp
    from brevitas_amd.core.quant import IntQuant, RescalingIntQuant
    from brevitas_amd.core.restrict_val import FloatRestrictValue
    from brevitas_amd.core.scaling import IntScaling, ParameterScaling
    from brevitas_amd.core.zero_point import ZeroZeroPoint
    torch.manual_seed(123456)
    N, C, H, W = 6, 16, 5, 7
    x = torch.randn(N, C, H, W, device=DEV).to(dtype)
    g = torch.randn(N, C, H, W, device=DEV).to(dtype)
    outs = []
    for fmt in (torch.contiguous_format, torch.channels_last):
        if kind == 'stats_scaled':
            q = Q.Int8ActPerChannelFloat(C).to(DEV)
        else:
            q = RescalingIntQuant(
                IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
                ParameterScaling(torch.rand(1, C, 1, 1) + 0.5, (1, C, 1, 1), FloatRestrictValue(), 1e-10),
                IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(8)).to(DEV).to(dtype)
            torch.manual_seed(5)
            with torch.no_grad():
                q.scaling_impl.value.copy_((torch.rand(1, C, 1, 1, device=DEV) + 0.5).to(dtype))
        xi = x.clone().to(memory_format=fmt).requires_grad_(True)
        y, scale, _, _ = q(xi)
        y.backward(g.to(memory_format=fmt))
        if fmt is torch.channels_last:
            assert y.is_contiguous(memory_format=torch.channels_last)
            assert xi.grad.is_contiguous(memory_format=torch.channels_last)
        outs.append((y.detach(), scale.detach(), xi.grad, None if kind == 'stats_scaled' else q.scaling_impl.value.grad))
    (y0, s0, dx0, dv0), (y1, s1, dx1, dv1) = outs
    assert torch.equal(bits(y0), bits(y1)) and torch.equal(bits(s0), bits(s1))
    diff = (bits(dx0) != bits(dx1)).reshape(-1).nonzero().reshape(-1)
    assert diff.numel() <= (C if kind == 'stats_scaled' else 0)
    if dv0 is not None:
        assert torch.allclose(dv0.float(), dv1.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=1e-3)
