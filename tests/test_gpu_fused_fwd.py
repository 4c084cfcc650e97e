"""bvq_stats_fakequant_fwd (statistic + quantizer in one launch, the channel held in ONE workgroup's
registers: weights, small activations) against the two-call route it replaces (bvq_absmax_scale +
bvq_fakequant_fwd, itself pinned to the reference's golden vectors): statistic, scale and y identical bit
for bit -- every dtype, ReLU pre-op, rounding modes, the lower bound on the scale, NaN / inf / -0.0; larger
channels must report "not covered" and take the two-call route."""

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


def fits_one_workgroup(x, outer, inner):
    """the register-resident form: at most 8 slices of 512 16-byte chunks per channel"""
    cpr = inner // (16 // x.element_size())
    return outer * ((cpr + 511) // 512) <= 8


def bits(t):
    return t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)


def both(nat, x, outer, ch, inner, *, min_val=1e-10, thr=128.0, qmin=-128.0, qmax=127.0, rm=0, pre=0, scale_dtype=None,
         scalar_mode=0):
    code = nat.dtype_code(x.dtype)
    scale_dtype = scale_dtype or x.dtype
    d = nat.QuantDesc(outer, ch, inner, code, code, nat.dtype_code(scale_dtype), nat.F32, int(ch > 1), 0, qmin, qmax,
                      rm, scalar_mode, 0, nat.OUT_DEQUANT, pre)
    flat = x.reshape(-1)
    fused = nat.stats_fakequant_fwd(d, flat, min_val, thr, scale_dtype)
    stat, scale = nat.absmax_scale(flat, outer, ch, inner, min_val, thr, scale_dtype, pre)
    y = nat.fakequant_fwd(d, flat, scale, torch.zeros(1, device=DEV))
    return fused, (stat, scale, y)


SHAPES = [  # (outer, channels, inner) and what it exercises
    (8, 16, 196),      # activation [8,16,14,14]: one slice per row (f32 only: 196 % 8 != 0)
    (8, 16, 784),      # activation [8,16,28,28]
    (64, 4, 3136),     # 64 rows per channel: more than one workgroup holds -> two-call route
    (1, 64, 4608),     # conv weight [64,512,3,3]: several slices per row, outer = 1
    (1, 1, 4096),      # per-tensor, one wave
    (2, 5, 8192 + 64),  # rows longer than a slice, with a short last slice
    (5, 7, 1000),      # odd sizes (1000 = 125 chunks of 8 for 16-bit types)
]


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_fused_equals_two_calls(dn, shape):
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3).to(DT[dn])
    x[0, 0, 0] = -0.0
    if ch > 1:
        x[:, 1, :] = 0.0  # an all-zero channel: the lower bound on the scale decides
    for kw in (dict(), dict(pre=1), dict(rm=1), dict(rm=4, qmin=0.0, qmax=255.0), dict(min_val=None), dict(thr=7.0, qmin=-7.0, qmax=7.0)):
        fused, ref = both(nat, x, outer, ch, inner, **kw)
        if inner % (16 // x.element_size()) or not fits_one_workgroup(x, outer, inner):
            assert fused is None  # ragged rows and channels larger than a workgroup's registers: two-call route
            continue
        assert fused is not None, kw
        for a, b, name in zip(fused, ref, ('stat', 'scale', 'y')):
            assert torch.equal(bits(a), bits(b)), (name, kw)


@pytest.mark.parametrize('dn', ['f32', 'bf16'])
def test_fused_special_values_and_scalar_scale(dn):
    from brevitas_amd import _native as nat
    torch.manual_seed(1)
    x = torch.randn(4, 6, 128, device=DEV).to(DT[dn])
    x[0, 2, 5] = float('nan')   # the channel's statistic, scale and outputs become NaN, as in the reference
    x[1, 3, 7] = float('inf')
    x[2, 4, 9] = float('-inf')
    fused, ref = both(nat, x, 4, 6, 128)
    for a, b in zip(fused, ref):
        assert torch.equal(bits(a), bits(b))
    # per-tensor: 0-dim float32 scale next to a 16-bit tensor, both scalar semantics
    for mode in (0, 1):
        fused, ref = both(nat, x[:, :2].contiguous(), 1, 1, 4 * 2 * 128, scale_dtype=torch.float32, scalar_mode=mode)
        assert fused is not None
        for a, b in zip(fused, ref):
            assert torch.equal(bits(a), bits(b)), mode


def test_shapes_outside_the_fused_form_fall_back():
    from brevitas_amd import _native as nat
    x = torch.randn(3, 5, 49, device=DEV).to(torch.bfloat16)      # 49 elements per row: not a multiple of 8
    fused, _ = both(nat, x, 3, 5, 49)
    assert fused is None
    big = torch.randn(1 << 26, device=DEV).to(torch.bfloat16)       # one channel of 128 MiB: per-tensor, no pipeline
    fused, _ = both(nat, big, 1, 1, big.numel())
    assert fused is None


def test_module_forward_uses_it_and_backward_is_unchanged():
    """RescalingIntQuant on the stats-scaled graphs: same outputs and gradients with the one-kernel forward
    switched off (BVQ-level kill switch is per process, so compare against the op-by-op configuration)"""
    import brevitas_amd.config as config
    import brevitas_amd.quant as Q
    from bench import build_quantizer
    torch.manual_seed(123456)
    x = torch.randn(8, 16, 14, 14, device=DEV, dtype=torch.bfloat16)
    g = torch.randn_like(x)
    outs = []
    for fused in (True, False):
        q = build_quantizer(16, True, torch.device(DEV))
        xi = x.clone().requires_grad_(True)
        old = config.FUSED_PATHS
        config.FUSED_PATHS = fused
        try:
            y, scale, _, _ = q(xi)
            y.backward(g)
        finally:
            config.FUSED_PATHS = old
        outs.append((y.detach(), scale.detach(), xi.grad, q.scaling_impl.runtime_stats.running_stats.clone()))
    (y0, s0, dx0, r0), (y1, s1, dx1, r1) = outs
    assert torch.equal(bits(y0), bits(y1)) and torch.equal(bits(s0), bits(s1)) and torch.equal(r0, r1)
    diff = (bits(dx0) != bits(dx1)).sum()
    assert int(diff) <= 16  # only the arg-max deposits may differ (reduced gradient)
    w = torch.nn.Parameter(torch.randn(32, 16, 3, 3, device=DEV) * 0.1)
    qw = Q.Int8WeightPerChannelFloat(w).to(DEV)
    yw = qw(w)[0]
    yw.sum().backward()
    assert w.grad is not None and bool(torch.isfinite(yw).all())


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', [(8, 16, 196), (64, 4, 3136), (1, 64, 4608), (37, 7, 1000), (5, 3, 49)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_two_launch_backward_equals_the_four_launch_route(dn, shape):
    """bvq_fakequant_bwd_stats (backward kernel + one finishing kernel) against bvq_fakequant_bwd(tie_stat) +
    bvq_stat_tie_apply_dscale (tie init, backward kernel, channel sums, deposit): dx and dscale bit for bit"""
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    dt = DT[dn]
    code = nat.dtype_code(dt)
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 2).to(dt)
    x[0, 1, 3] = 9.0
    if outer > 1:
        x[1, 1, 2] = -9.0  # a +-max tie inside a channel: the first one gets the deposit
    x[:, 2, :] = 0.0      # an all-zero channel: every element ties, sgn(0) = 0
    g = torch.randn(outer, ch, inner, device=DEV).to(dt)
    flat, gf = x.reshape(-1), g.reshape(-1)
    zp = torch.zeros(1, device=DEV)
    for pre, clamp_ste in ((0, 0), (1, 0), (0, 1)):
        stat, scale = nat.absmax_scale(flat, outer, ch, inner, 1e-10, 128.0, dt, pre)
        d = nat.QuantDesc(outer, ch, inner, code, code, code, nat.F32, 1, 0, -128.0, 127.0, 0, 0, clamp_ste,
                          nat.OUT_DEQUANT, pre)
        res = nat.fakequant_bwd_stats(d, gf, flat, scale, zp, stat, dt, 128.0, dt, want_dscale=True)
        assert res is not None
        dx_a, ds_a = res
        dx_b, ds_b, _, ties = nat.fakequant_bwd(d, gf, flat, scale, zp, True, False, tie_stat=stat)
        nat.stat_tie_apply_dscale(flat, stat, ds_b, dt, 128.0, dt, ties, dx_b, outer, ch, inner, pre_op=pre)
        assert torch.equal(bits(ds_a), bits(ds_b)), (pre, clamp_ste)  # as bits: a zero scale (f16) gives NaN sums
        assert torch.equal(bits(dx_a), bits(dx_b)), (pre, clamp_ste)
    # per-tensor scale: not covered (its gradient is shared by all ties)
    dpt = nat.QuantDesc(1, 1, flat.numel(), code, code, code, nat.F32, 0, 0, -128.0, 127.0, 0, 0, 0, nat.OUT_DEQUANT, 0)
    assert nat.fakequant_bwd_stats(dpt, gf, flat, scale[:1].contiguous(), zp, stat[:1].contiguous(), dt, 128.0, dt) is None
