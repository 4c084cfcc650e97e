"""Parity at BASELINE.json's full sizes.

The oracle would need minutes on 4.1e8 elements, so the [256,512,56,56] activation is checked through
size-independent properties of the domain (quantize/dequantize identity against the emitted integer
codes, idempotence, clamp mask, linearity of the backward in g, shard independence, run-to-run
determinism) plus an independent elementwise restatement with torch ops on the device; the weight
configs (2.4e6 and 6.7e7 elements) are compared with the oracle directly, bit for bit.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def nat():
    from brevitas_amd import _native
    return _native


@pytest.fixture(scope='module')
def act():
    """config 3 / the metric's tensor: [256,512,56,56] bf16, seed 123456"""
    torch.manual_seed(123456)
    x = torch.randn(256, 512, 56, 56, device=DEV, dtype=torch.bfloat16)
    g = torch.randn(256, 512, 56, 56, device=DEV, dtype=torch.bfloat16)
    return x, g


def bits(t):
    return t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)


def test_full_size_per_channel_properties(nat, act):
    x, g = act
    N, C, H, W = x.shape
    inner = H * W
    flat = x.reshape(-1)
    # statistic: one streaming read, equals an independent torch reduction
    stat = nat.stats(nat.STAT_ABSMAX, flat, N, C, inner)
    assert torch.equal(stat, x.abs().amax(dim=(0, 2, 3)))
    scale = (stat / torch.tensor(128.0, device=DEV)).clamp_min(1e-10)  # bf16 [C]
    zp = torch.zeros(1, device=DEV)
    d = nat.QuantDesc(N, C, inner, nat.BF16, nat.BF16, nat.BF16, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    y, codes = nat.fakequant_fwd(d, flat, scale, zp, want_codes=True)
    # codes in range, and the absolute maximum of every channel hits +-128 -> clipped to 127 on the + side
    assert int(codes.min()) >= -128 and int(codes.max()) <= 127
    # dequantized value == code * scale, rounded once to bf16 (elementwise identity)
    s_full = scale.float().view(1, C, 1).expand(N, C, inner).reshape(-1)
    # (numerically: an int32 code cannot carry the sign of a -0.0 result; the bitwise check follows)
    assert torch.equal(y, (codes.float() * s_full).to(torch.bfloat16))
    # independent restatement of the op chain with torch's own bf16 kernels (same-dtype operands)
    sb = scale.view(1, C, 1, 1)
    t = torch.round(x / sb + 0.0)
    t = torch.where(t > 127.0, torch.full_like(t, 127.0), t)
    t = torch.where(t < -128.0, torch.full_like(t, -128.0), t)
    assert torch.equal(bits(y), bits((t * sb).reshape(-1)))
    assert torch.equal(codes, t.reshape(-1).to(torch.int32))
    # idempotence: a dequantized tensor is a fixed point of its own quantizer (as values: -0.0, which
    # small negative inputs produce, re-quantizes to +0.0 through the "+ zero_point" of the chain)
    y2 = nat.fakequant_fwd(d, y, scale, zp)
    assert torch.equal(y2, y)
    # shard independence: quantizing half the batch with the same scale gives that half of y
    dh = nat.QuantDesc(N // 2, C, inner, nat.BF16, nat.BF16, nat.BF16, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    half = flat.numel() // 2
    assert torch.equal(bits(nat.fakequant_fwd(dh, flat[half:], scale, zp)), bits(y[half:]))
    del y2, t

    # backward
    gf = g.reshape(-1)
    dx, ds, _, ties = nat.fakequant_bwd(d, gf, flat, scale, zp, True, False, tie_stat=stat)
    t3 = torch.round(x / sb + 0.0).reshape(-1)
    clipped = (t3 > 127.0) | (t3 < -128.0)
    assert bool((dx[clipped] == 0).all())
    want_dx = torch.where(clipped.view(N, C, H, W), torch.zeros_like(g), (g * sb) / sb).reshape(-1)
    assert torch.equal(bits(dx), bits(want_dx))
    # scale-gradient sums against a float64 reduction of the same per-element terms
    t1 = (x / sb)
    q = torch.clamp(torch.round(t1 + 0.0), -128.0, 127.0)
    dt = torch.where(clipped.view(N, C, H, W), torch.zeros_like(g), g * sb)
    terms = (g * q).double() + (-dt * (t1 / sb)).double()
    ref = terms.sum(dim=(0, 2, 3))
    mag = ((g * q).double().abs() + (dt * (t1 / sb)).double().abs()).sum(dim=(0, 2, 3))
    assert bool(((ds.double() - ref).abs() <= 1e-6 * mag + 1e-6).all())  # float32 partials, double combine
    # tie bookkeeping: the recorded first position of every channel attains the statistic
    pos = ties[:C]
    assert bool((pos >= 0).all())
    xs = x.permute(1, 0, 2, 3).reshape(C, -1)
    assert torch.equal(xs.gather(1, pos.view(C, 1)).abs().view(-1), stat)
    first = (xs.abs() == stat.view(C, 1)).float().argmax(dim=1)
    assert torch.equal(first, pos)
    # linearity in g (powers of two are exact): dx(2g) == 2 dx(g), dscale(2g) == 2 dscale(g)
    dx2, ds2, _ = nat.fakequant_bwd(d, gf * 2, flat, scale, zp, True, False)
    assert torch.equal(bits(dx2), bits(dx * 2)) and torch.equal(ds2, ds * 2)
    # determinism
    dx3, ds3, _ = nat.fakequant_bwd(d, gf, flat, scale, zp, True, False)
    assert torch.equal(bits(dx3), bits(dx)) and torch.equal(ds3, ds)


def test_full_size_per_tensor_properties(nat, act):
    """one scale for the whole [256,512,56,56] activation (Int8ActPerTensorFloat's layout): a single
    channel of 4.1e8 elements, ~1e5 work units, partials combined by the two-stage finish kernels"""
    x, g = act
    n = x.numel()
    flat, gf = x.reshape(-1), g.reshape(-1)
    stat = nat.stats(nat.STAT_ABSMAX, flat, 1, 1, n)
    assert torch.equal(stat.reshape(()), x.abs().amax())
    st2, scale = nat.absmax_scale(flat, 1, 1, n, 1e-10, 128.0, torch.bfloat16)
    assert torch.equal(st2, stat)
    assert torch.equal(scale, (stat / torch.tensor(128.0, device=DEV)).clamp_min(1e-10))
    mm = nat.stats(nat.STAT_MINMAX, flat, 1, 1, n)
    assert torch.equal(mm.reshape(-1), torch.stack([x.amax(), x.amin()]))
    zp = torch.zeros(1, device=DEV)
    d = nat.QuantDesc(1, 1, n, nat.BF16, nat.BF16, nat.BF16, nat.F32, 0, 0, -128.0, 127.0, 0, 0, 0, 0)
    y, codes = nat.fakequant_fwd(d, flat, scale, zp, want_codes=True)
    t = torch.round(flat / scale + 0.0)
    clipped = (t > 127.0) | (t < -128.0)
    q = torch.clamp(t, -128.0, 127.0)
    assert torch.equal(bits(y), bits(q * scale))
    assert torch.equal(codes, q.to(torch.int32))
    del codes, y
    dx, ds, _, ties = nat.fakequant_bwd(d, gf, flat, scale, zp, True, False, tie_stat=stat)
    assert torch.equal(bits(dx), bits(torch.where(clipped, torch.zeros_like(gf), (gf * scale) / scale)))
    t1 = flat / scale
    dt = torch.where(clipped, torch.zeros_like(gf), gf * scale)
    a, b = (gf * q).double(), (dt * (t1 / scale)).double()
    ref, mag = (a - b).sum(), (a.abs() + b.abs()).sum()
    assert ds.numel() == 1
    assert abs(float(ds.double()) - float(ref)) <= 1e-6 * float(mag) + 1e-6
    del a, b, t1, dt, t, q
    # full reduce: every position that attains the maximum is recorded (a handful at most here)
    total = int(ties[0])
    want_pos = torch.nonzero(flat.abs() == stat.reshape(())).reshape(-1)
    assert total == want_pos.numel()
    assert torch.equal(torch.sort(ties[2:2 + total]).values, want_pos)
    # determinism of the split reduction
    dx3, ds3, _ = nat.fakequant_bwd(d, gf, flat, scale, zp, True, False)
    assert torch.equal(bits(dx3), bits(dx)) and torch.equal(ds3, ds)
    # a zero-point gradient next to the scale gradient (both partial arrays through both stages)
    dzp_d = nat.QuantDesc(1, 1, n, nat.BF16, nat.BF16, nat.BF16, nat.F32, 0, 0, 0.0, 255.0, 0, 0, 0, 0)
    zp1 = torch.full((1,), 128.0, device=DEV)
    dx4, ds4, dz4 = nat.fakequant_bwd(dzp_d, gf, flat, scale, zp1, True, True)
    tz = torch.round(flat / scale + 128.0)
    cz = (tz > 255.0) | (tz < 0.0)
    dtz = torch.where(cz, torch.zeros_like(gf), gf * scale)
    refz = dtz.double().sum() - (gf * scale).double().sum()
    magz = dtz.double().abs().sum() + (gf * scale).double().abs().sum()
    assert abs(float(dz4.double()) - float(refz)) <= 1e-6 * float(magz) + 1e-6


def test_full_size_module_fused_equals_op_by_op(act):
    """config 3 through the module surface: the fused graph equals the reference's own op sequence
    run on the device (HIP-backed STE ops + torch's div/add/sub/mul)"""
    import brevitas_amd.config as config
    from bench import build_quantizer
    x, g = act
    outs = {}
    for fused in (True, False):
        config.FUSED_PATHS = fused
        try:
            for per_channel in (True, False):
                q = build_quantizer(x.shape[1], per_channel, torch.device(DEV))
                xi = x.detach().clone().requires_grad_(True)
                y, scale, zp, bw = q(xi)
                y.backward(g)
                outs[(fused, per_channel)] = (y.detach(), scale.detach(), xi.grad,
                                              q.scaling_impl.runtime_stats.running_stats.clone())
                del xi, y
        finally:
            config.FUSED_PATHS = True
    for per_channel in (True, False):
        yf, sf, dxf, rf = outs[(True, per_channel)]
        yg, sg, dxg, rg = outs[(False, per_channel)]
        assert torch.equal(bits(yf), bits(yg)) and torch.equal(sf, sg) and torch.equal(rf, rg)
        # dx: identical except at the (<= a few per channel) arg-max deposits.  The deposit is
        # (sum g*q - sum dt*(x/s)/s) / 128: the op-by-op route rounds each of the two ~4e4-sized sums to
        # bf16 before they cancel, the fused kernel keeps them in float32 -- so the bound is a few bf16
        # ulps of the LARGER partial sum, not of the (much smaller) difference.
        diff = (bits(dxf) != bits(dxg)).reshape(-1).nonzero().reshape(-1)
        assert diff.numel() <= (x.shape[1] if per_channel else 64), diff.numel()
        if diff.numel():
            N, C, H, W = x.shape
            sb = sf.reshape(1, -1, 1, 1).to(x.dtype) if per_channel else sf.to(x.dtype)
            q = torch.clamp(torch.round(x / sb), -128.0, 127.0)
            dims = (0, 2, 3) if per_channel else (0, 1, 2, 3)
            s1 = (g * q).double().sum(dim=dims).abs().reshape(-1)
            s2 = ((g * sb) * ((x / sb) / sb)).double().sum(dim=dims).abs().reshape(-1)
            bound = 2.0 ** -6 * (s1 + s2) / 128.0 + 1e-3
            ch = ((diff // (H * W)) % C) if per_channel else torch.zeros_like(diff)
            a, b = dxf.reshape(-1)[diff].double(), dxg.reshape(-1)[diff].double()
            assert bool(((a - b).abs() <= bound[ch]).all()), ((a - b).abs().max(), bound.max())


@pytest.mark.parametrize('shape,dtype,bit_width', [((512, 512, 3, 3), torch.float32, 8),       # config 2
                                                   ((8192, 8192), torch.bfloat16, 4)],         # config 5 weight
                         ids=['conv512x512x3x3_f32_int8', 'linear8192x8192_bf16_int4'])
def test_weight_configs_vs_oracle(nat, oracle, shape, dtype, bit_width):
    torch.manual_seed(123456)
    w = (torch.randn(shape) * 0.02).to(dtype)
    g = torch.randn(shape).to(dtype)
    cout = shape[0]
    k = w.numel() // cout
    qmax = float(2 ** (bit_width - 1) - 1)
    code = {torch.float32: oracle.F32, torch.bfloat16: oracle.BF16}[dtype]
    od = oracle.make_desc(1, cout, k, code, code, code, oracle.F32, scale_per_channel=True, qmin=-qmax, qmax=qmax,
                          clamp_ste=True)
    wn, _ = oracle.from_torch(w.reshape(-1))
    gn, _ = oracle.from_torch(g.reshape(-1))
    y_o, dx_o, scale_o, stat_o, ds_o = oracle.step_stats_scaled(od, wn, gn, 1e-10, qmax)
    # through the module surface (Int8/Int4WeightPerChannelFloat resolved graph)
    from test_gpu_modules import to_np, weight_quant
    wp = torch.nn.Parameter(w.to(DEV))
    q = weight_quant(wp, bit_width).to(DEV)
    y, scale, zp, bw = q(wp)
    y.backward(g.to(DEV))
    assert np.array_equal(to_np(y).reshape(-1), y_o)
    assert np.array_equal(to_np(scale).reshape(-1), scale_o)
    got, want = to_np(wp.grad).reshape(-1), dx_o
    diff = np.nonzero(got != want)[0]
    assert diff.size <= cout  # only the per-channel arg-max deposits (oracle step has no deposit)


def test_full_size_forward_takes_the_two_kernel_route(nat, act):
    """the one-launch forward covers channels that fit one workgroup's registers; the [256,512,56,56] activation
    (1.6 MB per channel) must report "not covered" so that the module takes statistic kernel + quantizer kernel"""
    x, _ = act
    N, C, H, W = x.shape
    d = nat.QuantDesc(N, C, H * W, nat.BF16, nat.BF16, nat.BF16, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    assert nat.stats_fakequant_fwd(d, x.reshape(-1), 1e-10, 128.0, torch.bfloat16) is None
