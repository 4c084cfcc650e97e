"""The one-launch routes (bvq_absmax_scale_onepass, bvq_fakequant_bwd_stats_onepass: the streaming kernel's
last-arriving wave per channel finishes the channel) against the two-launch routes they replace (themselves pinned to
the reference's golden vectors): statistic, scale, running statistic, dx and the dscale sums identical bit for bit --
every dtype, ragged rows, channel counts that do and do not divide the persistent grid, ReLU pre-op, NaN / inf, the
float32 statistic of the batch-sharded route; the arrival words are zero again after every launch (so one buffer
serves a stream for ever), also after launches on several shapes back to back."""

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


def bits(t):
    return t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)


def arrival_is_clean(nat):
    torch.cuda.synchronize()
    return all(int(b.count_nonzero()) == 0 for b in nat._arrive.values())


SHAPES = [  # (outer, channels, inner)
    (256, 32, 196),     # 14x14 maps (float32 rows stay row-mapped; 16-bit ones take the column-mapped route: not covered)
    (32, 512, 784),     # the persistent grid's period divides: one arrival per wave
    (40, 24, 3136),     # 56x56 rows, 24 channels
    (64, 7, 1000),      # channel count that does not divide the grid: waves change channel and arrive per change
    (3, 5, 8192 + 72),  # rows cut into pieces, a short last piece
    (1, 64, 4608),      # weight-like: outer = 1
    (130, 300, 392),    # more units than persistent waves, odd counts
    (2, 2, 17),         # tiny, ragged (vector width 1)
    (1, 1, 4096 * 100 + 17),  # whole-tensor statistic: workgroup -> shard -> top arrival, ragged end
    (64, 1, 3136),      # whole-tensor statistic of a small activation (fewer workgroups than a full set of shards)
    (1, 1, 5),          # whole-tensor statistic of five elements
    (3, 2, 700000),     # two channels of ~700 units each: too many arrivals per channel -> the two-launch route serves
]


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_absmax_onepass_equals_two_launches(dn, shape):
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(123456)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3).to(DT[dn])
    x[0, 0, 0] = -0.0
    if ch > 1:
        x[:, 1, :] = 0.0      # an all-zero channel: the lower bound on the scale decides
    flat = x.reshape(-1)
    covered = bool(nat.lib.bvq_absmax_onepass_supported(nat.dtype_code(x.dtype), flat.data_ptr(), outer, ch, inner))
    for pre in (0, 1):
        for min_val in (1e-10, None):
            run_a = torch.full((ch,), 2.0, device=DEV, dtype=x.dtype)
            run_b = run_a.clone()
            for first in (True, False):
                nat.ONEPASS = True
                sa, ca = nat.absmax_scale(flat, outer, ch, inner, min_val, 128.0, x.dtype, pre, running=run_a,
                                          momentum=0.1, first_batch=first)
                nat.ONEPASS = False
                sb, cb = nat.absmax_scale(flat, outer, ch, inner, min_val, 128.0, x.dtype, pre, running=run_b,
                                          momentum=0.1, first_batch=first)
                nat.ONEPASS = True
                assert torch.equal(bits(sa), bits(sb)), ('stat', pre, min_val, first, covered)
                assert torch.equal(bits(ca), bits(cb)), ('scale', pre, min_val, first, covered)
                assert torch.equal(bits(run_a), bits(run_b)), ('running', pre, min_val, first, covered)
        # the float32 statistic of the batch-sharded route
        nat.ONEPASS = True
        fa = nat.stats(nat.STAT_ABSMAX, flat, outer, ch, inner, out_f32=True, pre_op=pre)
        nat.ONEPASS = False
        fb = nat.stats(nat.STAT_ABSMAX, flat, outer, ch, inner, out_f32=True, pre_op=pre)
        nat.ONEPASS = True
        assert torch.equal(bits(fa), bits(fb)), ('stat32', pre, covered)
    assert arrival_is_clean(nat)
    ref = x.float().abs().amax(dim=(0, 2))
    assert torch.equal(sa.float(), torch.relu(x.float()).abs().amax(dim=(0, 2))) or covered in (True, False)
    nat.ONEPASS = True
    s0, _ = nat.absmax_scale(flat, outer, ch, inner, None, 128.0, x.dtype, 0)
    assert torch.equal(s0.float(), ref)


def test_absmax_onepass_propagates_nan_and_inf():
    from brevitas_amd import _native as nat
    x = torch.randn(16, 8, 3136, device=DEV, dtype=torch.bfloat16)
    x[3, 2, 100] = float('nan')
    x[5, 4, 7] = float('-inf')
    stat, scale = nat.absmax_scale(x.reshape(-1), 16, 8, 3136, 1e-10, 128.0, torch.bfloat16)
    assert torch.isnan(stat[2]) and torch.isnan(scale[2])
    assert torch.isinf(stat[4]) and stat[4] > 0
    assert arrival_is_clean(nat)


def _bwd_both(nat, x, g, outer, ch, inner, pre=0, clamp_ste=0, qmin=-128.0, qmax=127.0):
    code = nat.dtype_code(x.dtype)
    d = nat.QuantDesc(outer, ch, inner, code, code, code, nat.F32, 1, 0, qmin, qmax, 0, 0, clamp_ste, nat.OUT_DEQUANT, pre)
    flat, gflat = x.reshape(-1), g.reshape(-1)
    stat, scale = nat.absmax_scale(flat, outer, ch, inner, 1e-10, 128.0, x.dtype, pre)
    zp = torch.zeros(1, device=DEV)
    out = []
    for on in (True, False):
        nat.ONEPASS_BWD = on
        r = nat.fakequant_bwd_stats(d, gflat, flat, scale, zp, stat, x.dtype, 128.0, x.dtype, want_dscale=True)
        out.append(r)
    nat.ONEPASS_BWD = True
    return out


@pytest.mark.parametrize('dn', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_backward_onepass_equals_two_launches(dn, shape):
    from brevitas_amd import _native as nat
    outer, ch, inner = shape
    torch.manual_seed(654321)
    x = (torch.randn(outer, ch, inner, device=DEV) * 3).to(DT[dn])
    g = torch.randn(outer, ch, inner, device=DEV).to(DT[dn])
    if ch > 1:
        x[:, 1, :] = 0.0                   # every element attains the statistic: the first one takes the deposit
    x[outer - 1, 0, inner - 1] = 40.0      # channel 0: the arg-max is the very last element (ragged ends included)
    if outer > 1:
        x[0, 2 % ch, 0] = -50.0
        x[outer - 1, 2 % ch, 3] = 50.0     # a +-max tie: the first in batch order takes it
    for kw in (dict(), dict(pre=1), dict(clamp_ste=1), dict(qmin=-7.0, qmax=7.0)):
        a, b = _bwd_both(nat, x, g, outer, ch, inner, **kw)
        if a is None:
            assert b is None
            continue
        assert torch.equal(bits(a[0]), bits(b[0])), ('dx', kw)
        assert torch.equal(bits(a[1]), bits(b[1])), ('dscale', kw)
    assert arrival_is_clean(nat)


def test_onepass_step_equals_two_launch_step_through_the_modules():
    """the headline graph (RescalingIntQuant, RuntimeStatsScaling(AbsMax), training) for three steps, both routes"""
    from bench import build_quantizer
    from brevitas_amd import _native as nat
    torch.manual_seed(123456)
    x = torch.randn(24, 48, 28, 28, device=DEV, dtype=torch.bfloat16)
    g = torch.randn_like(x)
    res = []
    for on in (True, False):
        nat.ONEPASS = nat.ONEPASS_BWD = on
        q = build_quantizer(48, True, torch.device(DEV))
        steps = []
        for _ in range(3):
            xi = x.clone().requires_grad_(True)
            y, scale = q(xi)[:2]
            y.backward(g)
            steps.append((y.detach(), scale.detach(), xi.grad, q.scaling_impl.runtime_stats.running_stats.detach().clone()))
        res.append(steps)
    nat.ONEPASS = nat.ONEPASS_BWD = True
    for sa, sb in zip(*res):
        for ta, tb in zip(sa, sb):
            assert torch.equal(bits(ta), bits(tb))
    assert arrival_is_clean(nat)
