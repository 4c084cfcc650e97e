"""Seeded random sweep of the C-ABI hot path against the CPU oracle: shapes chosen to hit every tiling the
library has (long rows cut into pieces, one row per unit, several rows per unit, ragged rows with 16-byte
accesses, element-granular rows, column-mapped layouts, single-workgroup one-launch forward), every dtype,
rounding mode, clamp flavour, bit width, zero-point kind and the fused ReLU.  Bit-exact for the statistic,
y and dx; the reduced gradients within float32 summation error."""
import os

import numpy as np
import pytest
import torch

from test_gpu_cabi import ndesc, to_np

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'f32': torch.float32, 'bf16': torch.bfloat16, 'f16': torch.float16}


def _cases(n=72, seed=123456):
    rng = np.random.RandomState(seed)
    inner_choices = [1, 2, 3, 7, 8, 12, 16, 25, 49, 64, 100, 196, 200, 392, 1000, 3136, 4608, 9000]
    out = []
    for i in range(n):
        inner = int(inner_choices[rng.randint(len(inner_choices))])
        channels = int([1, 2, 3, 8, 16, 33, 64][rng.randint(7)])
        budget = 60000 // max(1, channels * inner)
        outer = int(max(1, min(budget, [1, 2, 5, 17, 64, 300][rng.randint(6)])))
        out.append(dict(
            i=i, outer=outer, channels=channels, inner=inner, dn=['f32', 'bf16', 'f16'][rng.randint(3)],
            rm=int(rng.randint(5)), clamp_ste=int(rng.randint(2)), pre=int(rng.randint(2)),
            bits=int([2, 4, 8, 8, 8][rng.randint(5)]), signed=int(rng.randint(2)), narrow=int(rng.randint(2)),
            zp_kind=['zero', 'zero', 'scalar', 'channel'][rng.randint(4)], seed=int(rng.randint(1 << 30))))
    return out


# BVQ_FUZZ_CASES / BVQ_FUZZ_SEED widen the sweep for a one-off soak run (the defaults are what the suite runs)
CASES = _cases(int(os.environ.get('BVQ_FUZZ_CASES', '72')), int(os.environ.get('BVQ_FUZZ_SEED', '123456')))


@pytest.mark.parametrize('c', CASES, ids=lambda c: '%d-%dx%dx%d-%s' % (c['i'], c['outer'], c['channels'], c['inner'], c['dn']))
def test_random_case_against_oracle(oracle, c):
    from brevitas_amd import _native as nat
    O = oracle
    dt = DT[c['dn']]
    code = nat.dtype_code(dt)
    outer, ch, inner = c['outer'], c['channels'], c['inner']
    g = torch.Generator().manual_seed(c['seed'])
    x = (torch.randn(outer, ch, inner, generator=g) * 2).to(dt)
    gr = torch.randn(outer, ch, inner, generator=g).to(dt)
    if c['signed']:
        qmin = -(2 ** (c['bits'] - 1)) + (1 if c['narrow'] else 0)
        qmax = 2 ** (c['bits'] - 1) - 1
    else:
        qmin, qmax = 0, 2 ** c['bits'] - 1 - (1 if c['narrow'] else 0)
    pc = ch > 1
    xd, gd = x.to(DEV).reshape(-1), gr.to(DEV).reshape(-1)
    # statistic
    stat = nat.stats(nat.STAT_ABSMAX, xd, outer if pc else 1, ch if pc else 1, inner if pc else x.numel(), pre_op=c['pre'])
    xn, _ = O.from_torch(x.reshape(-1))
    gn, _ = O.from_torch(gr.reshape(-1))
    lay = (outer, ch, inner) if pc else (1, 1, x.numel())
    want_stat = O.stats(O.STAT_ABSMAX, xn, code, *lay, pre_op=c['pre'])
    assert np.array_equal(stat.float().cpu().numpy(), want_stat)
    scale = (torch.clamp_min(stat.float(), 1e-3) / float(max(abs(qmin), abs(qmax)))).to(dt)
    if c['zp_kind'] == 'zero':
        zp = torch.zeros(1)
    elif c['zp_kind'] == 'scalar':
        zp = torch.tensor([3.0])
    else:
        zp = torch.randint(-3, 4, (ch if pc else 1,)).float()
    zp_pc = c['zp_kind'] == 'channel' and pc
    od = O.make_desc(*lay, code, code, code, O.F32, scale_per_channel=pc, zp_per_channel=zp_pc, qmin=float(qmin),
                     qmax=float(qmax), round_mode=c['rm'], clamp_ste=bool(c['clamp_ste']), pre_op=c['pre'])
    d = ndesc(nat, od)
    sn, _ = O.from_torch(scale.cpu().reshape(-1))
    zn = zp.numpy().astype(np.float32)
    y_o, codes_o = O.fakequant_fwd(od, xn, sn, zn)
    dx_o, ds_o, dz_o = O.fakequant_bwd(od, gn, xn, sn, zn)
    zd = zp.to(DEV)
    y, codes = nat.fakequant_fwd(d, xd, scale, zd, want_codes=True)
    assert np.array_equal(to_np(y), y_o), 'y'
    assert np.array_equal(to_np(codes), codes_o), 'codes'
    y2 = nat.fakequant_fwd(d, xd, scale, zd)  # the route without codes (column-mapped where applicable)
    assert np.array_equal(to_np(y2), y_o), 'y (no codes)'
    dx, ds, dz = nat.fakequant_bwd(d, gd, xd, scale, zd, True, True)
    assert np.array_equal(to_np(dx), dx_o), 'dx'
    dx3, ds3, _ = nat.fakequant_bwd(d, gd, xd, scale, zd, True, False)  # no dzp: column-mapped where applicable
    assert np.array_equal(to_np(dx3), dx_o), 'dx (dscale only)'
    for got, want, name in ((ds, ds_o, 'dscale'), (ds3, ds_o, 'dscale (no dzp)'), (dz, dz_o, 'dzp')):
        got = got.double().cpu().numpy().reshape(-1)
        want = want.astype(np.float64).reshape(-1)
        if got.size != want.size:
            got = np.array([got.sum()])
        ok = np.isfinite(want)
        mag = np.abs(want[ok]).max() if ok.any() else 0.0
        lim = {'f32': 2e-4, 'bf16': 2e-2, 'f16': 5e-3}[c['dn']]
        assert np.all(np.abs(got[ok] - want[ok]) <= lim * (np.abs(want[ok]) + mag + 1.0)), name


def _stat_cases(n=48, seed=4242):
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        inner = int([1, 2, 3, 7, 8, 16, 49, 64, 100, 196, 1000, 3136, 9000][rng.randint(13)])
        channels = int([1, 2, 3, 8, 16, 33, 64, 512][rng.randint(8)])
        budget = 200000 // max(1, channels * inner)
        outer = int(max(1, min(budget, [1, 2, 5, 17, 64, 300, 2000][rng.randint(7)])))
        out.append(dict(i=i, outer=outer, channels=channels, inner=inner, dn=['f32', 'bf16', 'f16'][rng.randint(3)],
                        pre=int(rng.randint(2)), seed=int(rng.randint(1 << 30)), k_frac=float(rng.rand())))
    return out


@pytest.mark.parametrize('c', _stat_cases(), ids=lambda c: '%d-%dx%dx%d-%s' % (c['i'], c['outer'], c['channels'], c['inner'], c['dn']))
def test_random_statistics_against_torch(c):
    """min/max, abs-max and the k-th value on the same random layouts (row-mapped, ragged, column-mapped) against
    torch's own reductions, which are exact"""
    from brevitas_amd import _native as nat
    dt = DT[c['dn']]
    outer, ch, inner = c['outer'], c['channels'], c['inner']
    g = torch.Generator().manual_seed(c['seed'])
    x = (torch.randn(outer, ch, inner, generator=g) * 2 + 0.3).to(dt).to(DEV)
    src = torch.relu(x) if c['pre'] else x
    flat = x.reshape(-1)
    mm = nat.stats(nat.STAT_MINMAX, flat, outer, ch, inner, pre_op=c['pre']).view(2, ch)
    assert torch.equal(mm[0], src.amax(dim=(0, 2))) and torch.equal(mm[1], src.amin(dim=(0, 2)))
    am = nat.stats(nat.STAT_ABSMAX, flat, outer, ch, inner, pre_op=c['pre'])
    assert torch.equal(am, src.abs().amax(dim=(0, 2)))
    whole = nat.stats(nat.STAT_MINMAX, flat, 1, 1, flat.numel(), pre_op=c['pre'])
    assert whole[0] == src.max() and whole[1] == src.min()
    per = outer * inner
    k = max(1, min(per, int(c['k_frac'] * per) + 1))
    for abs_key in (True, False):
        got = nat.kth_value(flat, k, outer, ch, inner, abs_key)
        v = (x.abs() if abs_key else x).permute(1, 0, 2).reshape(ch, per).float()
        assert torch.equal(got.float(), v.kthvalue(k, dim=1)[0]), abs_key
