"""More than 2^31 elements in one call (a 4.6 GB bf16 activation; MI355X has 288 GB): the statistic, forward and
backward index with 64-bit offsets end to end.  Checked against torch's reduction and against the library's
own results on the first and last slices quantized separately (those paths are pinned by the golden vectors)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('layout', ['per_tensor', 'per_channel'])
def test_more_than_2_31_elements(layout):
    from brevitas_amd import _native as nat
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip('needs 40 GB of free device memory')
    outer, ch, inner = 9, 64, 4_000_000  # 2.304e9 elements
    n = outer * ch * inner
    assert n > 2 ** 31
    dt = torch.bfloat16
    x = torch.empty(n, device=DEV, dtype=dt)
    blk = ch * inner
    gen = torch.Generator(device=DEV).manual_seed(123456)
    for o in range(outer):  # filled in pieces: a float32 randn of the whole tensor would need another 9 GB
        x[o * blk:(o + 1) * blk] = torch.randn(blk, device=DEV, generator=gen).to(dt)
    x[n - 5] = 9.5  # the maximum of the last channel sits in the very last elements
    g = torch.empty_like(x)
    for o in range(outer):
        g[o * blk:(o + 1) * blk] = torch.randn(blk, device=DEV, generator=gen).to(dt)
    pc = layout == 'per_channel'
    lay = (outer, ch, inner) if pc else (1, 1, n)
    stat = nat.stats(nat.STAT_ABSMAX, x, *lay)
    want = x.view(outer, ch, inner).abs().amax(dim=(0, 2)) if pc else x.abs().max().reshape(1)
    assert torch.equal(stat, want)
    assert stat[-1].item() == 9.5
    scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)
    zp = torch.zeros(1, device=DEV)
    code = nat.dtype_code(dt)
    d = nat.QuantDesc(*lay, code, code, code, 0, int(pc), 0, -128.0, 127.0, 0, 0, 0, 0)
    y = nat.fakequant_fwd(d, x, scale, zp)
    dx, ds, _ = nat.fakequant_bwd(d, g, x, scale, zp, True, False)[:3]
    # the same elements through calls that stay far below 2^31: first and last row of the last channel
    for lo in (0, n - inner):
        c = (lo // inner) % ch if pc else 0
        sub = nat.QuantDesc(1, 1, inner, code, code, code, 0, 0, 0, -128.0, 127.0, 0, 0, 0, 0)
        s1 = scale[c:c + 1].contiguous()
        y1 = nat.fakequant_fwd(sub, x[lo:lo + inner], s1, zp)
        dx1 = nat.fakequant_bwd(sub, g[lo:lo + inner], x[lo:lo + inner], s1, zp, False, False)[0]
        assert torch.equal(y[lo:lo + inner], y1) and torch.equal(dx[lo:lo + inner], dx1)
    # dscale of the last channel against float64 sums of the per-element terms over its rows, piecewise
    c = ch - 1 if pc else 0
    tot = 0.0
    for o in range(outer):
        for cc in ([c] if pc else range(ch)):
            lo = (o * ch + cc) * inner
            sub = nat.QuantDesc(1, 1, inner, code, code, code, 0, 0, 0, -128.0, 127.0, 0, 0, 0, 0)
            part = nat.fakequant_bwd(sub, g[lo:lo + inner], x[lo:lo + inner], scale[c:c + 1].contiguous(), zp, True,
                                     False)[1]
            tot += float(part.double().sum())
    got = float(ds[c])
    assert abs(got - tot) <= 1e-3 * max(1.0, abs(tot)), (got, tot)


def test_percentile_of_more_than_2_31_elements():
    """the radix select's counters and rank are 64-bit: the k-th smallest |x| of 2.3e9 values, verified by counting"""
    from brevitas_amd import _native as nat
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip('needs 40 GB of free device memory')
    n = 2_304_000_000
    x = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(123457)
    step = 256_000_000
    for lo in range(0, n, step):
        x[lo:lo + step] = torch.randn(min(step, n - lo), device=DEV, generator=gen).to(torch.bfloat16)
    k = int(0.99999 * n + 0.5)
    v = nat.kth_value(x, k, 1, 1, n, True)
    below = at_most = 0
    for lo in range(0, n, step):
        a = x[lo:lo + step].abs()
        below += int((a < v).sum())
        at_most += int((a <= v).sum())
    assert below < k <= at_most, (below, k, at_most)
