"""Per-tensor route of bvq_kth_value (>= 4M elements: 32768-bin LDS histogram of the key's top 15 bits, the low
bits through global counters) against torch's sort order: ranks at both ends and in the bulk, a ragged tail,
NaNs (sorted last, like torch.kthvalue), infinities, signed and absolute keys, every dtype -- and identical to the
digit-pass route (a view one element off the 16-byte grid takes that one)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}


def _off_grid(x):
    off = torch.empty(x.numel() + 8, device=DEV, dtype=x.dtype)[1:x.numel() + 1]
    off.copy_(x)
    assert off.data_ptr() % 16 != 0
    return off


@pytest.mark.parametrize('abs_key', [True, False], ids=['abs', 'signed'])
@pytest.mark.parametrize('dn', ['bf16', 'f16', 'f32'])
@pytest.mark.parametrize('n', [1 << 22, 6_000_003])
def test_wide_kth_value(dn, n, abs_key):
    from brevitas_amd import _native as nat
    dt = DT[dn]
    g = torch.Generator(device=DEV).manual_seed(123456 + n)
    x = torch.empty(n + 8, device=DEV, dtype=dt)[:n]
    x.copy_((torch.randn(n, device=DEV, generator=g) * 3).to(dt))
    x[7] = float('inf')
    x[n - 1] = -float('inf')
    x[n // 2] = float('nan')
    x[11] = 0.0
    x[12] = -0.0
    assert x.data_ptr() % 16 == 0
    off = _off_grid(x)
    ref_sorted = torch.sort((x.abs() if abs_key else x).float())[0]  # torch sorts NaN last too
    nan_count = int(torch.isnan(x).sum())
    for k in (1, 2, n // 3, n // 2, int(0.99999 * n + 0.5), n - nan_count - 2, n - nan_count, n):
        got = nat.kth_value(x, k, 1, 1, n, abs_key)
        two = nat.kth_value(off, k, 1, 1, n, abs_key)
        if k > n - nan_count:
            assert torch.isnan(got).all() and torch.isnan(two).all()
        else:
            assert got.float().item() == ref_sorted[k - 1].item(), k
            assert two.float().item() == got.float().item(), k


@pytest.mark.parametrize('dn', ['bf16', 'f32'])
def test_wide_kth_value_on_constant_runs(dn):
    """half zeros (a post-ReLU activation) and a constant tensor: every element lands in one bin"""
    from brevitas_amd import _native as nat
    dt = DT[dn]
    n = 5_000_000
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.relu(torch.randn(n, device=DEV, generator=g)).to(dt)
    srt = torch.sort(x.float())[0]
    for k in (1, n // 4, n // 2, int(0.9 * n)):
        for abs_key in (True, False):
            assert nat.kth_value(x, k, 1, 1, n, abs_key).float().item() == srt[k - 1].item()
    c = torch.full((n,), 1.37, device=DEV).to(dt)
    for k in (1, n // 2, n):
        assert nat.kth_value(c, k, 1, 1, n, False).item() == c[0].item()


@pytest.mark.parametrize('dn', ['bf16', 'f16', 'f32'])
@pytest.mark.parametrize('layout', [(1, 1, 6_000_003), (1, 1, 70_000), (40, 16, 900)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('abs_key', [False, True], ids=['signed', 'abs'])
def test_kth_pair_equals_two_selections(dn, layout, abs_key):
    """bvq_kth_pair (PercentileInterval's two ranks from one histogram read) against two bvq_kth_value calls,
    including both ranks in the same bin and equal ranks"""
    from brevitas_amd import _native as nat
    dt = DT[dn]
    outer, ch, inner = layout
    n = outer * inner
    g = torch.Generator(device=DEV).manual_seed(99 + inner)
    x = (torch.randn(outer * ch * inner, device=DEV, generator=g) * 2 - 0.5).to(dt)
    for k1, k2 in ((max(1, int(1e-5 * n)), int(0.99999 * n + 0.5)), (n // 2, n // 2 + 1), (7, 7), (n, 1)):
        both = nat.kth_pair(x, k1, k2, outer, ch, inner, abs_key)
        assert torch.equal(both[0], nat.kth_value(x, k1, outer, ch, inner, abs_key)), (k1, k2)
        assert torch.equal(both[1], nat.kth_value(x, k2, outer, ch, inner, abs_key)), (k1, k2)


def test_percentile_interval_module_uses_one_pass_and_matches_torch():
    from brevitas_amd.core.stats.stats_op import PercentileInterval
    n = 5_000_000
    x = (torch.randn(n, device=DEV) * 3 + 1).requires_grad_(True)
    m = PercentileInterval(low_percentile_q=0.01, high_percentile_q=99.99)
    out = m(x)
    k_low, k_high = 500, int(0.9999 * n + 0.5)
    lo, hi = x.detach().kthvalue(k_low)[0], x.detach().kthvalue(k_high)[0]
    assert out.item() == (hi - lo).abs().item()
    out.backward()
    nz = x.grad.nonzero().reshape(-1)
    assert nz.numel() == 2 and sorted(x.grad[nz].tolist()) == [-1.0, 1.0]
    assert x.detach()[x.grad > 0].item() == hi.item() and x.detach()[x.grad < 0].item() == lo.item()
