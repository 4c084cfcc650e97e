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
