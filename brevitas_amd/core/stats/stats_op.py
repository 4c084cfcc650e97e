"""Scale statistics on the accelerated path: AbsMax and AbsMinMax (B/core/stats/stats_op.py:129-158).

Forward: one streaming read of the input by the HIP reduction (no |x| temporary, no second pass).
Backward: what autograd derives from torch.max / torch.min / torch.abs in the reference -- the
gradient lands on the elements attaining the extremum (first one along a reduced dim, evenly over
all ties for a whole-tensor reduction).
"""
import math
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from brevitas_amd import _aten
from brevitas_amd import _native as nat
from brevitas_amd.core._state import TolerantLoad


def _on_cpu(x: Tensor, group=None, sharded_ok=False) -> bool:
    """CPU tensors take the pure-torch route (brevitas_amd._aten: the reference's own op composition)"""
    if x.is_cuda:
        return False
    if group is not None and not sharded_ok:
        raise NotImplementedError('this batch-sharded statistic runs on device tensors only')
    return True


def _abs_max(x: Tensor, dim, group=None) -> Tensor:
    if _on_cpu(x, group, sharded_ok=True):
        return _aten.abs_max(x, dim) if group is None else _aten.sharded_abs_max(x, dim, group)
    return _AbsMaxFn.apply(x, dim, group)


def _min_max(x: Tensor, dim, group=None):
    if _on_cpu(x, group, sharded_ok=True):
        return _aten.min_max(x, dim) if group is None else _aten.sharded_min_max(x, dim, group)
    return _MinMaxFn.apply(x, dim, group)


def _as_rows(x: Tensor, dim: Optional[int]):
    """-> (contiguous tensor, outer, channels, inner, output shape) for a reduction over `dim`
    (None: everything).  The kernel keeps the middle axis of [outer, channels, inner]."""
    if dim is None:
        xc = x.contiguous()
        return xc, 1, 1, xc.numel(), ()
    dim = dim % x.dim()
    out_shape = tuple(s for i, s in enumerate(x.shape) if i != dim)
    if x.dim() == 2 and dim == 1:
        xc = x.contiguous()
        return xc, 1, x.shape[0], x.shape[1], out_shape
    if x.dim() == 2 and dim == 0:
        xc = x.contiguous()
        return xc, x.shape[0], x.shape[1], 1, out_shape
    # general case: bring the reduced axis last, flatten the kept axes (one copy, like a
    # non-contiguous reshape in the reference)
    xc = x.movedim(dim, -1).reshape(-1, x.shape[dim]).contiguous()
    return xc, 1, xc.shape[0], xc.shape[1], out_shape


def _sharded_stat_bwd(match, flat, stat, gstat, outer, ch, inner, group, first_only):
    """backward of a max / min / k-th-value statistic of a batch-sharded tensor: the (summed) gradient goes
    where the single-device run on the concatenated batch puts it -- the lowest rank holding the first
    attaining element (or evenly over the ties of all shards for a whole-tensor max / min)."""
    from brevitas_amd.distributed import sync_backward
    flags = match | (nat.MATCH_FIRST if first_only else 0)
    dx = torch.empty_like(flat)
    info = nat.stat_tie_scan(flags, flat, stat, outer, ch, inner, dx_zero_fill=dx)
    gsum, info, total = sync_backward(gstat.reshape(-1).float(), info, ch, group, first_only)
    return nat.stat_tie_apply(flags, flat, stat, gsum.to(flat.dtype), info, dx, outer, ch, inner, 0, total)


class _AbsMaxFn(Function):

    @staticmethod
    def forward(ctx, x, dim, group=None):
        xc, outer, ch, inner, out_shape = _as_rows(x, dim)
        if group is None:
            stat = nat.stats(nat.STAT_ABSMAX, xc.reshape(-1), outer, ch, inner)
        else:
            from brevitas_amd.distributed import sync_stat_max
            stat = nat.stats(nat.STAT_ABSMAX, xc.reshape(-1), outer, ch, inner, out_f32=True)
            stat = sync_stat_max(stat, group).to(x.dtype)
        ctx.layout = (outer, ch, inner, dim)
        ctx.group = group
        ctx.save_for_backward(x, stat)
        return stat.reshape(out_shape)

    @staticmethod
    def backward(ctx, gstat):
        x, stat = ctx.saved_tensors
        outer, ch, inner, dim = ctx.layout
        xc, _, _, _, _ = _as_rows(x, dim)
        if ctx.group is not None:
            dx = _sharded_stat_bwd(nat.MATCH_ABS, xc.reshape(-1), stat, gstat, outer, ch, inner, ctx.group, False)
        else:
            dx = nat.stat_bwd(nat.MATCH_ABS, xc.reshape(-1), stat, gstat.reshape(-1), outer, ch, inner)
        return _unrows(dx, x, dim), None, None


class _MinMaxFn(Function):
    """returns (max, min) of x, each with the backward of torch.max / torch.min"""

    @staticmethod
    def forward(ctx, x, dim, group=None):
        xc, outer, ch, inner, out_shape = _as_rows(x, dim)
        if group is None:
            raw = nat.stats(nat.STAT_MINMAX, xc.reshape(-1), outer, ch, inner)
            mx, mn = raw[:ch], raw[ch:]
        else:
            from brevitas_amd.distributed import sync_stat_max, sync_stat_min
            raw = nat.stats(nat.STAT_MINMAX, xc.reshape(-1), outer, ch, inner, out_f32=True)
            mx = sync_stat_max(raw[:ch].contiguous(), group).to(x.dtype)
            mn = sync_stat_min(raw[ch:].contiguous(), group).to(x.dtype)
        ctx.layout = (outer, ch, inner, dim)
        ctx.group = group
        ctx.save_for_backward(x, mx, mn)
        return mx.reshape(out_shape), mn.reshape(out_shape)

    @staticmethod
    def backward(ctx, gmax, gmin):
        x, mx, mn = ctx.saved_tensors
        outer, ch, inner, dim = ctx.layout
        xc, _, _, _, _ = _as_rows(x, dim)
        flat = xc.reshape(-1)
        if ctx.group is not None:
            dx = _sharded_stat_bwd(nat.MATCH_VALUE, flat, mx, gmax, outer, ch, inner, ctx.group, False)
            dx = dx + _sharded_stat_bwd(nat.MATCH_VALUE, flat, mn, gmin, outer, ch, inner, ctx.group, False)
        else:
            dx = nat.stat_bwd(nat.MATCH_VALUE, flat, mx, gmax.reshape(-1), outer, ch, inner)
            dx = nat.stat_bwd(nat.MATCH_VALUE, flat, mn, gmin.reshape(-1), outer, ch, inner, dx=dx)
        return _unrows(dx, x, dim), None, None


def _unrows(dx_flat: Tensor, x: Tensor, dim: Optional[int]) -> Tensor:
    """inverse of _as_rows for the gradient"""
    if dim is None or x.dim() == 2:
        return dx_flat.reshape(x.shape)
    dim = dim % x.dim()
    moved_shape = tuple(s for i, s in enumerate(x.shape) if i != dim) + (x.shape[dim],)
    return dx_flat.reshape(moved_shape).movedim(-1, dim)


class _KthValueFn(Function):
    """k-th smallest of |x| (abs_key) or x: torch.kthvalue(k).values on the flat input or along `dim` of a
    2-D one.  Backward: the gradient goes to one element attaining the value -- the first in memory
    order (torch's choice among equal values is implementation-defined).

    rank: an int k, or (rule, q) with rule in {nat.KTH_HIGH, nat.KTH_LOW} for a batch-sharded tensor, where
    the rank follows from the global element count on the device (include/bvq.h, bvq_kth_rule)."""

    @staticmethod
    def forward(ctx, x, rank, dim, abs_key, group=None):
        xc, outer, ch, inner, out_shape = _as_rows(x, dim)
        if group is None:
            val = nat.kth_value(xc.reshape(-1), rank, outer, ch, inner, abs_key)
        else:
            from brevitas_amd.distributed import sharded_kth_value
            rule, q = rank
            # a whole-tensor statistic takes the 15-bit first digit (one read of a 16-bit |x|); the choice depends on
            # the layout alone, so every shard makes the same one
            steps = nat.KthWideSteps(xc.reshape(-1), abs_key, rule, q) if ch == 1 else \
                nat.KthSelectSteps(xc.reshape(-1), outer, ch, inner, abs_key, rule, q)
            val = sharded_kth_value(steps, group)
        ctx.layout = (outer, ch, inner, dim, abs_key)
        ctx.group = group
        ctx.save_for_backward(x, val)
        return val.reshape(out_shape)

    @staticmethod
    def backward(ctx, gval):
        x, val = ctx.saved_tensors
        outer, ch, inner, dim, abs_key = ctx.layout
        xc, _, _, _, _ = _as_rows(x, dim)
        kind = nat.MATCH_ABS if abs_key else nat.MATCH_VALUE
        if ctx.group is not None:
            dx = _sharded_stat_bwd(kind, xc.reshape(-1), val, gval, outer, ch, inner, ctx.group, True)
        else:
            dx = nat.stat_bwd(kind | nat.MATCH_FIRST, xc.reshape(-1), val, gval.reshape(-1), outer, ch, inner)
        return _unrows(dx, x, dim), None, None, None, None


class _KthPairFn(Function):
    """two ranks of x in one call (nat.kth_pair: one histogram read for a big flat input) -> (k_first-th value,
    k_second-th value); backward as two _KthValueFn backwards added up"""

    @staticmethod
    def forward(ctx, x, k_first, k_second, dim):
        xc, outer, ch, inner, out_shape = _as_rows(x, dim)
        both = nat.kth_pair(xc.reshape(-1), k_first, k_second, outer, ch, inner, False)
        ctx.layout = (outer, ch, inner, dim)
        ctx.save_for_backward(x, both)
        return both[0].reshape(out_shape), both[1].reshape(out_shape)

    @staticmethod
    def backward(ctx, g_first, g_second):
        x, both = ctx.saved_tensors
        outer, ch, inner, dim = ctx.layout
        xc, _, _, _, _ = _as_rows(x, dim)
        flat = xc.reshape(-1)
        kind = nat.MATCH_VALUE | nat.MATCH_FIRST
        dx = nat.stat_bwd(kind, flat, both[0].contiguous(), g_first.reshape(-1), outer, ch, inner)
        dx = nat.stat_bwd(kind, flat, both[1].contiguous(), g_second.reshape(-1), outer, ch, inner, dx=dx)
        return _unrows(dx, x, dim), None, None, None


def _kth(x: Tensor, k: int, dim: Optional[int], abs_key: bool) -> Tensor:
    return _KthValueFn.apply(x, k, dim, abs_key)


def _percentile(module, x: Tensor, rule: int, q: float, abs_key: bool) -> Tensor:
    """the percentile statistics' k-th value: k = floor(.01*q*n + .5) (KTH_HIGH) or ceil(.01*q*n) (KTH_LOW)
    of the n elements each value is selected from -- all shards' elements if the module is batch-sharded"""
    dim = module.stats_reduce_dim
    group = getattr(module, 'bvq_shard_group', None)
    if _on_cpu(x, group):
        if abs_key:
            return _aten.abs_percentile(x, q, dim)
        return _aten.high_percentile(x, q, dim) if rule == nat.KTH_HIGH else _aten.low_percentile(x, q, dim)
    if group is not None:
        return _KthValueFn.apply(x, (rule, q), dim, abs_key, group)
    n = _numel_along(x, dim)
    if rule == nat.KTH_HIGH:
        # k is 1-indexed, so round away from zero
        k = int(math.floor(.01 * q * n + 0.5))
    else:
        k = int(math.ceil(.01 * q * n))
    return _KthValueFn.apply(x, k, dim, abs_key)


def _numel_along(x: Tensor, dim: Optional[int]) -> int:
    """how many elements each k-th value is selected from"""
    if dim is None:
        return x.numel()
    assert len(x.size()) == 2, "Only 2-dim input is supported."
    return x.shape[dim]


class AbsPercentile(torch.nn.Module):
    """high_percentile_q-th percentile of |x| (B/core/stats/stats_op.py:41-66): the k-th smallest with
    k = floor(.01 * q * n + 0.5), an exact radix select on the device instead of torch.kthvalue"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, high_percentile_q: float, stats_reduce_dim: Optional[int], percentile_q=None):
        super().__init__()
        if percentile_q is not None:
            raise RuntimeError("percentile_q is deprecated, please pass high_percentile_q.")
        assert high_percentile_q <= 100, "q has to be a percentage"
        self.q = high_percentile_q
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: Tensor):
        return _percentile(self, x, nat.KTH_HIGH, self.q, True)


class NegativePercentileOrZero(torch.nn.Module):
    """min(low_percentile_q-th percentile of x, 0) (B/core/stats/stats_op.py:69-94)"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, low_percentile_q, stats_reduce_dim: Optional[int] = None) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim
        self.q = low_percentile_q

    def forward(self, x: Tensor) -> Tensor:
        result = _percentile(self, x, nat.KTH_LOW, self.q, False)
        zero = torch.zeros((), dtype=result.dtype, device=result.device)
        return torch.where(result <= zero, result, zero)


class PercentileInterval(torch.nn.Module):
    """|high percentile - low percentile| of x (B/core/stats/stats_op.py:97-126)"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, low_percentile_q, high_percentile_q, stats_reduce_dim: Optional[int] = None) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim
        self.low_q = low_percentile_q
        self.high_q = high_percentile_q

    def forward(self, x: Tensor) -> Tensor:
        if getattr(self, 'bvq_shard_group', None) is None and x.is_cuda:
            # both ranks from one pass over x (ranks as in _percentile)
            n = _numel_along(x, self.stats_reduce_dim)
            k_low = int(math.ceil(.01 * self.low_q * n))
            k_high = int(math.floor(.01 * self.high_q * n + 0.5))
            low_result, high_result = _KthPairFn.apply(x, k_low, k_high, self.stats_reduce_dim)
        else:
            low_result = _percentile(self, x, nat.KTH_LOW, self.low_q, False)
            high_result = _percentile(self, x, nat.KTH_HIGH, self.high_q, False)
        return torch.abs(high_result - low_result)


class NegativeMinOrZero(torch.nn.Module):
    """min(min(x), 0) over the whole input or along `stats_reduce_dim` (B/core/stats/stats_op.py:21-38): the
    (negated) offset of asymmetric quantizers.  One streaming read by the min/max reduction."""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, stats_reduce_dim: Optional[int] = None) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: Tensor) -> Tensor:
        _, min_val = _min_max(x, self.stats_reduce_dim, getattr(self, 'bvq_shard_group', None))
        zero = torch.zeros((), dtype=min_val.dtype, device=min_val.device)
        return torch.where(min_val <= zero, min_val, zero)


class AbsMax(torch.nn.Module):
    """max(|x|) over the whole (1-D) input, or along `stats_reduce_dim` of a [C, K] view"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, stats_reduce_dim: Optional[int] = None) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: Tensor):
        return _abs_max(x, self.stats_reduce_dim, getattr(self, 'bvq_shard_group', None))


class AbsMinMax(torch.nn.Module):
    """|max(x) - min(x)| over the whole input or along `stats_reduce_dim`"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, stats_reduce_dim: Optional[int] = None) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: Tensor):
        max_val, min_val = _min_max(x, self.stats_reduce_dim, getattr(self, 'bvq_shard_group', None))
        return torch.abs(max_val - min_val)


class AbsMaxAve(torch.nn.Module):
    """mean over channels of the per-channel abs-max (B/core/stats/stats_op.py:161-170): the streaming
    abs-max reduction followed by a mean over C values"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, stats_reduce_dim: int) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: Tensor):
        return torch.mean(_abs_max(x, self.stats_reduce_dim, getattr(self, 'bvq_shard_group', None)))


class AbsMaxL2(torch.nn.Module):
    """L2 norm of the per-channel abs-max over sqrt(C) (B/core/stats/stats_op.py:173-185)"""
    bvq_is_stat = True
    bvq_shardable_stat = True  # brevitas_amd.distributed.shard_over_batch

    def __init__(self, stats_reduce_dim: int) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: torch.Tensor):
        per_channel_max = _abs_max(x, self.stats_reduce_dim, getattr(self, 'bvq_shard_group', None))
        out = torch.norm(per_channel_max, p=2)
        return out / math.sqrt(per_channel_max.view(-1).shape[0])


class _AbsMomentsFn(Function):
    """(mean |x|, unbiased variance of |x|) over the whole input or along `dim`: one streaming read
    (bvq_abs_moments) instead of abs + mean + var; backward one read + one write (bvq_abs_affine_bwd):
    dx = sgn(x) * (gmean / n + gvar * 2 (|x| - mean) / (n - 1))"""

    @staticmethod
    def forward(ctx, x, dim):
        xc, outer, ch, inner, out_shape = _as_rows(x, dim)
        n = outer * inner
        # sums of d = |x| - pivot (the channel's first |x|): shifted, so the variance keeps its digits when the
        # mean is far larger than the spread (torch.var is two-pass / Welford)
        sums = nat.abs_moments(xc.reshape(-1), outer, ch, inner).double()
        d1, d2, pivot = sums[:ch], sums[ch:2 * ch], sums[2 * ch:]
        mean = pivot + d1 / n
        var = (d2 - d1 * (d1 / n)) / (n - 1) if n > 1 else torch.full_like(mean, float('nan'))
        var = var.clamp_min(0.0)  # rounding of the two sums can leave a tiny negative difference
        ctx.layout = (outer, ch, inner, dim, n)
        ctx.save_for_backward(x, mean)
        return mean.to(x.dtype).reshape(out_shape), var.to(x.dtype).reshape(out_shape)

    @staticmethod
    def backward(ctx, gmean, gvar):
        x, mean = ctx.saved_tensors
        outer, ch, inner, dim, n = ctx.layout
        xc, _, _, _, _ = _as_rows(x, dim)
        zero = torch.zeros(ch, dtype=torch.float64, device=x.device)
        gm = gmean.reshape(-1).double() if gmean is not None else zero
        gv = gvar.reshape(-1).double() if gvar is not None else zero
        b = 2.0 * gv / (n - 1) if n > 1 else zero
        a = gm / n - b * mean
        dx = nat.abs_affine_bwd(xc.reshape(-1), a.float(), b.float(), outer, ch, inner)
        return _unrows(dx, x, dim), None


def _abs_moments(x: Tensor, dim):
    return _aten.abs_mean_var(x, dim) if _on_cpu(x) else _AbsMomentsFn.apply(x, dim)


class AbsAve(torch.nn.Module):
    """mean(|x|) over the whole input or along `stats_reduce_dim` (B/core/stats/stats_op.py:186-199)"""
    bvq_is_stat = True

    def __init__(self, stats_reduce_dim: Optional[int] = None) -> None:
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim

    def forward(self, x: Tensor):
        return _abs_moments(x, self.stats_reduce_dim)[0]


DEFAULT_STD_DEV_EPSILON = 1e-8


class _MeanSigmaStdImpl(torch.nn.Module):
    """mean(|x|) + sigma * sqrt(var(|x|) + eps) (B/core/stats/stats_op.py:219-240)"""

    def __init__(self, stats_reduce_dim: Optional[int] = None, std_dev_epsilon: float = DEFAULT_STD_DEV_EPSILON):
        super().__init__()
        self.stats_reduce_dim = stats_reduce_dim
        self.epsilon = std_dev_epsilon

    def forward(self, x: Tensor, sigma: Tensor):
        mean_val, var_val = _abs_moments(x, self.stats_reduce_dim)
        std_val = torch.sqrt(var_val + self.epsilon)
        if self.stats_reduce_dim is not None:
            mean_val = mean_val.view(-1)
            std_val = std_val.view(-1)
        return mean_val + sigma * std_val


class MeanSigmaStd(torch.nn.Module):
    bvq_is_stat = True

    def __init__(self, sigma: float, stats_reduce_dim: Optional[int] = None,
                 std_dev_epsilon: float = DEFAULT_STD_DEV_EPSILON) -> None:
        super().__init__()
        from brevitas_amd.core.utils import StatelessBuffer
        self.impl = _MeanSigmaStdImpl(stats_reduce_dim, std_dev_epsilon)
        self.sigma = StatelessBuffer(torch.tensor(sigma))

    def forward(self, x: Tensor):
        return self.impl(x, self.sigma())


class MeanLearnedSigmaStd(TolerantLoad, torch.nn.Module):
    """MeanSigmaStd with a learned sigma (B/core/stats/stats_op.py:243-279).  The reference snapshot
    registers the parameter as `value` but reads `self.sigma` in forward and in its state-dict hook; the
    name used consistently here is `sigma` (with the reference's `learned_sigma` retro-compatibility key)."""
    bvq_is_stat = True
    bvq_float_checkpoint_ok = ('sigma',)

    def __init__(self, sigma: float, stats_output_shape, stats_reduce_dim: Optional[int] = None,
                 std_dev_epsilon: float = DEFAULT_STD_DEV_EPSILON) -> None:
        super().__init__()
        self.impl = _MeanSigmaStdImpl(stats_reduce_dim, std_dev_epsilon)
        if tuple(stats_output_shape) == ():
            self.sigma = torch.nn.Parameter(torch.tensor(sigma))
        else:
            self.sigma = torch.nn.Parameter(torch.full(stats_output_shape, sigma))

    def forward(self, x: Tensor):
        return self.impl(x, self.sigma.view(self.sigma.shape))

    def _load_from_state_dict(self, state_dict, prefix, *hook_args):
        legacy = prefix + 'learned_sigma'
        if legacy in state_dict:
            state_dict[prefix + 'sigma'] = state_dict.pop(legacy)
        super()._load_from_state_dict(state_dict, prefix, *hook_args)


class KLMinimizerThreshold(torch.nn.Module):
    """Clipping threshold that minimises the KL divergence between the histogram of x and its quantized version
    (drop-in for B/core/stats/stats_op.py:280-350, itself after MXNet's calibration).

    The two passes over x -- the abs-max and the `num_bins` histogram over [-absmax, absmax] -- are one streaming
    read each on the device (bvq_stats, bvq_histc; the histogram's range is read from device memory).  The search
    over the ~num_bins/2 candidate thresholds works on the 1001 counters on the host, like the reference's python
    loop (an offline calibration statistic: it synchronises, as the reference's `.int()` / indexing do)."""
    bvq_is_stat = True

    def __init__(self, signed, bit_width_impl, num_bins=1000 + 1, smoothing_eps=0.0001):
        super().__init__()
        self.num_bins = num_bins
        self.smoothing_eps = smoothing_eps
        self.signed = signed
        self.bit_width_impl = bit_width_impl
        self.absmax_impl = AbsMax()

    @staticmethod
    def smooth_normalize_distribution(p: Tensor, eps: float):
        """counts -> Categorical(logits = smoothed counts), None if every bin is empty (stats_op.py:295-305; the
        reference adds its correction term to every bin and hands the counts over as LOGITS: kept as it is)"""
        is_zeros = (p == 0).float()
        n_zeros = is_zeros.sum()
        n_nonzeros = torch.numel(p) - n_zeros
        if not n_nonzeros:
            return None
        eps1 = eps * n_zeros / n_nonzeros
        hist = p.float()
        hist = hist + (eps * is_zeros + (-eps1) * n_nonzeros)
        return torch.distributions.categorical.Categorical(logits=hist)

    def _histogram(self, x: Tensor, absmax: Tensor) -> Tensor:
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and self.num_bins <= 8192:
            flat = x.detach().reshape(-1)
            if flat.data_ptr() % 16 == 0:  # (bvq_histc takes 16-byte aligned data; a view into a buffer's middle below)
                return nat.histc(flat, absmax.detach(), self.num_bins).cpu()
        a = float(absmax)
        return torch.histc(x.detach().float().cpu(), bins=self.num_bins, min=-a, max=a).int()

    def forward(self, x: Tensor) -> Tensor:
        from brevitas_amd.function.ops import max_int
        absmax = self.absmax_impl(x)
        bit_width = self.bit_width_impl()
        bw = getattr(bit_width, 'bvq_host_value', None)
        nq = int(max_int(self.signed, False, torch.tensor(float(bw)) if bw is not None else bit_width.detach().cpu()))
        half, qhalf = self.num_bins // 2, nq // 2
        hist = self._histogram(x, absmax)                      # int32 [num_bins] on the host
        a = float(absmax)
        hist_edges = torch.linspace(-a, a, self.num_bins + 1)
        thresholds = torch.zeros(half + 1 - qhalf)
        divergence = torch.zeros_like(thresholds)
        total = hist.sum()
        csum = torch.cumsum(hist, 0)
        for i in range(qhalf, half + 1):
            start, stop = half - i, half + i + 1
            thresholds[i - qhalf] = hist_edges[stop]
            sliced = hist[start:stop]
            p = sliced.clone()
            p[0] += csum[start - 1] if start > 0 else 0         # outliers fold into the edge bins
            p[-1] += total - csum[stop - 1]
            nonzero = (sliced != 0).float()
            merged = torch.numel(p) // nq                       # histogram bins per quantized bin
            body = sliced[:nq * merged].reshape(nq, merged).sum(dim=1).float()
            body[-1] += sliced[nq * merged:].sum()
            # every non-empty histogram bin of a quantized bin gets that bin's mean count; the reference's slice
            # of the LAST quantized bin stops one element short of the end (stop = -1): kept
            q = torch.zeros(p.shape, dtype=torch.float32)
            norm = nonzero[:nq * merged].reshape(nq, merged).sum(dim=1)
            last_lo = (nq - 1) * merged
            norm[-1] = nonzero[last_lo:-1].sum()
            fill = torch.where(norm != 0, body / torch.where(norm != 0, norm, torch.ones_like(norm)), torch.zeros_like(body))
            q[:last_lo] = fill[:-1].repeat_interleave(merged)
            q[last_lo:-1] = fill[-1]
            q[sliced == 0] = 0.
            pd = self.smooth_normalize_distribution(p, self.smoothing_eps)
            qd = self.smooth_normalize_distribution(q, self.smoothing_eps)
            if qd is None:
                divergence[i - qhalf] = float('inf')
            else:
                divergence[i - qhalf] = torch.distributions.kl.kl_divergence(pd, qd)
        return thresholds[torch.argmin(divergence)].to(x.device)

