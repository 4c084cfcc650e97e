"""Pure-torch route for CPU tensors (SURVEY 8b "Errors").

The HIP engine takes device tensors only.  A layer built on the CPU, a CPU evaluation pass or a CPU unit
test must still run through the same modules -- the reference's own backend failure is a silent fall-back
to its Python ops (B/__init__.py:66-84) -- so every entry of `brevitas_amd._native` that the module
surface uses has an ATen-only counterpart here: the reference's op composition restated on torch ops,
nothing fused, nothing from oracle/ (that is test infrastructure).  `for_tensor(x)` picks the backend: a
ROCm tensor always gets the HIP library (and fails loudly if it is missing: importing
`brevitas_amd._native` without libbvq.so raises); only a CPU tensor gets this module.
"""
import torch

from . import _native as nat


def for_tensor(*tensors):
    """the backend module serving these tensors: `_native` (HIP) for device tensors, this module for CPU ones"""
    for t in tensors:
        if t is not None and t.is_cuda:
            return nat
    return _SELF


# ---- the 12 straight-through forwards (B/ops/autograd_ste_ops.py, B/function/ops.py) ------------------

def unary(op, x):
    if op == nat.OP_ROUND:
        return torch.round(x)
    if op == nat.OP_CEIL:
        return torch.ceil(x)
    if op == nat.OP_FLOOR:
        return torch.floor(x)
    if op == nat.OP_ROUND_TO_ZERO:  # B/function/ops.py:52
        return torch.sign(x) * torch.floor(torch.abs(x))
    if op == nat.OP_DPU_ROUND:      # B/function/ops.py:71
        return torch.where((x < 0.) & (x - torch.floor(x) == 0.5), torch.ceil(x), torch.round(x))
    if op == nat.OP_BINARY_SIGN:    # B/function/ops.py:31-34: +1 at 0
        return torch.ge(x, 0.0).type(x.dtype) + torch.lt(x, 0.0).type(x.dtype) * -1.0
    if op == nat.OP_TERNARY_SIGN:
        return torch.sign(x)
    if op == nat.OP_ABS:
        return torch.abs(x)
    raise nat.BvqError('brevitas_amd._aten.unary: unknown op %r' % (op,))


def scalar_clamp(x, min_val, max_val):
    if max_val is None:
        return torch.clamp_min(x, min_val)
    return torch.clamp(x, min_val, max_val)


def tensor_clamp(x, min_val, max_val, out=None):
    """where(x > max, max, x) then where(. < min, min, .)  (B/function/ops.py:98-100)"""
    y = torch.where(x > max_val, max_val.type_as(x), x)
    y = torch.where(y < min_val, min_val.type_as(y), y)
    if out is not None:
        out.copy_(y)
        return out
    return y


def tensor_clamp_bwd(grad_y, x, min_val, max_val):
    """autograd of the two torch.where w.r.t. x: the gradient passes where neither bound replaced the value"""
    hi = x > max_val
    y = torch.where(hi, max_val.type_as(x), x)
    lo = y < min_val
    return torch.where(hi | lo, torch.zeros_like(grad_y), grad_y)


def abs_binary_sign_grad_bwd(grad_y, x):
    return unary(nat.OP_BINARY_SIGN, x) * grad_y


def running_stats_update(running, stat, momentum, first_batch):
    """_RuntimeStats' in-place update (B/core/stats/stats_wrapper.py:61-66)"""
    if first_batch:
        running.mul_(stat)
    else:
        running.mul_(1 - momentum)
        running.add_(momentum * stat)
    return running


# ---- statistics (B/core/stats/stats_op.py): plain torch ops, autograd included --------------------

def _kth_rank_high(q, n):
    import math
    return int(math.floor(.01 * q * n + 0.5))  # k is 1-indexed: round away from zero (stats_op.py:56)


def _kth_rank_low(q, n):
    import math
    return int(math.ceil(.01 * q * n))  # stats_op.py:84


def _along(x, dim):
    if dim is None:
        return x.numel()
    assert len(x.size()) == 2, "Only 2-dim input is supported."
    return x.shape[dim]


def _kth(x, k, dim):
    return x.view(-1).kthvalue(k).values if dim is None else x.kthvalue(k, dim=dim).values


def abs_max(x, dim):
    return torch.max(torch.abs(x)) if dim is None else torch.max(torch.abs(x), dim=dim)[0]


def min_max(x, dim):
    """-> (max, min)"""
    if dim is None:
        return torch.max(x), torch.min(x)
    return torch.max(x, dim=dim)[0], torch.min(x, dim=dim)[0]


class _ShardedExtremum(torch.autograd.Function):
    """max |x| / max x / min x of a BATCH-SHARDED CPU tensor (brevitas_amd.distributed over gloo): the device route's
    protocol on torch ops -- all-reduce(MAX) of the float32 statistic forward; backward the summed gradient goes where
    the single-process run on the concatenated batch puts it (first attaining element of the lowest rank along a
    reduced dim, evenly over the ties of all shards for a whole-tensor reduction)."""

    @staticmethod
    def forward(ctx, x, dim, group, kind):
        import torch.distributed as dist
        key = torch.abs(x) if kind == 'abs' else (x if kind == 'max' else -x)
        rows = key.reshape(1, -1) if dim is None else key.movedim(dim, -1).reshape(-1, key.shape[dim])
        local = rows.max(dim=1).values.float()
        dist.all_reduce(local, op=dist.ReduceOp.MAX, group=group)
        stat_key = local.to(x.dtype)
        ctx.save_for_backward(x, stat_key)
        ctx.args = (dim, group, kind)
        stat = stat_key if kind != 'min' else -stat_key
        return stat.reshape(()) if dim is None else stat.reshape([s for i, s in enumerate(x.shape) if i != dim % x.dim()])

    @staticmethod
    def backward(ctx, gstat):
        from .distributed import sync_backward
        x, stat_key = ctx.saved_tensors
        dim, group, kind = ctx.args
        key = torch.abs(x) if kind == 'abs' else (x if kind == 'max' else -x)
        moved = key.reshape(1, -1) if dim is None else key.movedim(dim, -1)
        rows = moved.reshape(-1, moved.shape[-1])
        hit = rows == stat_key.reshape(-1, 1)
        ch = rows.shape[0]
        g32 = gstat.reshape(-1).float()
        if dim is None:
            info = hit.sum().reshape(1).to(torch.int64)
            gsum, _, total = sync_backward(g32, info, 1, group)
            drows = torch.where(hit, (gsum / total.to(gsum.dtype)).to(x.dtype), torch.zeros((), dtype=x.dtype))
        else:
            first = torch.where(hit.any(dim=1), hit.to(torch.int8).argmax(dim=1), torch.full((ch,), -1, dtype=torch.int64))
            gsum, info, _ = sync_backward(g32, first, ch, group, first_only=True)
            drows = torch.zeros_like(rows)
            own = info >= 0
            drows[own.nonzero().reshape(-1), info[own]] = gsum.to(x.dtype)[own]
        d = drows.reshape(moved.shape)
        d = d.reshape(x.shape) if dim is None else d.movedim(-1, dim)
        if kind == 'abs':
            d = d * torch.sgn(x)
        elif kind == 'min':
            pass  # d(min x)/dx = +1 at the arg-min: the key's sign and the statistic's cancel
        return d, None, None, None


def sharded_abs_max(x, dim, group):
    return _ShardedExtremum.apply(x, dim, group, 'abs')


def sharded_min_max(x, dim, group):
    """-> (max, min)"""
    return _ShardedExtremum.apply(x, dim, group, 'max'), _ShardedExtremum.apply(x, dim, group, 'min')


def abs_percentile(x, q, dim):
    return _kth(x.abs(), _kth_rank_high(q, _along(x, dim)), dim)


def low_percentile(x, q, dim):
    return _kth(x, _kth_rank_low(q, _along(x, dim)), dim)


def high_percentile(x, q, dim):
    return _kth(x, _kth_rank_high(q, _along(x, dim)), dim)


def abs_mean_var(x, dim):
    """-> (mean |x|, unbiased var |x|)"""
    a = torch.abs(x)
    if dim is None:
        return torch.mean(a), torch.var(a)
    return torch.mean(a, dim=dim), torch.var(a, dim=dim)


import sys  # noqa: E402

_SELF = sys.modules[__name__]
