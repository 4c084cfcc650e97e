"""Autograd entry points of the fused quantizer kernels.

FakeQuantFn        x, scale, zero_point  ->  y                  (IntQuant.forward in one kernel)
StatsFakeQuantFn   x                     ->  y, scale, stat     (AbsMax -> scale -> IntQuant in
                                                                 three kernels, backward with the
                                                                 arg-max deposit folded into dx)

Both reproduce the reference's type promotion: the compute dtype is torch.result_type(x, scale),
every intermediate is rounded to it, the output has it, and the gradient w.r.t. x comes back in
x's dtype.  Layouts the kernels do not cover make `plan()` return None and the caller falls back
to the op-by-op composition (same results, more passes).
"""
from typing import NamedTuple, Optional

import torch
from torch import Tensor
from torch.autograd import Function

import brevitas_amd.config as config
from brevitas_amd import _native as nat

_FLOATS = (torch.float32, torch.bfloat16, torch.float16)


class Plan(NamedTuple):
    outer: int
    channels: int
    inner: int
    scale_pc: bool
    zp_pc: bool
    ct: torch.dtype
    nhwc: bool = False  # x is a dense channels_last tensor taken in memory order: [N*H*W, C] rows, channel last


def is_nhwc(x: Tensor, channel_dim) -> bool:
    """per-channel (dim 1) quantization of a dense channels_last tensor: its memory IS [N*H*W, C], which the
    column-mapped kernels take as it lies (no NCHW copy and back)"""
    return channel_dim == 1 and x.dim() == 4 and not x.is_contiguous() \
        and x.is_contiguous(memory_format=torch.channels_last)


def _channel_dim(x: Tensor, t: Tensor):
    """None: one element (per-tensor); int: the single broadcast dim; -1: not a per-channel layout"""
    if t.numel() == 1:
        return None
    if t.dim() > x.dim():
        return -1
    shape = (1,) * (x.dim() - t.dim()) + tuple(t.shape)
    nz = [i for i, s in enumerate(shape) if s != 1]
    if len(nz) != 1 or shape[nz[0]] != x.shape[nz[0]]:
        return -1
    return nz[0]


def plan(x: Tensor, scale: Tensor, zp: Tensor) -> Optional[Plan]:
    """how (and whether) the fused kernels can run IntQuant on these operands"""
    if not (x.is_cuda and scale.is_cuda and zp.is_cuda) or x.dim() == 0 or x.numel() == 0:
        return None
    if x.dtype not in _FLOATS or scale.dtype not in _FLOATS or zp.dtype not in _FLOATS:
        return None
    if scale.dim() > x.dim() or zp.dim() > x.dim():
        return None
    sd, zd = _channel_dim(x, scale), _channel_dim(x, zp)
    if sd == -1 or zd == -1 or (sd is not None and zd is not None and sd != zd):
        return None
    ct = torch.result_type(x, scale)
    # a 0-dim zero-point never promotes a dimensioned tensor; a dimensioned one must not either
    if zp.dim() > 0 and torch.promote_types(ct, zp.dtype) != ct:
        return None
    if not (ct == x.dtype or ct == torch.float32):
        return None
    cd = sd if sd is not None else zd
    if cd is None:
        return Plan(1, 1, x.numel(), False, False, ct)
    if is_nhwc(x, cd):
        return Plan(x.numel() // x.shape[1], x.shape[1], 1, sd is not None, zd is not None, ct, True)
    outer = 1
    for s in x.shape[:cd]:
        outer *= s
    inner = 1
    for s in x.shape[cd + 1:]:
        inner *= s
    return Plan(outer, x.shape[cd], inner, sd is not None, zd is not None, ct)


def scalar_mode():
    return nat.SCALAR_CAST if config.SCALAR_OPERAND_MODE == 'device' else nat.SCALAR_OPMATH


def make_desc(p: Plan, x: Tensor, scale: Tensor, zp: Tensor, qmin, qmax, round_mode, clamp_ste, out_kind,
              pre_op=nat.PRE_NONE):
    return nat.QuantDesc(p.outer, p.channels, p.inner, nat.dtype_code(x.dtype), nat.dtype_code(p.ct),
                         nat.dtype_code(scale.dtype), nat.dtype_code(zp.dtype), int(p.scale_pc), int(p.zp_pc),
                         qmin, qmax, round_mode, scalar_mode(), int(clamp_ste), out_kind, pre_op)


def _reduce_like(sums: Tensor, like: Tensor) -> Tensor:
    """per-channel float32 sums -> gradient shaped and typed like `like`"""
    if like.numel() == 1 and sums.numel() > 1:
        sums = sums.sum()
    return sums.reshape(like.shape).to(like.dtype)


def _memory_order(x: Tensor, channels: int, nhwc: bool = False):
    """-> (contiguous tensor holding x's elements, permutation that maps a same-ordered result back to x's
    logical layout or None).  A per-tensor quantizer does not care about element order, so a dense
    channels_last tensor (the layout MIOpen prefers) is taken as it lies in memory instead of being copied to
    NCHW and back; every other case is `x.contiguous()` like the reference's own reshape."""
    if nhwc:
        return x.permute(0, 2, 3, 1), (0, 3, 1, 2)
    if x.is_contiguous() or channels != 1:
        return x.contiguous(), None
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return x.permute(0, 2, 3, 1), (0, 3, 1, 2)
    if x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d):
        return x.permute(0, 2, 3, 4, 1), (0, 4, 1, 2, 3)
    return x.contiguous(), None


def _like_memory_order(t: Tensor, back) -> Tensor:
    """bring an incoming gradient into the element order the forward used"""
    if back is None:
        return t.contiguous()
    inv = [0] * len(back)
    for i, b in enumerate(back):
        inv[b] = i
    return t.permute(inv).contiguous()


def _restore(t: Tensor, back) -> Tensor:
    return t if back is None else t.permute(back)


class FakeQuantFn(Function):
    """IntQuant.forward / IntQuant.to_int on the fused kernel (B/core/quant/int_base.py:63-97)"""

    @staticmethod
    def forward(ctx, x, scale, zp, p, qmin, qmax, round_mode, clamp_ste, out_kind, pre_op=nat.PRE_NONE):
        xc, back = _memory_order(x, p.channels, p.nhwc)
        sc = scale.reshape(-1).contiguous()
        zc = zp.reshape(-1).contiguous()
        desc = make_desc(p, xc, sc, zc, qmin, qmax, round_mode, clamp_ste, out_kind, pre_op)
        y = nat.fakequant_fwd(desc, xc, sc, zc)
        ctx.desc = desc
        ctx.back = back
        ctx.save_for_backward(xc, scale, zp)
        if back is not None:
            y = y.permute(back)
        if out_kind == nat.OUT_INT:
            ctx.mark_non_differentiable(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, scale, zp = ctx.saved_tensors
        desc = ctx.desc
        need_ds, need_dz = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
        gy = _like_memory_order(gy.to(ct), ctx.back)
        dx, ds, dz = nat.fakequant_bwd(desc, gy, xc, scale.reshape(-1).contiguous(), zp.reshape(-1).contiguous(),
                                       need_ds, need_dz)
        if ctx.back is not None:
            dx = dx.permute(ctx.back)
        ds = _reduce_like(ds, scale) if need_ds else None
        dz = _reduce_like(dz, zp) if need_dz else None
        return dx, ds, dz, None, None, None, None, None, None, None


class FakeQuantBoundsFn(Function):
    """IntQuant.forward whose integer range is a pair of 0-dim TENSORS (a learned bit width: min_int / max_int of the
    bit-width tensor, B/core/bit_width/parameter.py:23-100, B/function/ops.py:132-191): the fused kernels read the
    bounds from device memory, and the backward returns, next to dx and dscale, the gradient tensor_clamp's two
    torch.where send to the bounds (plain TensorClamp only; a straight-through clamp gives the bounds none)."""

    @staticmethod
    def forward(ctx, x, scale, zp, qmin_t, qmax_t, p, round_mode, clamp_ste, pre_op):
        xc, back = _memory_order(x, p.channels, p.nhwc)
        sc = scale.reshape(-1).contiguous()
        zc = zp.reshape(-1).contiguous()
        bounds = torch.stack([qmin_t.detach().reshape(()), qmax_t.detach().reshape(())]).float()
        desc = make_desc(p, xc, sc, zc, 0.0, 0.0, round_mode, clamp_ste, nat.OUT_DEQUANT, pre_op)
        y = nat.fakequant_fwd_bounds(desc, xc, sc, zc, bounds)
        ctx.desc, ctx.back = desc, back
        ctx.save_for_backward(xc, scale, zp, bounds, qmin_t, qmax_t)
        return y if back is None else y.permute(back)

    @staticmethod
    def backward(ctx, gy):
        xc, scale, zp, bounds, qmin_t, qmax_t = ctx.saved_tensors
        desc = ctx.desc
        ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
        gy = _like_memory_order(gy.to(ct), ctx.back)
        need_db = (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) and not desc.clamp_ste
        dx, ds, db = nat.fakequant_bwd_bounds(desc, gy, xc, scale.reshape(-1).contiguous(), zp.reshape(-1).contiguous(),
                                              bounds, need_db)
        dx = _restore(dx, ctx.back) if ctx.needs_input_grad[0] else None
        ds = _reduce_like(ds, scale) if ctx.needs_input_grad[1] else None
        dmin = dmax = None
        if need_db:
            # the reference sums where()'s masked gradient in the clamp's dtype, then casts to the bound's
            sums = db.sum(dim=1).to(ct)
            dmin = sums[0].to(qmin_t.dtype).reshape(qmin_t.shape) if ctx.needs_input_grad[3] else None
            dmax = sums[1].to(qmax_t.dtype).reshape(qmax_t.shape) if ctx.needs_input_grad[4] else None
        return dx, ds, None, dmin, dmax, None, None, None, None


class LearnedScaleFakeQuantFn(Function):
    """scale = |clamp_min(value, min_val)| / int_threshold  ->  IntQuant with that scale and a zero zero-point:
    the steady state of the learned-scale activation quantizers (ParameterScaling, and
    ParameterFromRuntimeStatsScaling after its collection phase -- the default Int8ActPerTensorFloat;
    B/core/scaling/standalone.py:75-152,155-298, B/core/quant/int.py:157-163).

    Forward: one launch for the scale (instead of clamp, |.|, division), the quantizer kernel.  Backward: the
    quantizer's backward kernel, its scale-gradient sums, and -- in the LAST reduction launch -- the chain
    dscale -> d(threshold) -> d(value) with torch's rounding points (instead of a cast, a division, a
    sign-multiply and their launches).  Returns (y, scale)."""

    @staticmethod
    def forward(ctx, x, value, p, min_val, thr_div, scale_dtype, qmin, qmax, round_mode, clamp_ste, pre_op):
        ctx.set_materialize_grads(False)
        xc, back = _memory_order(x, p.channels, p.nhwc)
        sc = nat.learned_scale(value, min_val, thr_div, scale_dtype)
        zp = _zero_zero_point(x.device)
        desc = make_desc(p, xc, sc, zp.reshape(-1), qmin, qmax, round_mode, clamp_ste, nat.OUT_DEQUANT, pre_op)
        y = nat.fakequant_fwd(desc, xc, sc, zp.reshape(-1))
        ctx.desc, ctx.back, ctx.args = desc, back, (min_val, thr_div)
        ctx.save_for_backward(xc, sc, zp, value)
        if back is not None:
            y = y.permute(back)
        return y, sc.view(value.shape)

    @staticmethod
    def backward(ctx, gy, gscale):
        xc, sc, zp, value = ctx.saved_tensors
        desc = ctx.desc
        min_val, thr_div = ctx.args
        ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
        if gy is None:  # only `scale` was used downstream: its own small autograd chain, no tensor pass
            if gscale is None:
                return (None,) * 11
            v = value.detach()
            if min_val:
                v = torch.clamp_min(v, min_val)
            sign = torch.ge(v, 0.0).to(v.dtype) - torch.lt(v, 0.0).to(v.dtype)
            return None, (sign * (gscale.reshape(value.shape) / thr_div).to(v.dtype)), *(None,) * 9
        gy = _like_memory_order(gy.to(ct), ctx.back)
        gs = gscale.reshape(-1).contiguous().to(sc.dtype) if gscale is not None else None
        dx, _, dvalue = nat.fakequant_bwd_learned(desc, gy, xc, sc, zp.reshape(-1), value, min_val, thr_div, gs)
        if not ctx.needs_input_grad[0]:
            dx = None
        elif ctx.back is not None:
            dx = dx.permute(ctx.back)
        return dx, dvalue.view(value.shape), None, None, None, None, None, None, None, None, None


class VariantFn(Function):
    """BinaryQuant / ClampedBinaryQuant / TernaryQuant / DecoupledIntQuant / TruncIntQuant on their fused kernels
    (include/bvq.h, bvq_variant_fwd / bvq_variant_bwd): one read + one write forward, two reads + one write backward,
    the scale gradients on the same reads.  `opts`: kind plus the kind's parameters (variant_plan below)."""

    @staticmethod
    def forward(ctx, x, scale, pre_scale, zp, pre_zp, p, opts):
        xc = x.contiguous()
        sc = scale.reshape(-1).contiguous()
        ps = pre_scale.reshape(-1).contiguous() if pre_scale is not None else None
        zc = zp.reshape(-1).contiguous() if zp is not None else None
        pz = pre_zp.reshape(-1).contiguous() if pre_zp is not None else None
        zdt = (zc if zc is not None else pz if pz is not None else _zero_zero_point(x.device)).dtype
        desc = nat.VariantDesc(p.outer, p.channels, p.inner, opts['kind'], nat.dtype_code(x.dtype),
                               nat.dtype_code(opts['ct']), nat.dtype_code(sc.dtype), nat.dtype_code(zdt), int(p.scale_pc),
                               opts.get('round_mode', nat.ROUND), int(opts.get('clamp_ste', False)), scalar_mode(),
                               opts.get('qmin', 0.0), opts.get('qmax', 0.0), opts.get('threshold', 0.0),
                               opts.get('trunc_scale', 1.0))
        y = nat.variant_fwd(desc, xc, sc, ps, zc, pz)
        ctx.desc = desc
        ctx.save_for_backward(xc, sc, ps, zc, pz)
        ctx.shapes = (scale.shape, scale.dtype, None if pre_scale is None else (pre_scale.shape, pre_scale.dtype))
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, sc, ps, zc, pz = ctx.saved_tensors
        desc = ctx.desc
        ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
        need_ds, need_dp = ctx.needs_input_grad[1], ctx.needs_input_grad[2] and ps is not None
        dx, ds, dp = nat.variant_bwd(desc, gy.to(ct).contiguous(), xc, sc, ps, zc, pz, need_ds, need_dp)
        sshape, sdt, pre = ctx.shapes
        ds = ds.reshape(sshape).to(sdt) if need_ds else None
        dp = dp.reshape(pre[0]).to(pre[1]) if need_dp else None
        return (dx if ctx.needs_input_grad[0] else None), ds, dp, None, None, None, None


def _dimensioned_wider_scalar(x: Tensor, t: Optional[Tensor]) -> bool:
    """a ONE-element operand that still has dimensions (shape (1,), (1,1,1,1)) and a dtype wider than x's: torch's type
    promotion treats it as a tensor, not as a scalar -- every op with it computes and returns in ITS dtype -- while the
    variant kernels' one-element operands follow the 0-dim rule (rounded to x's dtype, bvq_scalar_mode).  Such operands
    take the op-by-op route."""
    return t is not None and t.numel() == 1 and t.dim() > 0 and torch.promote_types(x.dtype, t.dtype) != x.dtype


def variant_plan(x: Tensor, scale: Tensor, *others: Optional[Tensor]) -> Optional[Plan]:
    """layout plan for a variant kernel: x against `scale`; every other scale-like operand must share scale's shape
    and dtype, every zero-point must be a single element that needs no gradient"""
    if not config.FUSED_PATHS or torch._C._get_tracing_state() is not None:
        return None
    if _dimensioned_wider_scalar(x, scale):
        return None
    p = plan(x, scale, _zero_zero_point(x.device) if x.is_cuda else scale)
    if p is None or p.nhwc or not x.is_contiguous():
        return None
    for t in others:
        if t is not None and (t.shape != scale.shape or t.dtype != scale.dtype or not t.is_cuda):
            return None
    return p


def scalar_zero_point_ok(*zps: Optional[Tensor], x: Optional[Tensor] = None) -> bool:
    """x given: also refuse a dimensioned one-element zero-point that would promote x (see _dimensioned_wider_scalar)"""
    return all(z is None or (z.numel() == 1 and z.is_cuda and z.dtype in _FLOATS and not z.requires_grad
                             and not (x is not None and _dimensioned_wider_scalar(x, z))) for z in zps)


class StatsPlan(NamedTuple):
    outer: int
    channels: int
    inner: int
    scaling_shape: tuple
    min_val: float
    int_threshold: float  # host value of int_scaling_impl(bit_width)
    nhwc: bool = False    # see Plan.nhwc


_ZERO_ZP = {}


def _zero_zero_point(device):
    """the 0-dim float32 zero the ZeroZeroPoint module returns, one per device (no fill kernel per call)"""
    z = _ZERO_ZP.get(device)
    if z is None:
        z = torch.zeros((), dtype=torch.float32, device=device)
        _ZERO_ZP[device] = z
    return z


_AS_DTYPE = {}


def _as_dtype_value(v: float, dtype: torch.dtype) -> float:
    """v after torch converts it to `dtype` (what a 0-dim float32 operand becomes next to a dimensioned
    tensor of that dtype on the device)"""
    key = (v, dtype)
    r = _AS_DTYPE.get(key)
    if r is None:
        r = _AS_DTYPE[key] = float(torch.tensor(v, dtype=torch.float32).to(dtype))
    return r


def stats_scale(flat: Tensor, int_threshold: Tensor, sp, group, pre_op, running=None):
    """AbsMax statistic of the [outer, channels, inner] tensor `flat` and the scale derived from it:
    -> (stat [channels] in x's dtype, scale shaped sp.scaling_shape).  group: the tensor is one batch shard.
    running: (buffer, momentum, first_batch) of a _RuntimeStats to fold the statistic into in the same launch
    (sharded: the statistic -> scale launch behind the all-reduce)."""
    if group is None:
        # statistic -> clamp_min -> / int_threshold in the reduction's own finishing launch.
        # torch's promotion: a dimensioned threshold keeps its dtype (the 0-dim float32
        # int_threshold is converted to it), a 0-dim one is promoted with float32.
        if len(sp.scaling_shape) > 0:
            scale_dtype, thr_div = flat.dtype, _as_dtype_value(sp.int_threshold, flat.dtype)
        else:
            scale_dtype = torch.promote_types(flat.dtype, int_threshold.dtype)
            thr_div = sp.int_threshold
        if running is not None:
            buf, momentum, first = running
            stat, scale = nat.absmax_scale(flat, sp.outer, sp.channels, sp.inner, sp.min_val, thr_div, scale_dtype,
                                           pre_op, running=buf, momentum=momentum, first_batch=first)
        else:
            stat, scale = nat.absmax_scale(flat, sp.outer, sp.channels, sp.inner, sp.min_val, thr_div, scale_dtype,
                                           pre_op)
        return stat, scale.view(sp.scaling_shape)
    # batch-sharded tensor: the statistic of the whole batch is the max over the shards
    from brevitas_amd.distributed import sync_stat_max
    stat32 = sync_stat_max(nat.stats(nat.STAT_ABSMAX, flat, sp.outer, sp.channels, sp.inner, out_f32=True,
                                     pre_op=pre_op), group)
    # statistic -> clamp_min -> / int_threshold with the promotion rules of the unsharded route, one launch
    if len(sp.scaling_shape) > 0:
        scale_dtype, thr_div = flat.dtype, _as_dtype_value(sp.int_threshold, flat.dtype)
    else:
        scale_dtype = torch.promote_types(flat.dtype, int_threshold.dtype)
        thr_div = sp.int_threshold
    if running is not None:
        buf, momentum, first = running
        stat, scale = nat.scale_from_stat(stat32, flat.dtype, sp.min_val, thr_div, scale_dtype, buf, momentum, first)
    else:
        stat, scale = nat.scale_from_stat(stat32, flat.dtype, sp.min_val, thr_div, scale_dtype)
    return stat, scale.view(sp.scaling_shape)


class StatsFakeQuantFn(Function):
    """AbsMax statistic -> clamp_min(min_val) -> / int_threshold -> IntQuant, zero zero-point.

    The resolved graphs of Int8WeightPerChannelFloat (StatsFromParameterScaling) and of stats-scaled
    activations (RuntimeStatsScaling in training): SURVEY 8a.  Forward: read x (reduce), read x, write
    y.  Backward: read g, read x, write dx; the statistic's gradient is deposited on the arg-max
    element(s) of dx in place instead of materialising the dense gradient torch.max would return.
    """

    @staticmethod
    def forward(ctx, x, int_threshold, sp, qmin, qmax, round_mode, clamp_ste, group=None, pre_op=nat.PRE_NONE,
                runtime=None):
        ctx.set_materialize_grads(False)  # an unused `scale` output must not cost a zero-fill + add
        ctx.pre_op = pre_op
        xc, back = _memory_order(x, sp.channels, sp.nhwc)
        ctx.back = back
        flat = xc.reshape(-1)
        zp = _zero_zero_point(x.device)
        fused = None
        if group is None and config.FUSED_PATHS and x.dtype in _FLOATS:
            # one launch: the channel stays in registers between the reduction and the quantization
            # (x is read once).  Covers channels that fit a team's registers; None otherwise.
            if len(sp.scaling_shape) > 0:
                scale_dtype, thr_div = x.dtype, _as_dtype_value(sp.int_threshold, x.dtype)
            else:
                scale_dtype = torch.promote_types(x.dtype, int_threshold.dtype)
                thr_div = sp.int_threshold
            code = nat.dtype_code(x.dtype)
            desc = nat.QuantDesc(sp.outer, sp.channels, sp.inner, code, code, nat.dtype_code(scale_dtype),
                                 nat.dtype_code(zp.dtype), int(sp.channels > 1), 0, qmin, qmax, round_mode,
                                 scalar_mode(), int(clamp_ste), nat.OUT_DEQUANT, pre_op)
            fused = nat.stats_fakequant_fwd(desc, xc, sp.min_val, thr_div, scale_dtype)
        if fused is not None:
            stat, scale, y = fused
            scale = scale.view(sp.scaling_shape)
        else:
            # _RuntimeStats' running average rides on the statistic's finishing launch when its buffer is a plain
            # [channels] device tensor (the module is told through `bvq_running_folded`)
            running = None
            if runtime is not None:
                buf = runtime.running_stats
                if buf.is_cuda and buf.is_contiguous() and buf.numel() == sp.channels and buf.dtype in _FLOATS:
                    running = (buf, runtime.momentum, runtime.first_batch)
                    runtime.bvq_running_folded = True
            stat, scale = stats_scale(flat, int_threshold, sp, group, pre_op, running)
            p = Plan(sp.outer, sp.channels, sp.inner, sp.channels > 1, False, torch.result_type(x, scale), sp.nhwc)
            if not (p.ct == x.dtype or p.ct == torch.float32):
                raise nat.BvqError('StatsFakeQuantFn: unsupported operand layout (caller must pre-check)')
            sc = scale.reshape(-1).contiguous()
            desc = make_desc(p, xc, sc, zp.reshape(-1), qmin, qmax, round_mode, clamp_ste, nat.OUT_DEQUANT, pre_op)
            y = nat.fakequant_fwd(desc, xc, sc, zp.reshape(-1))
        ctx.desc = desc
        ctx.sp = sp
        ctx.group = group
        ctx.save_for_backward(xc, scale, zp, stat, int_threshold)
        stat_out = stat.view(sp.scaling_shape)
        ctx.mark_non_differentiable(stat_out)
        if back is not None:
            y = y.permute(back)
        return y, scale, stat_out

    @staticmethod
    def backward(ctx, gy, gscale, _gstat):
        xc, scale, zp, stat, int_threshold = ctx.saved_tensors
        dx = stats_backward(xc, scale, zp, stat, int_threshold, ctx.desc, ctx.sp, ctx.group, ctx.pre_op, ctx.back, gy,
                            gscale)
        return dx, None, None, None, None, None, None, None, None, None


class StatsGraphFakeQuantFn(Function):
    """AbsMax statistic -> ANY scale-shaped map (`post`: a _StatsScaling with affine rescaling, a power-of-two
    restriction, ...) -> / int_threshold -> IntQuant, zero zero-point (B/core/scaling/runtime.py:19-72,
    B/core/quant/int.py:155-163).

    The tensor-sized work stays on the fused kernels -- statistic (1 read), quantizer (1 read + 1 write), backward
    (2 reads + 1 write, with the scale-gradient sums and the arg-max positions on the same reads) -- and only the map
    itself runs as torch ops on `channels`-sized tensors, recorded here and differentiated by autograd on those small
    tensors: dscale -> (d statistic, d affine_weight, d affine_bias); the statistic's gradient is then deposited on the
    arg-max elements in place.  The op-by-op route it replaces sends the statistic's gradient through AbsMax's own
    backward (a second pass over x writing a dense gradient) and adds the two x-sized gradients: 11 tensor passes
    instead of 6.  `params`: post's parameters, passed so that autograd sees them as inputs."""

    @staticmethod
    def forward(ctx, x, int_threshold, sp, qmin, qmax, round_mode, clamp_ste, pre_op, post, *params):
        ctx.set_materialize_grads(False)
        xc, back = _memory_order(x, sp.channels, sp.nhwc)
        flat = xc.reshape(-1)
        stat = nat.stats(nat.STAT_ABSMAX, flat, sp.outer, sp.channels, sp.inner, pre_op=pre_op).view(sp.scaling_shape)
        with torch.enable_grad():
            s_leaf = stat.detach().requires_grad_(True)
            scale_g = post(s_leaf) / int_threshold
        scale = scale_g.detach()
        zp = _zero_zero_point(x.device)
        p = plan(xc if back is None else x, scale, zp)
        if p is None or p.zp_pc or p.nhwc != sp.nhwc:
            raise nat.BvqError('StatsGraphFakeQuantFn: operand layout not covered by the quantizer kernels')
        sc = scale.reshape(-1).contiguous()
        desc = make_desc(p, xc, sc, zp.reshape(-1), qmin, qmax, round_mode, clamp_ste, nat.OUT_DEQUANT, pre_op)
        y = nat.fakequant_fwd(desc, xc, sc, zp.reshape(-1))
        ctx.desc, ctx.sp, ctx.back, ctx.pre_op = desc, sp, back, pre_op
        ctx.graph = (s_leaf, scale_g, params)
        ctx.save_for_backward(xc, scale, zp, stat)
        ctx.mark_non_differentiable(stat)
        if back is not None:
            y = y.permute(back)
        return y, scale, stat

    @staticmethod
    def backward(ctx, gy, gscale, _gstat):
        xc, scale, zp, stat = ctx.saved_tensors
        desc, sp = ctx.desc, ctx.sp
        s_leaf, scale_g, params = ctx.graph
        none = (None,) * 9
        ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
        if gy is None and gscale is None:
            return none + (None,) * len(params)
        if gy is None:
            gy = torch.zeros(xc.shape, dtype=ct, device=xc.device)
        else:
            gy = _like_memory_order(gy.to(ct), ctx.back)
        stat_x = stat.reshape(-1).to(xc.dtype).contiguous()
        dx, ds, _, ties = nat.fakequant_bwd(desc, gy, xc, scale.reshape(-1).contiguous(), zp.reshape(-1), True, False,
                                            tie_stat=stat_x)
        dscale = _reduce_like(ds, scale)
        if gscale is not None:
            dscale = dscale + gscale
        grads = torch.autograd.grad(scale_g, (s_leaf,) + tuple(params), dscale, retain_graph=True, allow_unused=True)
        dstat = grads[0]
        if dstat is not None:
            nat.stat_tie_apply(nat.MATCH_ABS, xc.reshape(-1), stat_x, dstat.reshape(-1).to(xc.dtype).contiguous(), ties,
                               dx.reshape(-1), sp.outer, sp.channels, sp.inner, mode_add=True, pre_op=ctx.pre_op)
        return (_restore(dx, ctx.back),) + none[1:] + tuple(grads[1:])


def _list_views(params, sp):
    """(flat tensors, outers, inners) of a shared quantizer's parameters as the list kernel takes them"""
    flats = [t.detach().reshape(-1) for t in params]
    return flats, [1] * len(params), [t.numel() // sp.channels for t in params]


def list_stats_supported(params, sp) -> bool:
    """the one-launch statistic over a parameter list applies (nat.absmax_scale_list would not return None)"""
    import ctypes
    if not config.FUSED_PATHS or params[0].dtype not in _FLOATS or sp.outer != 1 or sp.nhwc:
        return False
    flats, outers, inners = _list_views(params, sp)
    n = len(flats)
    st = nat.stream_ptr(flats[0].device)
    if sp.channels > 1 and nat.arrival_buffer(flats[0].device, st, max(2 * sp.channels, 18)) is None:
        return False
    return bool(nat.lib.bvq_absmax_list_supported(
        nat.dtype_code(flats[0].dtype), n, (ctypes.c_void_p * n)(*[f.data_ptr() for f in flats]),
        (ctypes.c_int64 * n)(*outers), sp.channels, (ctypes.c_int64 * n)(*inners)))


class ListStatsFakeQuantFn(Function):
    """StatsFakeQuantFn for a weight quantizer SHARED by several layers: the statistic is AbsMax over the concatenation
    of all tracked parameters' views (B/core/stats/stats_wrapper.py:83-114, _ParameterListStats with more than one
    parameter; B/core/scaling/standalone.py StatsFromParameterScaling), the tensor quantized is one of them.

    Forward, two launches: the statistic of the whole list + scale (nat.absmax_scale_list: the concatenation is never
    built), the quantizer kernel on x.  Backward: the quantizer's backward on x (dx, the scale-gradient sums and x's own
    arg-max positions on the same reads); the other parameters are scanned for the positions attaining the statistic,
    and the statistic's gradient goes where torch's backward of max over the concatenation puts it -- per channel on
    the FIRST position in concatenation order (extra parameters are concatenated in front: the highest list index
    that attains it), for a whole-tensor statistic evenly over all ties of all parameters -- into dx for x, into zero
    tensors for the others.  The reference reads
    and writes every parameter for torch.cat, reduces the copy, and in backward writes a dense gradient of the
    concatenation and slices it."""

    @staticmethod
    def forward(ctx, x, int_threshold, sp, qmin, qmax, round_mode, clamp_ste, index, *params):
        ctx.set_materialize_grads(False)
        flats, outers, inners = _list_views(params, sp)
        if len(sp.scaling_shape) > 0:
            scale_dtype, thr_div = x.dtype, _as_dtype_value(sp.int_threshold, x.dtype)
        else:
            scale_dtype = torch.promote_types(x.dtype, int_threshold.dtype)
            thr_div = sp.int_threshold
        out = nat.absmax_scale_list(flats, outers, sp.channels, inners, sp.min_val, thr_div, scale_dtype)
        if out is None:
            raise nat.BvqError('ListStatsFakeQuantFn: list not covered (caller must pre-check list_stats_supported)')
        stat, scale = out
        scale = scale.view(sp.scaling_shape)
        zp = _zero_zero_point(x.device)
        xc = x.detach()
        p = Plan(sp.outer, sp.channels, sp.inner, sp.channels > 1, False, torch.result_type(x, scale), False)
        if not (p.ct == x.dtype or p.ct == torch.float32):
            raise nat.BvqError('ListStatsFakeQuantFn: unsupported operand layout')
        sc = scale.reshape(-1).contiguous()
        desc = make_desc(p, xc, sc, zp.reshape(-1), qmin, qmax, round_mode, clamp_ste, nat.OUT_DEQUANT, nat.PRE_NONE)
        y = nat.fakequant_fwd(desc, xc, sc, zp.reshape(-1))
        ctx.desc, ctx.sp, ctx.index, ctx.inners = desc, sp, index, inners
        ctx.save_for_backward(scale, zp, stat, int_threshold, *params)
        stat_out = stat.view(sp.scaling_shape)
        ctx.mark_non_differentiable(stat_out)
        return y, scale, stat_out

    @staticmethod
    def backward(ctx, gy, gscale, _gstat):
        scale, zp, stat, int_threshold = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        desc, sp, index = ctx.desc, ctx.sp, ctx.index
        none = (None,) * 8
        if gy is None and gscale is None:
            return none + (None,) * len(params)
        x = params[index].detach()
        ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
        gy = torch.zeros(x.shape, dtype=ct, device=x.device) if gy is None else gy.to(ct).contiguous()
        stat_x = stat.reshape(-1).to(x.dtype).contiguous()
        dx, ds, _, own = nat.fakequant_bwd(desc, gy, x, scale.reshape(-1).contiguous(), zp.reshape(-1), True, False,
                                           tie_stat=stat_x)
        ds = _reduce_like(ds, scale)
        if gscale is not None:
            ds = ds + gscale
        # scale = thr / int_threshold  ->  dthr = dscale / int_threshold ; clamp_min_ste passes it on
        dstat = (ds / int_threshold).to(stat.dtype).reshape(-1).contiguous()
        ch = sp.channels
        # the other parameters: one read each, which also writes the zeros their non-attaining elements receive
        # (0 * sgn(x): the reference's gradient of |x| there, signed zeros included)
        outs = [dx if i == index else torch.empty_like(t, memory_format=torch.contiguous_format)
                for i, t in enumerate(params)]
        infos = [own if i == index else
                 nat.stat_tie_scan(nat.MATCH_ABS, t.detach().reshape(-1), stat_x, 1, ch, ctx.inners[i],
                                   dx_zero_fill=outs[i].reshape(-1))
                 for i, t in enumerate(params)]
        total = None
        if ch > 1:
            # the deposit of channel c belongs to the first tensor IN CONCATENATION ORDER that attains the statistic
            # there; every extra parameter is concatenated in FRONT of what came before
            # (B/core/stats/view_wrapper.py:49-50), so that order is the list's, reversed
            order = list(range(len(infos)))[::-1]
            first = torch.stack([infos[i][:ch] for i in order])                # [n, channels], -1: not attained
            has = first >= 0
            owner = torch.where(has.any(0), has.to(torch.int8).argmax(0), torch.full_like(first[0], -1))
            for r, i in enumerate(order):
                infos[i][:ch] = torch.where(owner == r, infos[i][:ch], torch.full_like(infos[i][:ch], -1))
        else:
            total = torch.stack([w[0] for w in infos]).sum().reshape(1)        # ties of the whole list share evenly
        for i, t in enumerate(params):
            nat.stat_tie_apply(nat.MATCH_ABS, t.detach().reshape(-1), stat_x, dstat, infos[i], outs[i].reshape(-1), 1, ch,
                               ctx.inners[i], mode_add=(i == index), total_ties=total)
        return (dx,) + none[1:] + tuple(g if i != index else None for i, g in enumerate(outs))


def stats_backward(xc, scale, zp, stat, int_threshold, desc, sp, group, pre_op, back, gy, gscale):
    """backward of StatsFakeQuantFn (also the fallback of the C++ node, brevitas_amd/csrc/bvq_autograd.cpp) -> dx or None;
    sp: anything with outer / channels / inner / int_threshold"""
    ct = {nat.F32: torch.float32, nat.BF16: torch.bfloat16, nat.F16: torch.float16}[desc.ct_dtype]
    if gy is None:  # only `scale` was used downstream
        if gscale is None:
            return None
        gy = torch.zeros(xc.shape, dtype=ct, device=xc.device)
    else:
        gy = _like_memory_order(gy.to(ct), back)
    if group is None and gscale is None and sp.channels > 1 and scale.numel() == sp.channels:
        # per-channel scale, nothing else feeding the scale's gradient: two launches in all -- the backward
        # kernel (dx, per-unit dscale sums and arg-max positions) and one finishing kernel that sums,
        # converts dscale into the statistic's gradient and deposits it (None: layout not covered)
        thr_div = _as_dtype_value(sp.int_threshold, scale.dtype)
        dx = nat.fakequant_bwd_stats(desc, gy, xc, scale.reshape(-1).contiguous(), zp.reshape(-1), stat,
                                     scale.dtype, thr_div, scale.dtype)
        if dx is not None:
            return _restore(dx, back)
    if group is not None and gscale is None and sp.channels > 1 and scale.numel() == sp.channels:
        # batch shard, per-channel scale: the backward kernel (its last-arriving wave per channel writes the shard's
        # all-gather message: double dscale sums + the claim on the deposit), ONE all-gather, one launch that adds the
        # shards' sums in double and deposits the statistic's gradient on the owning shard
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        out = nat.fakequant_bwd_shard(desc, gy, xc, scale.reshape(-1).contiguous(), zp.reshape(-1), stat, rank)
        if out is not None:
            dx, msg, pos = out
            if world > 1:
                gathered = torch.empty(world * msg.numel(), dtype=msg.dtype, device=msg.device)
                dist.all_gather_into_tensor(gathered, msg, group=group)
            else:
                gathered = msg
            thr_div = _as_dtype_value(sp.int_threshold, scale.dtype)
            nat.shard_unpack_deposit(xc.reshape(-1), dx.reshape(-1), gathered, world, sp.channels, rank, pos, sp.inner,
                                     scale.dtype, thr_div, scale.dtype, pre_op)
            return _restore(dx, back)
    # one pass: dx, the scale-gradient sums and the positions attaining the statistic
    dx, ds, _, ties = nat.fakequant_bwd(desc, gy, xc, scale.reshape(-1).contiguous(), zp.reshape(-1), True,
                                        False, tie_stat=stat)
    total_ties = None
    if group is None and gscale is None and ds.numel() == scale.numel():
        # dscale -> dstat -> deposit in one launch (same rounding points as the ops below)
        dimensioned = scale.dim() > 0
        quot_dtype = scale.dtype if dimensioned else torch.promote_types(scale.dtype, int_threshold.dtype)
        thr_div = _as_dtype_value(sp.int_threshold, scale.dtype) if dimensioned else sp.int_threshold
        nat.stat_tie_apply_dscale(xc.reshape(-1), stat, ds, scale.dtype, thr_div, quot_dtype, ties,
                                  dx.reshape(-1), sp.outer, sp.channels, sp.inner, pre_op=pre_op)
        return _restore(dx, back)
    if group is not None:
        # sum the shards' partial sums, and agree on which shard deposits the statistic's gradient
        from brevitas_amd.distributed import sync_backward
        if gscale is not None:
            ds = ds + gscale.reshape(-1).to(ds.dtype)
            gscale = None
        ds, ties, total_ties = sync_backward(ds, ties, sp.channels, group)
    ds = _reduce_like(ds, scale)
    if gscale is not None:
        ds = ds + gscale
    # scale = thr / int_threshold  ->  dthr = dscale / int_threshold ; clamp_min_ste passes it on
    dstat = (ds / int_threshold).to(stat.dtype).reshape(-1).contiguous()
    nat.stat_tie_apply(nat.MATCH_ABS, xc.reshape(-1), stat, dstat, ties, dx.reshape(-1), sp.outer,
                       sp.channels, sp.inner, mode_add=True, total_ties=total_ties, pre_op=pre_op)
    return _restore(dx, back)


# ---- the weight quantizer's autograd node in C++ (brevitas_amd/csrc/bvq_autograd.cpp) ------------------------------
# Optional host-side accelerator: one native call each way instead of the Python Function + ctypes wrappers (85 -> ~50 us
# of host time per weight-sized step).  Built in-tree by brevitas_amd/csrc/build.py; when the module is absent the
# Python Function above serves every call -- same kernels, same results either way.
_FAST = None          # the loaded extension module, False after a failed attempt
_FAST_PATH = None


class _SpLike(NamedTuple):
    outer: int
    channels: int
    inner: int
    int_threshold: float


def _fast_backward_fallback(xc, scale, zp, stat, int_threshold, dv, qr, shape, gy, gscale, group=None):
    """what the C++ nodes cannot do themselves (a gradient arriving through `scale`, an unaligned or strided gradient,
    a batch-sharded whole-tensor statistic)"""
    desc = nat.QuantDesc(dv[0], dv[1], dv[2], dv[3], dv[4], dv[5], dv[6], dv[7], dv[8], qr[0], qr[1], dv[9], dv[10],
                         dv[11], dv[12], dv[13], dv[14])
    sp = _SpLike(dv[0], dv[1], dv[2], qr[3])
    return stats_backward(xc, scale.view(shape), zp, stat, int_threshold, desc, sp, group, dv[13], None, gy, gscale)


def _fast_module():
    global _FAST, _FAST_PATH
    if _FAST is None:
        import importlib.util
        import os
        _FAST = False
        path = os.path.join(os.path.dirname(nat.LIB_PATH), '_bvq_autograd.so')
        if config.CPP_AUTOGRAD and os.path.exists(path):
            try:
                spec = importlib.util.spec_from_file_location('_bvq_autograd', path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.init(nat.LIB_PATH, _fast_backward_fallback)   # raises on another ABI version of libbvq.so
                _FAST, _FAST_PATH = mod, path
            except Exception:  # noqa: BLE001  (an extension built against another torch: the Python route serves)
                _FAST = False
    return _FAST


def fast_stats_fakequant(x, int_threshold, sp, qmin, qmax, round_mode, clamp_ste, pre_op):
    """StatsFakeQuantFn through the C++ node -> (y, scale, stat), or None: not a case it covers (per-tensor scale,
    channels_last / strided input, layouts outside the one-launch forward, no extension built, timers active)"""
    mod = _fast_module()
    if not mod or not x.is_cuda or sp.nhwc or sp.channels <= 1 or not x.is_contiguous() or x.dtype not in _FLOATS \
            or not config.FUSED_PATHS:
        return None
    timer = nat._timer
    if timer is not None and getattr(timer, 'enabled', True):
        return None  # bench.py is bracketing the C-ABI calls of this step with HIP events: keep them visible
    if x.device.index is not None and x.device.index != torch.cuda.current_device():
        return None  # the node launches on the current device: the Python route switches devices around its calls
    if len(sp.scaling_shape) > 0:
        scale_dtype, thr_div = x.dtype, _as_dtype_value(sp.int_threshold, x.dtype)
    else:
        scale_dtype = torch.promote_types(x.dtype, int_threshold.dtype)
        thr_div = sp.int_threshold
    code, sdt = nat.dtype_code(x.dtype), nat.dtype_code(scale_dtype)
    zp = _zero_zero_point(x.device)
    dv = (sp.outer, sp.channels, sp.inner, code, code, sdt, nat.dtype_code(zp.dtype), 1, 0, round_mode, scalar_mode(),
          int(clamp_ste), nat.OUT_DEQUANT, pre_op, 0)
    thr_bwd = _as_dtype_value(sp.int_threshold, scale_dtype)
    st = nat.stream_ptr(x.device)
    arrive = nat.arrival_buffer(x.device, st, sp.channels) if nat.ONEPASS_BWD else None
    return mod.stats_fakequant(x, zp, int_threshold, arrive, dv, float(qmin), float(qmax), float(sp.min_val or 0.0),
                               bool(sp.min_val), float(thr_div), float(thr_bwd), float(sp.int_threshold), sdt,
                               list(sp.scaling_shape), st)


def fast_act_stats_fakequant(x, int_threshold, sp, qmin, qmax, round_mode, clamp_ste, pre_op, group, runtime):
    """The activation route of StatsFakeQuantFn (statistic kernel + quantizer kernel, the running statistic folded in,
    optionally batch-sharded with its two collectives) through the C++ node -> (y, scale, stat) with
    `runtime.bvq_running_folded` set, or None: not a case it covers (channels_last / strided / unaligned input, a running
    buffer that is not a plain [channels] device tensor, no extension built, timers active)."""
    mod = _fast_module()
    if not mod or not hasattr(mod, 'act_stats_fakequant') or not x.is_cuda or sp.nhwc or not x.is_contiguous() \
            or x.dtype not in _FLOATS or not config.FUSED_PATHS or x.numel() == 0:
        return None
    timer = nat._timer
    if timer is not None and getattr(timer, 'enabled', True):
        return None  # bench.py is bracketing the C-ABI calls of this step with HIP events: keep them visible
    if x.device.index is not None and x.device.index != torch.cuda.current_device():
        return None
    if group is not None and not config.CPP_AUTOGRAD_SHARDED:
        return None
    running = None
    momentum, first = 0.0, False
    if runtime is not None:
        buf = runtime.running_stats
        if not (buf.is_cuda and buf.is_contiguous() and buf.numel() == sp.channels and buf.dtype in _FLOATS):
            return None
        running, momentum, first = buf, runtime.momentum, runtime.first_batch
    if len(sp.scaling_shape) > 0:
        scale_dtype, thr_div = x.dtype, _as_dtype_value(sp.int_threshold, x.dtype)
        quot_dtype, thr_bwd = scale_dtype, _as_dtype_value(sp.int_threshold, scale_dtype)
    else:
        scale_dtype = torch.promote_types(x.dtype, int_threshold.dtype)
        thr_div = sp.int_threshold
        quot_dtype, thr_bwd = torch.promote_types(scale_dtype, int_threshold.dtype), sp.int_threshold
    if sp.channels > 1 and len(sp.scaling_shape) == 0:
        return None
    ct = torch.result_type(x, torch.empty(sp.scaling_shape, dtype=scale_dtype, device='meta'))
    if ct != x.dtype:
        return None
    code, sdt = nat.dtype_code(x.dtype), nat.dtype_code(scale_dtype)
    zp = _zero_zero_point(x.device)
    st = nat.stream_ptr(x.device)
    arrive = nat.arrival_buffer(x.device, st, max(2 * sp.channels, 18))
    dv = (sp.outer, sp.channels, sp.inner, code, code, sdt, nat.dtype_code(zp.dtype), int(sp.channels > 1), 0, round_mode,
          scalar_mode(), int(clamp_ste), nat.OUT_DEQUANT, pre_op, 0)
    out = mod.act_stats_fakequant(x, zp, int_threshold, running, arrive, dv, float(qmin), float(qmax),
                                  float(sp.min_val or 0.0), bool(sp.min_val), float(thr_div), float(thr_bwd),
                                  float(sp.int_threshold), sdt, nat.dtype_code(quot_dtype), list(sp.scaling_shape), st,
                                  float(momentum), bool(first), group)
    if out is not None and runtime is not None:
        runtime.bvq_running_folded = True
    return out
