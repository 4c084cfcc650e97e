"""RescalingIntQuant (drop-in for B/core/quant/int.py:94-163): obtains bit width, scale and zero
point from its sub-modules and applies IntQuant; returns (y, scale, zero_point, bit_width).

Sub-module attribute names (`int_quant`, `scaling_impl`, `int_scaling_impl`, `zero_point_impl`,
`msb_clamp_bit_width_impl`) are the reference's: they are part of state-dict keys and are
introspected by proxies.

Two recognised graphs skip the op-by-op orchestration and run as statistic kernel + tiny scale
ops + quantize kernel, with a backward that never materialises torch.max's dense gradient:
  * weights:      StatsFromParameterScaling(AbsMax) tracking exactly the tensor being quantized
                  (Int8WeightPerChannelFloat and friends);
  * activations:  RuntimeStatsScaling(AbsMax) in training mode.
Everything else takes the generic route below, which is the reference's own sequence.
"""
from typing import Tuple

import torch
from torch import Tensor
from torch.nn import Module

import brevitas_amd.config as config
from brevitas_amd.core.function_wrapper.shape import OverOutputChannelView, OverTensorView
from brevitas_amd.core.quant import _fused
from brevitas_amd.core.quant.delay import _NoDelay
from brevitas_amd.core.quant.int_base import IntQuant
from brevitas_amd.core.scaling.int_scaling import IntScaling
from brevitas_amd import _native as nat
from brevitas_amd.core.scaling.runtime import RuntimeStatsScaling, StatsFromParameterScaling
from brevitas_amd.core.scaling.standalone import ConstScaling, ParameterFromRuntimeStatsScaling, ParameterScaling
from brevitas_amd.core.stats.stats_op import AbsMax
from brevitas_amd.core.zero_point import ZeroZeroPoint
from brevitas_amd.function.ops import int_range_host


class RescalingIntQuant(torch.nn.Module):
    """
    Examples (B/core/quant/int.py:113-134):
        >>> q = RescalingIntQuant(IntQuant(narrow_range=True, signed=True), ConstScaling(0.1),
        ...                       IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthConst(4))
        >>> out, scale, zero_point, bit_width = q(torch.Tensor([0.042, -0.053, 0.31, -0.44]))
        >>> out
        tensor([ 0.0429, -0.0571,  0.1000, -0.1000])
    """

    def __init__(self, int_quant: Module, scaling_impl: Module, int_scaling_impl: Module, zero_point_impl: Module,
                 bit_width_impl: Module):
        super().__init__()
        self.int_quant = int_quant
        self.scaling_impl = scaling_impl
        self.int_scaling_impl = int_scaling_impl
        self.zero_point_impl = zero_point_impl
        self.msb_clamp_bit_width_impl = bit_width_impl

    # ---- recognised graphs -----------------------------------------------------------------------------

    def _stats_template(self, bit_width: Tensor):
        """Structural part of the fused-graph recognition, cached until the module graph, the training
        flag or the configuration changes: None, or a dict describing the statistic -> scale chain."""
        iq, sc = self.int_quant, self.scaling_impl
        bw = getattr(bit_width, 'bvq_host_value', None)
        key = (config.FUSED_PATHS, id(iq), id(sc), id(self.zero_point_impl), id(self.int_scaling_impl), bw,
               getattr(sc, 'training', None), id(getattr(iq, 'float_to_int_impl', None)),
               id(getattr(iq, 'tensor_clamp_impl', None)),
               id(getattr(getattr(iq, 'delay_wrapper', None), 'delay_impl', None)))
        cached = self.__dict__.get('_bvq_template')
        if cached is not None and cached[0] == key:
            return cached[1]
        tmpl = self._build_stats_template(bw)
        self.__dict__['_bvq_template'] = (key, tmpl)
        return tmpl

    def _build_stats_template(self, bw):
        if not config.FUSED_PATHS or bw is None:
            return None
        iq = self.int_quant
        if type(iq) is not IntQuant or not isinstance(iq.delay_wrapper.delay_impl, _NoDelay):
            return None
        if getattr(iq.float_to_int_impl, 'bvq_round_mode', None) is None or \
                getattr(iq.tensor_clamp_impl, 'bvq_clamp_ste', None) is None:
            return None
        if type(self.zero_point_impl) is not ZeroZeroPoint or type(self.int_scaling_impl) is not IntScaling:
            return None
        sc = self.scaling_impl
        runtime, weight, shared = None, None, None
        if type(sc) is RuntimeStatsScaling:
            if not sc.training:
                return None
            runtime = sc.runtime_stats
            view, stats = runtime.stats_input_view_shape_impl, runtime.stats
        elif type(sc) is StatsFromParameterScaling:
            pls = sc.parameter_list_stats
            weight = pls.first_tracked_param.parameter
            view, stats = pls.first_tracked_param.view_shape_impl, pls.stats
            if pls.extra_tracked_params_list is not None:
                # a quantizer shared by several layers: the statistic of all their weights in ONE launch over the list
                # (_fused.ListStatsFakeQuantFn), no torch.cat
                if any(type(e.view_shape_impl) is not type(view) for e in pls.extra_tracked_params_list):
                    return None
                shared = [weight] + [e.parameter for e in pls.extra_tracked_params_list]
        else:
            return None
        min_val = sc.stats_scaling_impl.bvq_plain_min_val()
        if type(stats.stats_impl) is not AbsMax:
            return None
        # a statistic -> threshold map that is NOT the plain lower bound (affine rescaling, a power-of-two restriction):
        # the statistic and quantizer kernels still apply, the map runs as a few scale-shaped torch ops between them and
        # is differentiated by autograd on those small tensors (_fused.StatsGraphFakeQuantFn)
        post = sc.stats_scaling_impl if min_val is None else None
        shape = tuple(stats.stats_output_shape)
        if type(view) is OverTensorView and stats.stats_impl.stats_reduce_dim is None:
            if shape != ():
                return None
            per_channel = False
        elif type(view) is OverOutputChannelView and stats.stats_impl.stats_reduce_dim in (1, -1):
            per_channel = True
        else:
            return None
        qmin, qmax = int_range_host(iq.signed, iq.narrow_range, bw)
        if shared is not None and (post is not None or len(shared) > 8):
            return None
        return dict(runtime=runtime, weight=weight, shared=shared, view=view, shape=shape, min_val=min_val, post=post,
                    per_channel=per_channel, int_thr=self.int_scaling_impl.host_value(bw), qmin=qmin, qmax=qmax,
                    round_mode=iq.float_to_int_impl.bvq_round_mode, clamp_ste=iq.tensor_clamp_impl.bvq_clamp_ste)

    def _stats_plan(self, x: Tensor, bit_width: Tensor):
        """(StatsPlan, template) if a recognised fused graph applies to this input, else None"""
        if not x.is_cuda or x.dim() == 0 or x.numel() == 0:
            return None
        tmpl = self._stats_template(bit_width)
        if tmpl is None:
            return None
        w = tmpl['weight']
        if tmpl['shared'] is not None:
            # the tensor being quantized is one of the list; every tensor lies [channels, ...] (or flat) in memory
            same = [i for i, t in enumerate(tmpl['shared'])
                    if t is x or (t.data_ptr() == x.data_ptr() and t.shape == x.shape and t.stride() == x.stride())]
            ch0 = x.shape[0] if tmpl['per_channel'] else None
            if not same or not x.is_contiguous() or any(
                    (not t.is_cuda) or t.device != x.device or t.dtype != x.dtype or not t.is_contiguous()
                    or t.dim() == 0 or t.numel() == 0 or (ch0 is not None and t.shape[0] != ch0)
                    for t in tmpl['shared']):
                return None
            if tmpl['per_channel'] and tmpl['view'].bvq_channel_dim(x.dim()) != 0:
                return None
        elif w is not None and not (w is x or (w.data_ptr() == x.data_ptr() and w.shape == x.shape
                                               and w.stride() == x.stride() and w.dtype == x.dtype)):
            return None  # the statistic is taken of another tensor than the one being quantized
        shape = tmpl['shape']
        if not tmpl['per_channel']:
            return _fused.StatsPlan(1, 1, x.numel(), shape, tmpl['min_val'], tmpl['int_thr']), tmpl
        cd = tmpl['view'].bvq_channel_dim(x.dim())
        if cd is None or cd < 0:
            return None
        # the scaling shape must broadcast against x exactly at the channel dim
        want = tuple(x.shape[cd] if i == cd else 1 for i in range(x.dim()))
        padded = (1,) * (x.dim() - len(shape)) + shape
        if len(shape) > x.dim() or padded != want or x.shape[cd] == 1:
            return None
        if _fused.is_nhwc(x, cd):  # dense channels_last activation: [N*H*W, C] in memory, column-mapped kernels
            return _fused.StatsPlan(x.numel() // x.shape[1], x.shape[1], 1, shape, tmpl['min_val'], tmpl['int_thr'],
                                    True), tmpl
        outer = 1
        for d in x.shape[:cd]:
            outer *= d
        inner = 1
        for d in x.shape[cd + 1:]:
            inner *= d
        return _fused.StatsPlan(outer, x.shape[cd], inner, shape, tmpl['min_val'], tmpl['int_thr']), tmpl

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        return self.bvq_forward_pre(x, nat.PRE_NONE)

    def _learned_scale_args(self, x: Tensor, bit_width: Tensor):
        """(value, min_val, Plan, thr_div, scale_dtype, qmin, qmax, round_mode, clamp_ste) if the scale is a learned
        parameter with a float restriction right now -- scale = |clamp_min(value)| / int_threshold -- and the fused
        quantizer kernel covers the operands; else None"""
        if not config.FUSED_PATHS or not x.is_cuda:
            return None
        bw = getattr(bit_width, 'bvq_host_value', None)
        iq, sc = self.int_quant, self.scaling_impl
        if bw is None or type(iq) is not IntQuant or not isinstance(iq.delay_wrapper.delay_impl, _NoDelay):
            return None
        round_mode = getattr(iq.float_to_int_impl, 'bvq_round_mode', None)
        clamp_ste = getattr(iq.tensor_clamp_impl, 'bvq_clamp_ste', None)
        if round_mode is None or clamp_ste is None:
            return None
        if type(self.zero_point_impl) is not ZeroZeroPoint or type(self.int_scaling_impl) is not IntScaling:
            return None
        learned = getattr(sc, 'bvq_learned_scale', None)
        found = learned() if learned is not None else None
        if found is None:
            return None
        value, min_val = found
        if not value.is_cuda or value.dtype not in _fused._FLOATS or value.numel() < 1:
            return None
        zp = _fused._zero_zero_point(x.device)
        p = _fused.plan(x, value, zp)
        if p is None or p.zp_pc:
            return None
        int_thr = self.int_scaling_impl.host_value(bw)
        if value.dim() > 0:  # a dimensioned threshold keeps its dtype: the 0-dim int_threshold is converted to it
            scale_dtype, thr_div = value.dtype, _fused._as_dtype_value(int_thr, value.dtype)
        else:                # two 0-dim operands promote like tensors (int_threshold has the bit width's dtype)
            scale_dtype = torch.promote_types(value.dtype, bit_width.dtype)
            thr_div = _fused._as_dtype_value(int_thr, bit_width.dtype)
        qmin, qmax = int_range_host(iq.signed, iq.narrow_range, bw)
        return value, min_val, p, thr_div, scale_dtype, qmin, qmax, round_mode, clamp_ste

    def _scale_ignores_input(self) -> bool:
        """True if scaling_impl(x) does not read x right now (learned / constant / frozen statistics)"""
        sc = self.scaling_impl
        if type(sc) in (ParameterScaling, ConstScaling):
            return True
        if type(sc) is RuntimeStatsScaling:
            return not sc.training
        if type(sc) is ParameterFromRuntimeStatsScaling:
            return (not sc.training) or sc.counter >= sc.collect_stats_steps
        return False

    def _collect_only(self, x: Tensor, pre_op: int, bit_width: Tensor):
        """Calibration forward (brevitas_amd.graph.calibrate): advance the statistics of the scale and
        zero-point modules exactly as a full forward would, hand the (activated) float tensor on, skip the
        quantize/dequantize pass -- the reference computes it and a hook discards it
        (B/graph/calibrate.py:115-127).  Not differentiable, like a PTQ forward under no_grad."""
        if not isinstance(self.int_quant.delay_wrapper.delay_impl, _NoDelay):
            raise NotImplementedError('calibration with quant_delay_steps')
        with torch.no_grad():
            fused = self._stats_plan(x, bit_width)
            if fused is not None and fused[1]['runtime'] is not None:
                sp, tmpl = fused
                group = getattr(self, 'bvq_shard_group', None)
                flat = _fused._memory_order(x, sp.channels, sp.nhwc)[0].reshape(-1)
                stat, scale = _fused.stats_scale(flat, self.int_scaling_impl(bit_width), sp, group, pre_op)
                tmpl['runtime'].update_running_stats(stat.view(sp.scaling_shape))
                x_act = torch.relu(x) if pre_op == nat.PRE_RELU else x
            else:
                x_act = torch.relu(x) if pre_op == nat.PRE_RELU else x
                threshold = self.scaling_impl(x_act)
                scale = threshold / self.int_scaling_impl(bit_width)
            zero_point = self.zero_point_impl(x_act, scale, bit_width)
        return x_act, scale, zero_point, bit_width

    def bvq_forward_pre(self, x: Tensor, pre_op: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """forward(pre_op(x)) -- FusedActivationQuantProxy.forward (B/proxy/runtime_quant.py:80-84) -- with
        the activation folded into the statistic / quantizer kernels wherever those kernels apply, so
        the activation's own read + write (and its backward pass) disappear."""
        bit_width = self.msb_clamp_bit_width_impl()
        if getattr(self, 'bvq_collect_only', False):
            return self._collect_only(x, pre_op, bit_width)
        fused = self._stats_plan(x, bit_width)
        if fused is None and pre_op != nat.PRE_NONE:
            if type(self.int_quant) is IntQuant and type(self.zero_point_impl) is ZeroZeroPoint \
                    and self._scale_ignores_input():
                threshold = self.scaling_impl(x)
                int_threshold = self.int_scaling_impl(bit_width)
                scale = threshold / int_threshold
                zero_point = self.zero_point_impl(x, scale, bit_width)
                y = self.int_quant.bvq_forward_pre(scale, zero_point, bit_width, x, pre_op)
                return y, scale, zero_point, bit_width
            x = torch.relu(x)  # not fusable here: materialise the activation like the reference does
            pre_op = nat.PRE_NONE
        if fused is not None:
            sp, tmpl = fused
            runtime = tmpl['runtime']
            int_threshold = self.int_scaling_impl(bit_width)
            # a batch-sharded activation (brevitas_amd.distributed.shard_over_batch); weights are replicated
            group = getattr(self, 'bvq_shard_group', None) if runtime is not None else None
            if runtime is not None:
                runtime.bvq_running_folded = False
            fast = None
            if tmpl['shared'] is not None:
                params = tuple(tmpl['shared'])
                index = next(i for i, t in enumerate(params) if t.data_ptr() == x.data_ptr() and t.shape == x.shape)
                if pre_op == nat.PRE_NONE and _fused.list_stats_supported(params, sp):
                    y, scale, stat = _fused.ListStatsFakeQuantFn.apply(
                        x, int_threshold, sp, tmpl['qmin'], tmpl['qmax'], tmpl['round_mode'], tmpl['clamp_ste'], index,
                        *params)
                    zero_point = self.zero_point_impl(x, scale, bit_width)
                    return y, scale, zero_point, bit_width
                fused = None  # list not covered by the list kernel: op by op
                if pre_op != nat.PRE_NONE:
                    x, pre_op = torch.relu(x), nat.PRE_NONE
        if fused is not None:
            sp, tmpl = fused
            if tmpl['post'] is not None:
                if group is not None:
                    raise NotImplementedError('batch-sharded quantizer with a non-trivial statistic -> scale map')
                post = tmpl['post']
                y, scale, stat = _fused.StatsGraphFakeQuantFn.apply(
                    x, int_threshold, sp, tmpl['qmin'], tmpl['qmax'], tmpl['round_mode'], tmpl['clamp_ste'], pre_op,
                    post, *tuple(post.parameters()))
                if runtime is not None:
                    runtime.update_running_stats(stat)
                zero_point = self.zero_point_impl(x, scale, bit_width)
                return y, scale, zero_point, bit_width
            if runtime is None and group is None:  # a weight: the autograd node in C++ when it is built and applies
                fast = _fused.fast_stats_fakequant(x, int_threshold, sp, tmpl['qmin'], tmpl['qmax'], tmpl['round_mode'],
                                                   tmpl['clamp_ste'], pre_op)
            elif config.CPP_AUTOGRAD:              # an activation: statistic kernel + quantizer kernel, same idea
                fast = _fused.fast_act_stats_fakequant(x, int_threshold, sp, tmpl['qmin'], tmpl['qmax'],
                                                       tmpl['round_mode'], tmpl['clamp_ste'], pre_op, group, runtime)
            if fast is not None:
                y, scale, stat = fast
            else:
                y, scale, stat = _fused.StatsFakeQuantFn.apply(
                    x, int_threshold, sp, tmpl['qmin'], tmpl['qmax'], tmpl['round_mode'], tmpl['clamp_ste'], group,
                    pre_op, runtime)
            if runtime is not None:
                if runtime.bvq_running_folded:   # updated by the statistic's own finishing launch
                    runtime.first_batch = False
                else:
                    runtime.update_running_stats(stat)
            zero_point = self.zero_point_impl(x, scale, bit_width)
            return y, scale, zero_point, bit_width
        learned = self._learned_scale_args(x, bit_width)
        if learned is not None:
            # learned scale (steady state of the default activation quantizers): one launch for the scale, the
            # quantizer kernel; in backward the scale's own chain rides on the last reduction launch
            value, min_val, p, thr_div, scale_dtype, qmin, qmax, round_mode, clamp_ste = learned
            y, scale = _fused.LearnedScaleFakeQuantFn.apply(x, value, p, min_val, thr_div, scale_dtype, qmin, qmax,
                                                            round_mode, clamp_ste, pre_op)
            zero_point = self.zero_point_impl(x, scale, bit_width)
            return y, scale, zero_point, bit_width
        # generic orchestration (B/core/quant/int.py:157-163)
        threshold = self.scaling_impl(x)
        int_threshold = self.int_scaling_impl(bit_width)
        scale = threshold / int_threshold
        zero_point = self.zero_point_impl(x, scale, bit_width)
        y = self.int_quant(scale, zero_point, bit_width, x)
        return y, scale, zero_point, bit_width


class PrescaledRestrictIntQuantWithInputBitWidth(torch.nn.Module):
    """IntQuant with an externally supplied scale, zero zero-point, and a bit width derived from the
    input's (drop-in for B/core/quant/int.py:17-69; bias quantization with scale = input * weight scale).

    Examples (B/core/quant/int.py:32-48):
        >>> q = PrescaledRestrictIntQuantWithInputBitWidth(IntQuant(narrow_range=True, signed=True), Identity())
        >>> out, scale, zero_point, bit_width = q(inp, torch.tensor(0.01), torch.tensor(4.))
        >>> out
        tensor([ 0.0400, -0.0500,  0.0700, -0.0700])
    """

    def __init__(self, int_quant: Module, bit_width_impl: Module):
        super().__init__()
        from brevitas_amd.core.utils import StatelessBuffer
        self.int_quant = int_quant
        self.msb_clamp_bit_width_impl = bit_width_impl
        self.zero_point = StatelessBuffer(torch.tensor(0.0))

    def forward(self, x: Tensor, scale: Tensor, input_bit_width: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        bit_width = self.msb_clamp_bit_width_impl(input_bit_width)
        zero_point = self.zero_point()
        y = self.int_quant(scale, zero_point, bit_width, x)
        return y, scale, zero_point, bit_width


class PrescaledRestrictIntQuant(torch.nn.Module):
    """IntQuant with an externally supplied scale, zero zero-point and its own bit width
    (drop-in for B/core/quant/int.py:72-91)"""

    def __init__(self, int_quant: Module, bit_width_impl: Module):
        super().__init__()
        from brevitas_amd.core.utils import StatelessBuffer
        self.int_quant = int_quant
        self.msb_clamp_bit_width_impl = bit_width_impl
        self.zero_point = StatelessBuffer(torch.tensor(0.0))

    def forward(self, x: Tensor, scale: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        msb_clamp_bit_width = self.msb_clamp_bit_width_impl()
        zero_point = self.zero_point()
        y = self.int_quant(scale, zero_point, msb_clamp_bit_width, x)
        return y, scale, zero_point, msb_clamp_bit_width


class DecoupledRescalingIntQuant(torch.nn.Module):
    """DecoupledIntQuant with scales / zero-points from sub-modules (drop-in for B/core/quant/int.py:166-196);
    returns (y, scale, zero_point, bit_width, pre_scale, pre_zero_point)"""

    def __init__(self, decoupled_int_quant: Module, pre_scaling_impl: Module, scaling_impl: Module,
                 int_scaling_impl: Module, pre_zero_point_impl: Module, zero_point_impl: Module, bit_width_impl: Module):
        super().__init__()
        self.decoupled_int_quant = decoupled_int_quant
        self.pre_scaling_impl = pre_scaling_impl
        self.scaling_impl = scaling_impl
        self.int_scaling_impl = int_scaling_impl
        self.pre_zero_point_impl = pre_zero_point_impl
        self.zero_point_impl = zero_point_impl
        self.msb_clamp_bit_width_impl = bit_width_impl

    def forward(self, x: Tensor):
        bit_width = self.msb_clamp_bit_width_impl()
        int_threshold = self.int_scaling_impl(bit_width)
        pre_threshold = self.pre_scaling_impl(x)
        pre_scale = pre_threshold / int_threshold
        pre_zero_point = self.pre_zero_point_impl(x, pre_scale, bit_width)
        threshold = self.scaling_impl(x)
        scale = threshold / int_threshold
        zero_point = self.zero_point_impl(x, scale, bit_width)
        y = self.decoupled_int_quant(pre_scale, pre_zero_point, scale, zero_point, bit_width, x)
        return y, scale, zero_point, bit_width, pre_scale, pre_zero_point


class TruncIntQuant(torch.nn.Module):
    """Truncation of an already quantized value to fewer bits (drop-in for B/core/quant/int.py:199-229):
    recover the integer, drop `input_bit_width - output_bit_width` LSBs with float_to_int_impl, de-quantize"""

    def __init__(self, float_to_int_impl: Module, bit_width_impl: Module, quant_delay_steps: int = 0):
        super().__init__()
        from brevitas_amd.core.quant.delay import DelayWrapper
        self.msb_clamp_bit_width_impl = bit_width_impl
        self.float_to_int_impl = float_to_int_impl
        self.delay_wrapper = DelayWrapper(quant_delay_steps)

    def forward(self, x: Tensor, scale: Tensor, zero_point: Tensor, input_bit_width: Tensor
                ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        from brevitas_amd.function.ops_ste import round_ste
        output_bit_width = self.msb_clamp_bit_width_impl()
        # both bit widths known on the host (constant bit widths handed on by the layers): the whole chain is ONE
        # kernel, its autograd one more (include/bvq.h, bvq_variant_fwd: BVQ_VAR_TRUNC)
        in_bw = getattr(input_bit_width, 'bvq_host_value', None)
        out_bw = getattr(output_bit_width, 'bvq_host_value', None)
        round_mode = getattr(self.float_to_int_impl, 'bvq_round_mode', None)
        if in_bw is not None and out_bw is not None and round_mode is not None and \
                _fused.scalar_zero_point_ok(zero_point, x=x):
            p = _fused.variant_plan(x, scale)
            ct = torch.result_type(x, scale)
            if p is not None and (ct == x.dtype or ct == torch.float32):
                y = _fused.VariantFn.apply(x, scale, None, zero_point, None, p,
                                           dict(kind=nat.VAR_TRUNC, ct=ct, round_mode=round_mode,
                                                trunc_scale=float(2.0 ** (in_bw - out_bw))))
                return self.delay_wrapper(x, y), scale, zero_point, output_bit_width
        y = x / scale
        y = y + zero_point
        y = round_ste(y)  # clean up floating point error
        trunc_bit_width = input_bit_width - output_bit_width
        trunc_scale = 2.0 ** trunc_bit_width
        y = y / trunc_scale
        y = self.float_to_int_impl(y)
        y = y - zero_point
        y = y * scale
        y = self.delay_wrapper(x, y)
        return y, scale, zero_point, output_bit_width
