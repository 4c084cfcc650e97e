"""Resolved module graphs of the named quantizers on the accelerated path.

In Brevitas these are injector classes (B/quant/scaled_int.py:144-193) that the solvers
(B/quant/solver/*.py) resolve into a `tensor_quant` module graph; the injector machinery itself is
out of scope, so the functions here assemble exactly the graphs the solvers produce
(SURVEY 8a lists them with file:line) from brevitas_amd's same-named modules.  Each returns the
`tensor_quant` a proxy would own: `q(x) -> (y, scale, zero_point, bit_width)`.
"""
from typing import List, Optional, Sequence, Union

import torch

from brevitas_amd.core.bit_width import BitWidthConst
from brevitas_amd.core.function_wrapper import (CeilSte, OverOutputChannelView, OverTensorView, RoundSte, TensorClamp,
                                                TensorClampSte)
from brevitas_amd.core.quant import IntQuant, PrescaledRestrictIntQuant, RescalingIntQuant
from brevitas_amd.core.restrict_val import FloatRestrictValue, PowerOfTwoRestrictValue
from brevitas_amd.core.scaling import (IntScaling, ParameterFromRuntimeStatsScaling, ParameterScaling,
                                       PowerOfTwoIntScaling, RuntimeStatsScaling, StatsFromParameterScaling)
from brevitas_amd.core.stats import (AbsMax, AbsMinMax, AbsPercentile, NegativeMinOrZero, NegativePercentileOrZero,
                                     PercentileInterval)
from brevitas_amd.core.zero_point import ParameterFromRuntimeZeroPoint, StatsFromParameterZeroPoint, ZeroZeroPoint

__all__ = ['Int8WeightPerChannelFloat', 'Int4WeightPerChannelFloat', 'Int8WeightPerTensorFloat',
           'Int8ActPerTensorFloat', 'Uint8ActPerTensorFloat', 'Int8ActPerChannelFloat',
           'ShiftedUint8WeightPerTensorFloat', 'ShiftedUint8WeightPerChannelFloat', 'ShiftedUint8ActPerTensorFloat',
           'Int8WeightPerTensorFixedPoint', 'Int8WeightPerChannelFixedPoint', 'Int8ActPerTensorFixedPoint',
           'Uint8ActPerTensorFixedPoint', 'Uint8ActPerTensorFixedPointMaxInit', 'Int8Bias', 'Int16Bias', 'Int24Bias',
           'Int32Bias', 'Int8BiasPerTensorFloatInternalScaling', 'Int8BiasPerTensorFixedPointInternalScaling']

SCALING_MIN_VAL = 1e-10  # B/quant/base.py:115-123, 169-182


def _params(weights) -> List[torch.nn.Parameter]:
    return list(weights) if isinstance(weights, (list, tuple)) else [weights]


def Int8WeightPerChannelFloat(weights: Union[torch.nn.Parameter, Sequence[torch.nn.Parameter]],
                              bit_width: int = 8) -> RescalingIntQuant:
    """NarrowIntQuant + MaxStatsScaling + PerChannelFloatScaling8bit + WeightQuantSolver
    (B/quant/scaled_int.py:157-167): scale[c] = max(max_k |w[c,k]|, 1e-10) / (2^(b-1) - 1), narrow signed range,
    straight-through clamp; weights are re-quantized on every forward."""
    tracked = _params(weights)
    w = tracked[0]
    shape = (w.shape[0],) + (1,) * (w.dim() - 1)
    return RescalingIntQuant(
        IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
        StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, tracked, FloatRestrictValue(), shape,
                                  affine_rescaling=False, scaling_min_val=SCALING_MIN_VAL),
        IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthConst(bit_width))


def Int4WeightPerChannelFloat(weights) -> RescalingIntQuant:
    """not in this reference snapshot; defined as Int8WeightPerChannelFloat with bit_width = 4"""
    return Int8WeightPerChannelFloat(weights, bit_width=4)


def Int8WeightPerTensorFloat(weights, bit_width: int = 8) -> RescalingIntQuant:
    """B/quant/scaled_int.py:144-154: one scale for the whole weight tensor"""
    tracked = _params(weights)
    return RescalingIntQuant(
        IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
        StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, tracked, FloatRestrictValue(), (),
                                  affine_rescaling=False, scaling_min_val=SCALING_MIN_VAL),
        IntScaling(signed=True, narrow_range=True), ZeroZeroPoint(), BitWidthConst(bit_width))


def _act_quant(signed: bool, bit_width: int, scaling_impl_type: str, collect_stats_steps: int,
               channels: Optional[int], scaling_init: Optional[float], scaling_stats_op: str = 'max'
               ) -> RescalingIntQuant:
    if channels is None:
        # the reference's default statistic is the 99.999th percentile (B/quant/base.py:68-75);
        # scaling_stats_op='max' is its supported StatsOp.MAX override
        stats = AbsPercentile(99.999, None) if scaling_stats_op == 'percentile' else AbsMax()
        view, shape = OverTensorView(), ()
    else:
        # scaling_per_output_channel=True, per_channel_broadcastable_shape=(1,C,1,1),
        # scaling_stats_permute_dims=(1,0,2,3)  (B/quant/solver/act.py:91-105)
        view, stats, shape = OverOutputChannelView((1, 0, 2, 3)), AbsMax(1), (1, channels, 1, 1)
    if scaling_impl_type == 'parameter_from_stats':
        scaling = ParameterFromRuntimeStatsScaling(collect_stats_steps, stats, view, shape, FloatRestrictValue(),
                                                   0.1, SCALING_MIN_VAL)
    elif scaling_impl_type == 'stats':
        scaling = RuntimeStatsScaling(stats, view, FloatRestrictValue(), shape, affine_rescaling=False,
                                      scaling_stats_momentum=0.1, scaling_min_val=SCALING_MIN_VAL)
    elif scaling_impl_type == 'parameter':
        scaling = ParameterScaling(scaling_init, shape if shape else None, FloatRestrictValue(), SCALING_MIN_VAL)
    else:
        raise ValueError("scaling_impl_type must be 'parameter_from_stats', 'stats' or 'parameter'")
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=signed, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        scaling, IntScaling(signed=signed, narrow_range=False), ZeroZeroPoint(), BitWidthConst(bit_width))


def Int8ActPerTensorFloat(scaling_impl_type: str = 'parameter_from_stats', collect_stats_steps: int = 300,
                          bit_width: int = 8, scaling_init: Optional[float] = None,
                          scaling_stats_op: str = 'percentile') -> RescalingIntQuant:
    """IntQuant + ParamFromRuntimePercentileScaling + PerTensorFloatScaling8bit + ActQuantSolver
    (B/quant/scaled_int.py:170-180): collects the 99.999th percentile of |x| (scaling_stats_op='max':
    AbsMax, the reference's StatsOp.MAX override) for `collect_stats_steps` training steps, then learns
    the scale."""
    return _act_quant(True, bit_width, scaling_impl_type, collect_stats_steps, None, scaling_init,
                      scaling_stats_op)


def Uint8ActPerTensorFloat(scaling_impl_type: str = 'parameter_from_stats', collect_stats_steps: int = 300,
                           bit_width: int = 8, scaling_init: Optional[float] = None,
                           scaling_stats_op: str = 'percentile') -> RescalingIntQuant:
    """unsigned variant for post-ReLU activations (B/quant/scaled_int.py:183-193)"""
    return _act_quant(False, bit_width, scaling_impl_type, collect_stats_steps, None, scaling_init,
                      scaling_stats_op)


def Int8ActPerChannelFloat(channels: int, scaling_impl_type: str = 'stats', collect_stats_steps: int = 300,
                           bit_width: int = 8) -> RescalingIntQuant:
    """Int8ActPerTensorFloat with scaling_per_output_channel=True over NCHW channel `channels`
    (the layout of BASELINE.json's metric)"""
    return _act_quant(True, bit_width, scaling_impl_type, collect_stats_steps, channels, None)


def _shifted_weight_quant(weights, per_channel: bool, bit_width: int) -> RescalingIntQuant:
    """ShiftedMinUintQuant + MinMaxStatsScaling (B/quant/base.py:60-65,137-150): unsigned codes, scale from
    max - min, integer zero-point from -min / scale; both statistics are back-propagated through"""
    tracked = _params(weights)
    w = tracked[0]
    if per_channel:
        shape = (w.shape[0],) + (1,) * (w.dim() - 1)
        view = lambda: OverOutputChannelView(None)  # noqa: E731
        scale_stat, zp_stat, cat = AbsMinMax(1), NegativeMinOrZero(1), 1
    else:
        shape = ()
        view = lambda: OverTensorView()  # noqa: E731
        scale_stat, zp_stat, cat = AbsMinMax(), NegativeMinOrZero(), 0
    int_quant = IntQuant(narrow_range=False, signed=False, float_to_int_impl=RoundSte(),
                         tensor_clamp_impl=TensorClampSte())
    return RescalingIntQuant(
        int_quant,
        StatsFromParameterScaling(scale_stat, view(), cat, tracked, FloatRestrictValue(), shape,
                                  affine_rescaling=False, scaling_min_val=SCALING_MIN_VAL),
        IntScaling(signed=False, narrow_range=False),
        StatsFromParameterZeroPoint(int_quant, True, view(), cat, zp_stat, shape, tracked),
        BitWidthConst(bit_width))


def ShiftedUint8WeightPerTensorFloat(weights, bit_width: int = 8) -> RescalingIntQuant:
    """B/quant/shifted_scaled_int.py:37-52"""
    return _shifted_weight_quant(weights, False, bit_width)


def ShiftedUint8WeightPerChannelFloat(weights, bit_width: int = 8) -> RescalingIntQuant:
    """B/quant/shifted_scaled_int.py:55-70"""
    return _shifted_weight_quant(weights, True, bit_width)


def ShiftedUint8ActPerTensorFloat(collect_stats_steps: int = 300, bit_width: int = 8) -> RescalingIntQuant:
    """ShiftedParamFromPercentileUintQuant + ParamFromRuntimePercentileIntervalScaling
    (B/quant/shifted_scaled_int.py:19-34, base.py:87-95,153-166): scale from the 0.001..99.999 percentile
    interval and zero-point from the 0.001th percentile, both collected for `collect_stats_steps`
    training steps and then learned"""
    int_quant = IntQuant(narrow_range=False, signed=False, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp())
    return RescalingIntQuant(
        int_quant,
        ParameterFromRuntimeStatsScaling(collect_stats_steps, PercentileInterval(0.001, 99.999, None),
                                         OverTensorView(), (), FloatRestrictValue(), 0.1, SCALING_MIN_VAL),
        IntScaling(signed=False, narrow_range=False),
        ParameterFromRuntimeZeroPoint(collect_stats_steps, int_quant, True, NegativePercentileOrZero(0.001, None),
                                      (), OverTensorView(), 0.1),
        BitWidthConst(bit_width))


# ---- fixed point: power-of-two scales (B/quant/fixed_point.py:23-73, PerTensorPoTScaling8bit B/quant/base.py:185-191)

def _pot():
    return PowerOfTwoRestrictValue(CeilSte())


def Int8WeightPerTensorFixedPoint(weights, bit_width: int = 8) -> RescalingIntQuant:
    """NarrowIntQuant + MaxStatsScaling + PerTensorPoTScaling8bit (B/quant/fixed_point.py:23-34):
    scale = 2^ceil(log2 max|w|) / 2^(b-1); the radix point follows the back-propagated statistic"""
    tracked = _params(weights)
    return RescalingIntQuant(
        IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
        StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, tracked, _pot(), (), affine_rescaling=False,
                                  scaling_min_val=SCALING_MIN_VAL),
        PowerOfTwoIntScaling(signed=True), ZeroZeroPoint(), BitWidthConst(bit_width))


def Int8WeightPerChannelFixedPoint(weights, bit_width: int = 8) -> RescalingIntQuant:
    """Int8WeightPerTensorFixedPoint with scaling_per_output_channel=True: one radix point per output channel"""
    tracked = _params(weights)
    w = tracked[0]
    shape = (w.shape[0],) + (1,) * (w.dim() - 1)
    return RescalingIntQuant(
        IntQuant(narrow_range=True, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClampSte()),
        StatsFromParameterScaling(AbsMax(1), OverOutputChannelView(None), 1, tracked, _pot(), shape,
                                  affine_rescaling=False, scaling_min_val=SCALING_MIN_VAL),
        PowerOfTwoIntScaling(signed=True), ZeroZeroPoint(), BitWidthConst(bit_width))


def _act_fixed_point(signed: bool, collect_stats_steps: int, bit_width: int) -> RescalingIntQuant:
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=signed, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        ParameterFromRuntimeStatsScaling(collect_stats_steps, AbsPercentile(99.999, None), OverTensorView(), (),
                                         _pot(), 0.1, SCALING_MIN_VAL),
        PowerOfTwoIntScaling(signed=signed), ZeroZeroPoint(), BitWidthConst(bit_width))


def Int8ActPerTensorFixedPoint(collect_stats_steps: int = 300, bit_width: int = 8) -> RescalingIntQuant:
    """IntQuant + ParamFromRuntimePercentileScaling + PerTensorPoTScaling8bit (B/quant/fixed_point.py:37-47):
    log2 of the 99.999th percentile is collected, then learned; the scale is 2^ceil(value) / 2^(b-1)"""
    return _act_fixed_point(True, collect_stats_steps, bit_width)


def Uint8ActPerTensorFixedPoint(collect_stats_steps: int = 300, bit_width: int = 8) -> RescalingIntQuant:
    """unsigned variant (B/quant/fixed_point.py:50-60)"""
    return _act_fixed_point(False, collect_stats_steps, bit_width)


def Uint8ActPerTensorFixedPointMaxInit(max_val: float, bit_width: int = 8) -> RescalingIntQuant:
    """UintQuant + ParamMinMaxInitScaling + PerTensorPoTScaling8bit (B/quant/fixed_point.py:63-76): learned
    radix point initialised from a user-defined max_val"""
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=False, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        ParameterScaling(max_val, None, _pot(), None),
        PowerOfTwoIntScaling(signed=False), ZeroZeroPoint(), BitWidthConst(bit_width))


# ---- bias quantizers (B/quant/scaled_int.py:64-132, fixed_point.py:79-90) ---------------------------------

def _int_bias(bit_width: int) -> PrescaledRestrictIntQuant:
    return PrescaledRestrictIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        BitWidthConst(bit_width))


def Int8Bias() -> PrescaledRestrictIntQuant:
    """IntBias with bit_width = 8 (B/quant/scaled_int.py:78-88): `q(bias, scale)` with the scale of the
    accumulator the bias is added to, typically quant_input_scale * quant_weight_scale (one per output channel)"""
    return _int_bias(8)


def Int16Bias() -> PrescaledRestrictIntQuant:
    return _int_bias(16)


def Int24Bias() -> PrescaledRestrictIntQuant:
    return _int_bias(24)


def Int32Bias() -> PrescaledRestrictIntQuant:
    return _int_bias(32)


def Int8BiasPerTensorFloatInternalScaling(bias: torch.nn.Parameter, bit_width: int = 8) -> RescalingIntQuant:
    """IntQuant + MaxStatsScaling + PerTensorFloatScaling8bit + BiasQuantSolver (B/quant/scaled_int.py:135-141):
    the bias quantized with a scale of its own, from its abs-max (requires no input scale)"""
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, [bias], FloatRestrictValue(), (),
                                  affine_rescaling=False, scaling_min_val=SCALING_MIN_VAL),
        IntScaling(signed=True, narrow_range=False), ZeroZeroPoint(), BitWidthConst(bit_width))


def Int8BiasPerTensorFixedPointInternalScaling(bias: torch.nn.Parameter, bit_width: int = 8) -> RescalingIntQuant:
    """IntQuant + MaxStatsScaling + PerTensorPoTScaling8bit + BiasQuantSolver (B/quant/fixed_point.py:79-90)"""
    return RescalingIntQuant(
        IntQuant(narrow_range=False, signed=True, float_to_int_impl=RoundSte(), tensor_clamp_impl=TensorClamp()),
        StatsFromParameterScaling(AbsMax(), OverTensorView(), 0, [bias], _pot(), (), affine_rescaling=False,
                                  scaling_min_val=SCALING_MIN_VAL),
        PowerOfTwoIntScaling(signed=True), ZeroZeroPoint(), BitWidthConst(bit_width))
