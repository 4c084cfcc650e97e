"""Zero-point implementations (drop-ins for B/core/zero_point.py:27-258).

ZeroZeroPoint serves the symmetric quantizers of the headline path; the statistics- and
parameter-based ones serve the asymmetric ("shifted") quantizers.  All of them work on scale-shaped
tensors (1 or C elements) except for the statistic itself, which is the library's streaming min/max
or percentile reduction.  State-dict behaviour (dropped `buffer`, collected value saved as `value`,
load-time jump past collection) follows the reference.
"""
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn import Module, Parameter

from brevitas_amd.core._state import CollectThenLearn, TolerantLoad
from brevitas_amd.core.stats import DEFAULT_MOMENTUM, SCALAR_SHAPE, _ParameterListStats
from brevitas_amd.core.utils import StatelessBuffer, inplace_tensor_add
from brevitas_amd.function.ops_ste import abs_binary_sign_grad

__all__ = ['ZeroZeroPoint', 'StatsFromParameterZeroPoint', 'ParameterFromRuntimeZeroPoint', 'ParameterZeroPoint']


class ZeroZeroPoint(torch.nn.Module):
    """constant 0. zero-point (symmetric quantization)"""

    def __init__(self) -> None:
        super().__init__()
        self.zero_point = StatelessBuffer(torch.tensor(0.0))

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        return self.zero_point()


class _ScaleShiftZeroPoint(torch.nn.Module):
    """float offset -> integer zero-point: zero_point / scale + min_int, optionally itself quantized
    (B/core/zero_point.py:38-54)"""

    def __init__(self, int_quant: Module, quantize_zero_point: bool) -> None:
        super().__init__()
        self.int_quant = int_quant
        self.quantize_zero_point = quantize_zero_point

    def forward(self, zero_point: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        min_int = self.int_quant.min_int(bit_width)
        if self.quantize_zero_point:
            return self.int_quant.to_int(scale, min_int, bit_width, zero_point)
        return zero_point / scale + min_int


class StatsFromParameterZeroPoint(torch.nn.Module):
    """zero-point from a statistic (e.g. NegativeMinOrZero) of the tracked weights (:57-83)"""

    def __init__(self, int_quant: Module, quantize_zero_point: bool, zero_point_stats_input_view_shape_impl: Module,
                 zero_point_stats_input_concat_dim: int, zero_point_stats_impl: Module,
                 zero_point_shape: Tuple[int, ...], tracked_parameter_list: List[torch.nn.Parameter]) -> None:
        super().__init__()
        self.parameter_list_stats = _ParameterListStats(
            zero_point_stats_impl, zero_point_shape, zero_point_stats_input_view_shape_impl,
            zero_point_stats_input_concat_dim, tracked_parameter_list)
        self.scale_shift_zero_point = _ScaleShiftZeroPoint(int_quant, quantize_zero_point)

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> torch.Tensor:
        stats = self.parameter_list_stats()
        return self.scale_shift_zero_point(-stats, scale, bit_width)


class ParameterFromRuntimeZeroPoint(CollectThenLearn, torch.nn.Module):
    """a statistic of the activation (e.g. its low percentile) averaged over `collect_stats_steps` training steps,
    then learned as the parameter `value` (B/core/zero_point.py:86-183); the float offset goes through
    _ScaleShiftZeroPoint like every other zero-point"""

    def __init__(self, collect_stats_steps: int, int_quant: Module, quantize_zero_point: bool,
                 zero_point_stats_impl: Optional[Module], zero_point_shape: Tuple[int, ...],
                 zero_point_stats_input_view_shape_impl: Module,
                 zero_point_stats_momentum: Optional[float] = DEFAULT_MOMENTUM) -> None:
        super().__init__()
        self.bvq_init_collection(collect_stats_steps, zero_point_shape, 0.0, zero_point_stats_momentum)
        self.zero_point_shape = zero_point_shape
        self.stats_input_view_shape_impl = zero_point_stats_input_view_shape_impl
        self.zero_point_stats_impl = zero_point_stats_impl
        self.scale_shift_zero_point = _ScaleShiftZeroPoint(int_quant, quantize_zero_point)

    def training_forward(self, x) -> Tensor:
        if self.counter < self.collect_stats_steps:
            batch_stat = self.zero_point_stats_impl(self.stats_input_view_shape_impl(x)).view(self.zero_point_shape)
            self.bvq_fold(batch_stat.detach(), inplace_tensor_add)  # buffer starts at 0: the first fold is a sum
            return batch_stat + 0. * self.value  # `value` stays in the graph with a zero gradient (DDP)
        if self.counter == self.collect_stats_steps:
            inplace_tensor_add(self.value.detach(), self.buffer)  # hand over: parameter (0) += average
            self.counter += 1
        return self.value

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        if self.training:
            offset = self.training_forward(x)
        else:
            offset = self.buffer if self.counter <= self.collect_stats_steps else self.value
        return self.scale_shift_zero_point(abs_binary_sign_grad(offset), scale, bit_width)


class ParameterZeroPoint(TolerantLoad, torch.nn.Module):
    """learned zero-point (B/core/zero_point.py:186-258)"""

    bvq_float_checkpoint_ok = ('value',)

    def __init__(self, zero_point_init: Union[float, torch.Tensor], int_quant: Module, quantize_zero_point: bool,
                 zero_point_shape: Tuple[int, ...] = None) -> None:
        super().__init__()
        from brevitas_amd.core.scaling.standalone import _as_parameter_init
        init = _as_parameter_init(zero_point_init, zero_point_shape, 'zero_point_init')
        if zero_point_shape is not None and init.shape == SCALAR_SHAPE:
            init = torch.full(zero_point_shape, init)
        self.value = Parameter(init)
        self.scale_shift_zero_point = _ScaleShiftZeroPoint(int_quant, quantize_zero_point)

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        return self.scale_shift_zero_point(abs_binary_sign_grad(self.value), scale, bit_width)
