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

import brevitas_amd.config as config
from brevitas_amd.core.stats import DEFAULT_MOMENTUM, SCALAR_SHAPE, _ParameterListStats
from brevitas_amd.core.utils import StatelessBuffer, inplace_momentum_update, inplace_tensor_add
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


class ParameterFromRuntimeZeroPoint(torch.nn.Module):
    """collect a statistic of the activation for `collect_stats_steps` training steps, then learn the
    zero-point as a parameter (:86-183)"""

    def __init__(self, collect_stats_steps: int, int_quant: Module, quantize_zero_point: bool,
                 zero_point_stats_impl: Optional[Module], zero_point_shape: Tuple[int, ...],
                 zero_point_stats_input_view_shape_impl: Module,
                 zero_point_stats_momentum: Optional[float] = DEFAULT_MOMENTUM) -> None:
        super().__init__()
        assert collect_stats_steps > 0, 'Steps should be more than 0'
        self.collect_stats_steps = collect_stats_steps
        self.counter = 0
        self.zero_point_shape = zero_point_shape
        self.stats_input_view_shape_impl = zero_point_stats_input_view_shape_impl
        self.momentum = zero_point_stats_momentum
        self.value = Parameter(torch.full(zero_point_shape, 0.0))
        self.register_buffer('buffer', torch.full(zero_point_shape, 0.0))
        self.zero_point_stats_impl = zero_point_stats_impl
        self.scale_shift_zero_point = _ScaleShiftZeroPoint(int_quant, quantize_zero_point)

    def training_forward(self, x) -> Tensor:
        if self.counter < self.collect_stats_steps:
            stats = self.zero_point_stats_impl(self.stats_input_view_shape_impl(x))
            stats = stats.view(self.zero_point_shape)
            new_counter = self.counter + 1
            if self.counter == 0:
                inplace_tensor_add(self.buffer, stats.detach())
            else:
                inplace_momentum_update(self.buffer, stats.detach(), self.momentum, self.counter, new_counter)
            self.counter = new_counter
            return stats + 0. * self.value  # keeps `value` in the graph with a zero gradient (DDP)
        if self.counter == self.collect_stats_steps:
            inplace_tensor_add(self.value.detach(), self.buffer)
            self.counter = self.counter + 1
        return self.value

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        if self.training:
            out = self.training_forward(x)
        elif self.counter <= self.collect_stats_steps:
            out = self.buffer
        else:
            out = self.value
        out = abs_binary_sign_grad(out)
        return self.scale_shift_zero_point(out, scale, bit_width)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        out = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        del out[prefix + 'buffer']
        if self.counter == 0:
            del out[prefix + 'value']
        elif self.counter <= self.collect_stats_steps:
            out[prefix + 'value'] = self.buffer
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        value_key = prefix + 'value'
        missing_keys.remove(prefix + 'buffer')
        training_key = prefix + 'training'
        if training_key in missing_keys:
            missing_keys.remove(training_key)
        if value_key not in missing_keys:
            self.counter = self.collect_stats_steps + 1
        if config.IGNORE_MISSING_KEYS and value_key in missing_keys:
            missing_keys.remove(value_key)


class ParameterZeroPoint(torch.nn.Module):
    """learned zero-point (:186-258)"""

    def __init__(self, zero_point_init: Union[float, torch.Tensor], int_quant: Module, quantize_zero_point: bool,
                 zero_point_shape: Tuple[int, ...] = None) -> None:
        super().__init__()
        if (isinstance(zero_point_init, Tensor) and zero_point_shape is not None
                and zero_point_init.shape != SCALAR_SHAPE and zero_point_init.shape != zero_point_shape):
            raise RuntimeError("zero_point_init.shape is non-scalar and != from zero_point_shape.")
        zero_point_init = zero_point_init.detach() if isinstance(zero_point_init, Tensor) \
            else torch.tensor(zero_point_init)
        if zero_point_init.shape == SCALAR_SHAPE and zero_point_shape is not None:
            zero_point_init = torch.full(zero_point_shape, zero_point_init)
        self.value = Parameter(zero_point_init)
        self.scale_shift_zero_point = _ScaleShiftZeroPoint(int_quant, quantize_zero_point)

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        out = abs_binary_sign_grad(self.value)
        return self.scale_shift_zero_point(out, scale, bit_width)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        value_key = prefix + 'value'
        if config.IGNORE_MISSING_KEYS and value_key in missing_keys:
            missing_keys.remove(value_key)
