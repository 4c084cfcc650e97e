"""Plumbing between a statistic and a scale (B/core/stats/stats_wrapper.py:19-114): reshape to the
scaling shape, batch-norm style running average for activations, parameter tracking for weights.
State-dict keys (`running_stats`, ...) are the reference's."""
from typing import List, Tuple

import torch
from torch import Tensor, nn

from brevitas_amd.core._state import TolerantLoad

from .view_wrapper import _ViewCatParameterWrapper, _ViewParameterWrapper

DEFAULT_MOMENTUM = 0.1
SCALAR_SHAPE = ()


class _Stats(torch.nn.Module):

    def __init__(self, stats_impl: nn.Module, stats_output_shape: Tuple[int, ...]) -> None:
        super().__init__()
        self.stats_output_shape = stats_output_shape
        self.stats_impl = stats_impl

    def forward(self, input: Tensor) -> Tensor:
        stats = self.stats_impl(input)
        return stats.view(self.stats_output_shape)


class _RuntimeStats(TolerantLoad, torch.nn.Module):
    """training: statistic of the current batch, folded into `running_stats`; eval: the buffer"""

    bvq_float_checkpoint_ok = ('running_stats',)

    def __init__(self, stats_impl: nn.Module, stats_output_shape: Tuple[int, ...],
                 stats_input_view_shape_impl: nn.Module, stats_buffer_momentum: float = DEFAULT_MOMENTUM) -> None:
        super().__init__()
        self.first_batch = True
        self.stats_input_view_shape_impl = stats_input_view_shape_impl
        self.stats = _Stats(stats_impl, stats_output_shape)
        self.momentum = stats_buffer_momentum
        self.register_buffer('running_stats', torch.full(stats_output_shape, 1.0))

    def update_running_stats(self, out: Tensor) -> None:
        """fold one batch statistic into the buffer (B/core/stats/stats_wrapper.py:61-66)"""
        out = out.detach()
        if out.is_cuda and self.running_stats.is_cuda and self.running_stats.is_contiguous() \
                and out.shape == self.running_stats.shape:
            # the same three in-place ops as below, with the same rounding points, in one launch
            from brevitas_amd import _native as nat
            nat.running_stats_update(self.running_stats, out, self.momentum, self.first_batch)
            self.first_batch = False
            return
        if self.first_batch:
            self.running_stats *= out
            self.first_batch = False
        else:
            self.running_stats *= (1 - self.momentum)
            self.running_stats += self.momentum * out

    def forward(self, stats_input) -> Tensor:
        if not self.training:
            return self.running_stats
        batch_stat = self.stats(self.stats_input_view_shape_impl(stats_input))
        self.update_running_stats(batch_stat)
        return batch_stat


class _ParameterListStats(torch.nn.Module):
    """statistic of one tracked parameter, or of several concatenated along `stats_input_concat_dim`"""

    def __init__(self, stats_impl: nn.Module, stats_output_shape: Tuple[int, ...],
                 stats_input_view_shape_impl: nn.Module, stats_input_concat_dim: int,
                 tracked_parameter_list: List[torch.nn.Parameter]) -> None:
        super().__init__()
        head, others = tracked_parameter_list[0], tracked_parameter_list[1:]
        self.stats_input_concat_dim = stats_input_concat_dim
        self.first_tracked_param = _ViewParameterWrapper(head, stats_input_view_shape_impl)
        # None (not an empty list) when a single parameter is tracked: the fused weight path keys on it
        self.extra_tracked_params_list = torch.nn.ModuleList(
            _ViewCatParameterWrapper(w, stats_input_view_shape_impl, stats_input_concat_dim) for w in others
        ) if others else None
        self.stats = _Stats(stats_impl, stats_output_shape)

    def forward(self) -> torch.Tensor:
        viewed = self.first_tracked_param()
        for appended in (self.extra_tracked_params_list or ()):
            viewed = appended(viewed)  # concatenates its own parameter's view to what came before
        return self.stats(viewed)
