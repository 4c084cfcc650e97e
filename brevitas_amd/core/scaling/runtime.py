"""Scales computed from statistics at run time (B/core/scaling/runtime.py:19-135)."""
from typing import List, Optional, Tuple

import torch
from torch.nn import Module, Parameter

from brevitas_amd.core._state import TolerantLoad
from brevitas_amd.core.function_wrapper import Identity
from brevitas_amd.core.restrict_val import _RestrictClampValue
from brevitas_amd.core.stats import DEFAULT_MOMENTUM, _ParameterListStats, _RuntimeStats
from brevitas_amd.function.ops_ste import abs_binary_sign_grad


class _AffineRescaling(TolerantLoad, torch.nn.Module):
    """learned affine map of the statistic, kept positive: |stat * affine_weight + affine_bias|"""

    bvq_float_checkpoint_ok = ('affine_weight', 'affine_bias')

    def __init__(self, scaling_shape):
        super().__init__()
        self.affine_weight = Parameter(torch.ones(scaling_shape))
        self.affine_bias = Parameter(torch.zeros(scaling_shape))

    def forward(self, x):
        return abs_binary_sign_grad(x * self.affine_weight + self.affine_bias)


class _StatsScaling(torch.nn.Module):
    """statistic -> threshold: restriction pre-processing, optional affine, restriction + lower bound"""

    def __init__(self, restrict_scaling_impl: Module, scaling_shape: Tuple[int, ...],
                 scaling_min_val: Optional[float] = None, affine_rescaling: bool = False) -> None:
        super().__init__()
        self.affine_rescaling = _AffineRescaling(scaling_shape) if affine_rescaling else Identity()
        self.restrict_clamp_scaling = _RestrictClampValue(scaling_min_val, restrict_scaling_impl)
        self.restrict_scaling_pre = restrict_scaling_impl.restrict_init_module()
        self.scaling_min_val = scaling_min_val

    def forward(self, stats: torch.Tensor) -> torch.Tensor:
        stats = self.restrict_scaling_pre(stats)
        stats = self.affine_rescaling(stats)
        return self.restrict_clamp_scaling(stats)

    def bvq_plain_min_val(self):
        """the lower bound if this module is exactly clamp_min_ste(stat, min_val) (float restriction,
        no affine) -- the form the fused quantizer path folds in; else None"""
        from brevitas_amd.core.restrict_val import FloatRestrictValue
        if not isinstance(self.affine_rescaling, Identity) or not isinstance(self.restrict_scaling_pre, Identity):
            return None
        if not isinstance(self.restrict_clamp_scaling.restrict_value_impl, (FloatRestrictValue, Identity)):
            return None
        return self.scaling_min_val if self.scaling_min_val else 0.0


class StatsFromParameterScaling(torch.nn.Module):
    """threshold from the statistic of the tracked weights; the forward argument is ignored"""

    def __init__(self, scaling_stats_impl: Module, scaling_stats_input_view_shape_impl: Module,
                 scaling_stats_input_concat_dim: int, tracked_parameter_list: List[torch.nn.Parameter],
                 restrict_scaling_impl: Module, scaling_shape: Tuple[int, ...], affine_rescaling: bool = False,
                 scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        self.parameter_list_stats = _ParameterListStats(
            scaling_stats_impl, scaling_shape, scaling_stats_input_view_shape_impl,
            scaling_stats_input_concat_dim, tracked_parameter_list)
        self.stats_scaling_impl = _StatsScaling(restrict_scaling_impl, scaling_shape, scaling_min_val,
                                                affine_rescaling)

    def forward(self, ignored: torch.Tensor) -> torch.Tensor:
        return self.stats_scaling_impl(self.parameter_list_stats())


class RuntimeStatsScaling(torch.nn.Module):
    """threshold from the statistic of the activation itself (training) or its running average (eval)"""

    def __init__(self, scaling_stats_impl: Module, scaling_stats_input_view_shape_impl: Module,
                 restrict_scaling_impl: Module, scaling_shape: Tuple[int, ...], affine_rescaling: bool,
                 scaling_stats_momentum: float = DEFAULT_MOMENTUM, scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        self.runtime_stats = _RuntimeStats(scaling_stats_impl, scaling_shape, scaling_stats_input_view_shape_impl,
                                           scaling_stats_momentum)
        self.stats_scaling_impl = _StatsScaling(restrict_scaling_impl, scaling_shape, scaling_min_val,
                                                affine_rescaling)

    def forward(self, x: torch.Tensor):
        return self.stats_scaling_impl(self.runtime_stats(x))
