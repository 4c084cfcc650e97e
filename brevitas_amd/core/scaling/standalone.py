"""Scales that do not depend on the current input, or only during an initial collection phase
(B/core/scaling/standalone.py:22-298).  Parameter names (`value`), the dropped `buffer` and the
load-time jump past collection are the reference's, so checkpoints interchange."""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn import Module, Parameter

from brevitas_amd.core._state import CollectThenLearn, TolerantLoad
from brevitas_amd.core.function_wrapper import Identity, OverBatchOverTensorView
from brevitas_amd.core.restrict_val import _ClampValue, _RestrictClampValue, _RestrictValue
from brevitas_amd.core.stats import DEFAULT_MOMENTUM, SCALAR_SHAPE, _Stats
from brevitas_amd.core.utils import StatelessBuffer, inplace_tensor_mul
from brevitas_amd.function.ops_ste import abs_binary_sign_grad


class ConstScaling(torch.nn.Module):
    """constant threshold (B/core/scaling/standalone.py:22-72)"""

    def __init__(self, scaling_init: Union[float, Tensor], restrict_scaling_impl: Optional[Module] = None,
                 scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        self.restrict_clamp_scaling = _RestrictClampValue(scaling_min_val, restrict_scaling_impl)
        if isinstance(scaling_init, Tensor):
            if restrict_scaling_impl is not None:
                scaling_init = restrict_scaling_impl.restrict_init_tensor(scaling_init)
            self.value = StatelessBuffer(scaling_init.detach())
        else:
            if restrict_scaling_impl is not None:
                scaling_init = restrict_scaling_impl.restrict_init_float(scaling_init)
            self.value = StatelessBuffer(torch.tensor(scaling_init))

    def forward(self, placeholder: Tensor) -> Tensor:
        return self.restrict_clamp_scaling(self.value())


def _as_parameter_init(init: Union[float, Tensor], shape, what: str) -> Tensor:
    """float or tensor initial value -> detached tensor; a scalar is broadcast to `shape` later by the caller"""
    if isinstance(init, Tensor):
        if shape is not None and init.shape != SCALAR_SHAPE and init.shape != shape:
            raise RuntimeError("%s.shape is non-scalar and != from %s." % (what, what.replace('_init', '_shape')))
        return init.detach()
    return torch.tensor(init)


def _is_float_restriction(m) -> bool:
    from brevitas_amd.core.restrict_val import FloatRestrictValue
    return type(m) in (Identity, FloatRestrictValue)


class ParameterScaling(TolerantLoad, torch.nn.Module):
    """learned threshold |clamp_min(restrict(value))| (B/core/scaling/standalone.py:75-152)"""

    bvq_float_checkpoint_ok = ('value',)

    def __init__(self, scaling_init: Union[float, Tensor], scaling_shape: Optional[Tuple[int, ...]] = None,
                 restrict_scaling_impl: Optional[Module] = None, scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        init = _as_parameter_init(scaling_init, scaling_shape, 'scaling_init')
        if restrict_scaling_impl is not None:
            init = restrict_scaling_impl.restrict_init_tensor(init)  # e.g. log2 for power-of-two scales
        if scaling_shape is not None and init.shape == SCALAR_SHAPE:
            init = torch.full(scaling_shape, init)
        self.value = Parameter(init)
        self.restrict_clamp_scaling = _RestrictClampValue(scaling_min_val, restrict_scaling_impl)

    def forward(self, placeholder: Tensor) -> Tensor:
        return abs_binary_sign_grad(self.restrict_clamp_scaling(self.value))

    def bvq_learned_scale(self):
        """(value, min_val) if the threshold is exactly |clamp_min_ste(value, min_val)| (float restriction): the form
        RescalingIntQuant folds into one launch; else None"""
        rc = self.restrict_clamp_scaling
        if not _is_float_restriction(rc.restrict_value_impl):
            return None
        return self.value, getattr(rc.clamp_min_ste, 'min_val', None)

    def _load_from_state_dict(self, state_dict, prefix, *hook_args):
        legacy = prefix + 'learned_value'  # the parameter's name in old checkpoints
        if legacy in state_dict:
            state_dict[prefix + 'value'] = state_dict.pop(legacy)
        super()._load_from_state_dict(state_dict, prefix, *hook_args)


class ParameterFromRuntimeStatsScaling(CollectThenLearn, torch.nn.Module):
    """Collect a statistic for `collect_stats_steps` training steps (running average in `buffer`), then turn it
    into the learned parameter `value` (B/core/scaling/standalone.py:155-298).  While collecting, the threshold
    is the clamped statistic of the current batch itself, unrestricted; afterwards
    |clamp_min(restrict(value))|, with `value` kept in the restriction's domain (log2 for power-of-two scales)."""

    def __init__(self, collect_stats_steps: int, scaling_stats_impl: Module,
                 scaling_stats_input_view_shape_impl: Module = OverBatchOverTensorView(),
                 scaling_shape: Tuple[int, ...] = SCALAR_SHAPE, restrict_scaling_impl: Optional[Module] = None,
                 scaling_stats_momentum: Optional[float] = DEFAULT_MOMENTUM,
                 scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        self.bvq_init_collection(collect_stats_steps, scaling_shape, 1.0, scaling_stats_momentum)
        self.stats_input_view_shape_impl = scaling_stats_input_view_shape_impl
        self.stats = _Stats(scaling_stats_impl, scaling_shape)
        self.restrict_scaling = _RestrictValue(restrict_scaling_impl)
        self.clamp_scaling = _ClampValue(scaling_min_val)
        restricted = restrict_scaling_impl is not None
        self.restrict_inplace_preprocess = restrict_scaling_impl.restrict_init_inplace_module() if restricted else Identity()
        self.restrict_preprocess = restrict_scaling_impl.restrict_init_module() if restricted else Identity()

    def bvq_collected(self) -> Tensor:
        return self.restrict_preprocess(self.buffer)

    def _learned(self, out: Tensor) -> Tensor:
        return abs_binary_sign_grad(self.clamp_scaling(self.restrict_scaling(out)))

    def training_forward(self, stats_input: Tensor) -> Tensor:
        if self.counter < self.collect_stats_steps:
            batch_stat = self.stats(self.stats_input_view_shape_impl(stats_input))
            # `+ 0 * value` keeps the parameter in the autograd graph with a zero gradient, for DDP without
            # find_unused_parameters (B/core/scaling/standalone.py:234-235)
            threshold = self.clamp_scaling(batch_stat + 0. * self.value)
            self.bvq_fold(threshold.detach(), inplace_tensor_mul)  # buffer starts at 1: the first fold is a product
            return abs_binary_sign_grad(threshold)
        if self.counter == self.collect_stats_steps:
            # hand over: the average, moved into the restriction's domain, becomes the parameter (which starts at 1)
            self.restrict_inplace_preprocess(self.buffer)
            inplace_tensor_mul(self.value.detach(), self.buffer)
            self.counter += 1
        return self._learned(self.value)

    def forward(self, stats_input: Tensor) -> Tensor:
        if self.training:
            return self.training_forward(stats_input)
        frozen = self.bvq_collected() if self.counter <= self.collect_stats_steps else self.value
        return self._learned(frozen)

    def bvq_learned_scale(self):
        """(value, min_val) once the collection phase is over and the threshold is |clamp_min_ste(value)| (float
        restriction); None while statistics are still collected or handed over"""
        if self.counter <= self.collect_stats_steps or not _is_float_restriction(self.restrict_scaling.restrict_value_impl):
            return None
        return self.value, getattr(self.clamp_scaling, 'min_val', None)
