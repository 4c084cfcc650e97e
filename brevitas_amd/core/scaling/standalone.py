"""Scales that do not depend on the current input, or only during an initial collection phase
(B/core/scaling/standalone.py:22-298).  Parameter names (`value`), the dropped `buffer` and the
load-time jump past collection are the reference's, so checkpoints interchange."""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor
from torch.nn import Module, Parameter

import brevitas_amd.config as config
from brevitas_amd.core.function_wrapper import Identity, OverBatchOverTensorView
from brevitas_amd.core.restrict_val import _ClampValue, _RestrictClampValue, _RestrictValue
from brevitas_amd.core.stats import DEFAULT_MOMENTUM, SCALAR_SHAPE, _Stats
from brevitas_amd.core.utils import StatelessBuffer, inplace_momentum_update, inplace_tensor_mul
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


class ParameterScaling(torch.nn.Module):
    """learned threshold |clamp_min(value)| (B/core/scaling/standalone.py:75-152)"""

    def __init__(self, scaling_init: Union[float, Tensor], scaling_shape: Optional[Tuple[int, ...]] = None,
                 restrict_scaling_impl: Optional[Module] = None, scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        if (isinstance(scaling_init, Tensor) and scaling_shape is not None
                and scaling_init.shape != SCALAR_SHAPE and scaling_init.shape != scaling_shape):
            raise RuntimeError("scaling_init.shape is non-scalar and != from scaling_shape.")
        scaling_init = scaling_init.detach() if isinstance(scaling_init, Tensor) else torch.tensor(scaling_init)
        if restrict_scaling_impl is not None:
            scaling_init = restrict_scaling_impl.restrict_init_tensor(scaling_init)
        if scaling_init.shape == SCALAR_SHAPE and scaling_shape is not None:
            scaling_init = torch.full(scaling_shape, scaling_init)
        self.value = Parameter(scaling_init)
        self.restrict_clamp_scaling = _RestrictClampValue(scaling_min_val, restrict_scaling_impl)

    def forward(self, placeholder: Tensor) -> Tensor:
        return abs_binary_sign_grad(self.restrict_clamp_scaling(self.value))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        value_key = prefix + 'value'
        retrocomp_value_key = prefix + 'learned_value'
        if retrocomp_value_key in state_dict:
            state_dict[value_key] = state_dict.pop(retrocomp_value_key)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        if config.IGNORE_MISSING_KEYS and value_key in missing_keys:
            missing_keys.remove(value_key)


class ParameterFromRuntimeStatsScaling(torch.nn.Module):
    """Collect a statistic for `collect_stats_steps` training steps (running average in `buffer`),
    then turn it into the learned parameter `value` (B/core/scaling/standalone.py:155-298)."""

    def __init__(self, collect_stats_steps: int, scaling_stats_impl: Module,
                 scaling_stats_input_view_shape_impl: Module = OverBatchOverTensorView(),
                 scaling_shape: Tuple[int, ...] = SCALAR_SHAPE, restrict_scaling_impl: Optional[Module] = None,
                 scaling_stats_momentum: Optional[float] = DEFAULT_MOMENTUM,
                 scaling_min_val: Optional[float] = None) -> None:
        super().__init__()
        assert collect_stats_steps > 0, 'Steps should be more than 0'
        self.collect_stats_steps = collect_stats_steps
        self.counter = 0
        self.stats_input_view_shape_impl = scaling_stats_input_view_shape_impl
        self.stats = _Stats(scaling_stats_impl, scaling_shape)
        self.momentum = scaling_stats_momentum
        self.register_buffer('buffer', torch.full(scaling_shape, 1.0))
        self.value = Parameter(torch.full(scaling_shape, 1.0))
        self.restrict_scaling = _RestrictValue(restrict_scaling_impl)
        self.clamp_scaling = _ClampValue(scaling_min_val)
        if restrict_scaling_impl is not None:
            self.restrict_inplace_preprocess = restrict_scaling_impl.restrict_init_inplace_module()
            self.restrict_preprocess = restrict_scaling_impl.restrict_init_module()
        else:
            self.restrict_inplace_preprocess = Identity()
            self.restrict_preprocess = Identity()

    def _learned(self, out: Tensor) -> Tensor:
        return abs_binary_sign_grad(self.clamp_scaling(self.restrict_scaling(out)))

    def training_forward(self, stats_input: Tensor) -> Tensor:
        if self.counter < self.collect_stats_steps:
            stats = self.stats(self.stats_input_view_shape_impl(stats_input))
            # keeps `value` in the autograd graph with a zero gradient (DDP without
            # find_unused_parameters, B/core/scaling/standalone.py:234-235)
            stats = stats + 0. * self.value
            clamped_stats = self.clamp_scaling(stats)
            new_counter = self.counter + 1
            if self.counter == 0:
                inplace_tensor_mul(self.buffer, clamped_stats.detach())
            else:
                inplace_momentum_update(self.buffer, clamped_stats.detach(), self.momentum, self.counter,
                                        new_counter)
            self.counter = new_counter
            return abs_binary_sign_grad(clamped_stats)
        if self.counter == self.collect_stats_steps:
            self.restrict_inplace_preprocess(self.buffer)
            inplace_tensor_mul(self.value.detach(), self.buffer)
            self.counter = self.counter + 1
        return self._learned(self.value)

    def forward(self, stats_input: Tensor) -> Tensor:
        if self.training:
            return self.training_forward(stats_input)
        if self.counter <= self.collect_stats_steps:
            out = self.restrict_preprocess(self.buffer)
        else:
            out = self.value
        return self._learned(out)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        out = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        del out[prefix + 'buffer']  # never saved
        if self.counter == 0:
            del out[prefix + 'value']  # nothing collected yet: do not save the init value
        elif self.counter <= self.collect_stats_steps:
            out[prefix + 'value'] = self.restrict_preprocess(self.buffer)  # save what was collected so far
        return out

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        value_key = prefix + 'value'
        missing_keys.remove(prefix + 'buffer')  # always absent by design
        retrocomp_value_key = prefix + 'learned_value'
        if retrocomp_value_key in state_dict:
            state_dict[value_key] = state_dict.pop(retrocomp_value_key)
        training_key = prefix + 'training'
        if training_key in missing_keys:
            missing_keys.remove(training_key)
        if value_key not in missing_keys:
            self.counter = self.collect_stats_steps + 1  # a loaded value ends the collection phase
        if config.IGNORE_MISSING_KEYS and value_key in missing_keys:
            missing_keys.remove(value_key)
