"""Wrappers that view a tracked parameter as a statistics input without owning it
(B/core/stats/view_wrapper.py:13-67): the aliased weight is neither saved nor required on load."""
import torch
from torch import Tensor
from torch.nn import Module, Parameter


class _AliasedParameterMixin:

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        key = prefix + 'parameter'
        if key in missing_keys:
            missing_keys.remove(key)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        out = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        out.pop(prefix + 'parameter', None)
        return out


class _ViewParameterWrapper(_AliasedParameterMixin, torch.nn.Module):

    def __init__(self, parameter: Parameter, view_shape_impl: Module) -> None:
        super().__init__()
        self.parameter = parameter
        self.view_shape_impl = view_shape_impl

    def forward(self) -> Tensor:
        return self.view_shape_impl(self.parameter)


class _ViewCatParameterWrapper(_AliasedParameterMixin, torch.nn.Module):

    def __init__(self, parameter: Parameter, view_shape_impl: Module, cat_dim: int) -> None:
        super().__init__()
        self.parameter = parameter
        self.view_shape_impl = view_shape_impl
        self.cat_dim = cat_dim

    def forward(self, x: Tensor) -> Tensor:
        return torch.cat([self.view_shape_impl(self.parameter), x], dim=self.cat_dim)
