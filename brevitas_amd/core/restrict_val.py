"""Restrictions applied to a scale before it is used (B/core/restrict_val.py:22-111).
Only the float (identity) restriction is on the accelerated path; the wrappers keep the reference's
structure so state-dict keys and injected modules line up."""
from typing import Optional

import torch
from torch import Tensor
from torch.nn import Module

from brevitas_amd.core.function_wrapper import Identity, ScalarClampMinSte


class _RestrictClampValue(torch.nn.Module):

    def __init__(self, scaling_min_val: Optional[float], restrict_value_impl: Optional[Module]):
        super().__init__()
        if scaling_min_val is not None and scaling_min_val != 0:
            self.clamp_min_ste = ScalarClampMinSte(scaling_min_val)
        else:
            self.clamp_min_ste = Identity()
        self.restrict_value_impl = restrict_value_impl if restrict_value_impl is not None else Identity()

    def forward(self, x: Tensor):
        x = self.restrict_value_impl(x)
        return self.clamp_min_ste(x)


class _RestrictValue(torch.nn.Module):

    def __init__(self, restrict_value_impl: Optional[Module]):
        super().__init__()
        self.restrict_value_impl = restrict_value_impl if restrict_value_impl is not None else Identity()

    def forward(self, x: Tensor):
        return self.restrict_value_impl(x)


class _ClampValue(torch.nn.Module):

    def __init__(self, scaling_min_val: Optional[float]):
        super().__init__()
        if scaling_min_val is not None and scaling_min_val != 0:
            self.clamp_min_ste = ScalarClampMinSte(scaling_min_val)
        else:
            self.clamp_min_ste = Identity()
        self.min_val = scaling_min_val

    def forward(self, x: Tensor):
        return self.clamp_min_ste(x)


class FloatRestrictValue(torch.nn.Module):
    """no restriction: the scale is any positive float"""

    def restrict_init_float(self, x: float) -> float:
        return x

    def restrict_init_tensor(self, x: Tensor) -> Tensor:
        return x

    def restrict_init_module(self):
        return Identity()

    def restrict_init_inplace_module(self):
        return Identity()

    def forward(self, x: Tensor) -> Tensor:
        return x
