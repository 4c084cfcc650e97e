"""Restrictions applied to a scale before it is used (B/core/restrict_val.py:22-111).
The float (identity) restriction folds into the fused statistic -> scale epilogue; the log / integer /
power-of-two restrictions act on scale-shaped tensors (1..C elements) between the statistic kernel and
the quantizer kernel.  The wrappers keep the reference's structure so state-dict keys and injected
modules line up."""
import math
from typing import Optional

import torch
from torch import Tensor
from torch.nn import Module

from brevitas_amd.core.function_wrapper import (Identity, InplaceLogTwo, LogTwo, PowerOfTwo, RoundSte,
                                                ScalarClampMinSte)


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


class LogFloatRestrictValue(torch.nn.Module):
    """the learned / tracked quantity is log2 of the scale (B/core/restrict_val.py:103-125)"""

    def __init__(self):
        super().__init__()
        self.power_of_two = PowerOfTwo()

    def restrict_init_float(self, x: float) -> float:
        return math.log2(x)

    def restrict_init_tensor(self, x: Tensor) -> Tensor:
        return torch.log2(x)

    def restrict_init_module(self):
        return LogTwo()

    def restrict_init_inplace_module(self):
        return InplaceLogTwo()

    def forward(self, x: Tensor) -> Tensor:
        return self.power_of_two(x)


class IntRestrictValue(torch.nn.Module):
    """scale rounded to an integer (B/core/restrict_val.py:128-149)"""

    def __init__(self, restrict_value_float_to_int_impl: Optional[Module] = None):
        super().__init__()
        self.float_to_int_impl = restrict_value_float_to_int_impl if restrict_value_float_to_int_impl is not None \
            else RoundSte()

    def restrict_init_float(self, x: float) -> float:
        return x

    def restrict_init_tensor(self, x: Tensor) -> Tensor:
        return x

    def restrict_init_module(self):
        return Identity()

    def restrict_init_inplace_module(self):
        return Identity()

    def forward(self, x: Tensor) -> Tensor:
        return self.float_to_int_impl(x)


class PowerOfTwoRestrictValue(torch.nn.Module):
    """scale = 2^int(log2 threshold): fixed-point quantizers (B/core/restrict_val.py:152-175).  The
    integer cast is a straight-through op, so the gradient reaches the log-domain value unchanged."""

    def __init__(self, restrict_value_float_to_int_impl: Optional[Module] = None):
        super().__init__()
        self.float_to_int_impl = restrict_value_float_to_int_impl if restrict_value_float_to_int_impl is not None \
            else RoundSte()
        self.power_of_two = PowerOfTwo()

    def restrict_init_float(self, x: float) -> float:
        return math.log2(x)

    def restrict_init_tensor(self, x: Tensor) -> Tensor:
        return torch.log2(x)

    def restrict_init_module(self):
        return LogTwo()

    def restrict_init_inplace_module(self):
        return InplaceLogTwo()

    def forward(self, x: Tensor) -> Tensor:
        x = self.float_to_int_impl(x)
        return self.power_of_two(x)
