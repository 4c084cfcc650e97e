"""Module wrappers of clamps (B/core/function_wrapper/clamp.py:16-76)."""
import torch
from torch import Tensor

from brevitas_amd.function.ops import tensor_clamp


class TensorClamp(torch.nn.Module):
    """tensor_clamp: gradient masked where clipped (the default tensor_clamp_impl of IntQuant)"""
    bvq_clamp_ste = False

    def forward(self, x: Tensor, min_val: Tensor, max_val: Tensor):
        return tensor_clamp(x, min_val=min_val, max_val=max_val)


class ScalarClamp(torch.nn.Module):

    def __init__(self, min_val, max_val) -> None:
        super().__init__()
        self.min_val = min_val
        self.max_val = max_val

    def forward(self, x: Tensor):
        return torch.clamp(x, min=self.min_val, max=self.max_val)


class ClampMin(torch.nn.Module):

    def __init__(self, min_val: float) -> None:
        super().__init__()
        self.min_val = min_val

    def forward(self, x: Tensor):
        return x.clamp_min(self.min_val)
