"""Module wrappers of the straight-through functions, so they can be injected as
float_to_int_impl / tensor_clamp_impl (B/core/function_wrapper/ops_ste.py:14-118).

`bvq_round_mode` / `bvq_clamp_ste` tell the fused quantizer kernel which variant a module stands
for; a module without them (user-defined) makes IntQuant fall back to calling it as a function.
"""
import torch

from brevitas_amd import _native as nat
from brevitas_amd.function.ops_ste import (ceil_ste, dpu_round_ste, floor_ste, round_ste, round_to_zero_ste,
                                           scalar_clamp_min_ste, tensor_clamp_ste, tensor_clamp_ste_)


class RoundSte(torch.nn.Module):
    bvq_round_mode = nat.ROUND

    def forward(self, x: torch.Tensor):
        return round_ste(x)


class FloorSte(torch.nn.Module):
    bvq_round_mode = nat.FLOOR

    def forward(self, x: torch.Tensor):
        return floor_ste(x)


class RoundToZeroSte(torch.nn.Module):
    bvq_round_mode = nat.ROUND_TO_ZERO

    def forward(self, x: torch.Tensor):
        return round_to_zero_ste(x)


class DPURoundSte(torch.nn.Module):
    bvq_round_mode = nat.DPU_ROUND

    def forward(self, x: torch.Tensor):
        return dpu_round_ste(x)


class CeilSte(torch.nn.Module):
    bvq_round_mode = nat.CEIL

    def forward(self, x: torch.Tensor):
        return ceil_ste(x)


class ScalarClampMinSte(torch.nn.Module):

    def __init__(self, min_val: float) -> None:
        super().__init__()
        self.min_val = min_val

    def forward(self, x: torch.Tensor):
        return scalar_clamp_min_ste(x, self.min_val)


class TensorClampSte(torch.nn.Module):
    bvq_clamp_ste = True

    def forward(self, x: torch.Tensor, min_val: torch.Tensor, max_val: torch.Tensor):
        return tensor_clamp_ste(x, min_val, max_val)


class InplaceTensorClampSte(torch.nn.Module):

    def forward(self, x: torch.Tensor, min_val: torch.Tensor, max_val: torch.Tensor):
        return tensor_clamp_ste_(x, min_val, max_val)
