"""Straight-through-estimator functions -- mirror of B/function/ops_ste.py:46-370.

Each wrapper forwards to the same-named `<name>_impl` of the `autograd_ste_ops` namespace, which
here is brevitas_amd.ops.autograd_ste_ops (HIP-backed).  The lookup goes through the module at call
time, like the reference's `fn_prefix.ops.autograd_ste_ops.<name>_impl(...)` (ops_ste.py:67), so the
dispatch contract pinned by the reference's tests/brevitas/function/test_ops_ste.py holds.  Under
torch.jit tracing the plain op is emitted instead (ops_ste.py:65-66).
"""
import torch
from torch import Tensor

import brevitas_amd
from brevitas_amd.function.ops import binary_sign, dpu_round, round_to_zero, tensor_clamp, tensor_clamp_

__all__ = ['round_ste', 'ceil_ste', 'floor_ste', 'tensor_clamp_ste', 'tensor_clamp_ste_', 'scalar_clamp_ste',
           'scalar_clamp_min_ste', 'binary_sign_ste', 'ternary_sign_ste', 'round_to_zero_ste', 'dpu_round_ste',
           'abs_binary_sign_grad']

fn_prefix = brevitas_amd


def _tracing():
    return torch._C._get_tracing_state()


def round_ste(x: Tensor) -> Tensor:
    if _tracing():
        return torch.round(x)
    return fn_prefix.ops.autograd_ste_ops.round_ste_impl(x)


def ceil_ste(x: Tensor) -> Tensor:
    if _tracing():
        return torch.ceil(x)
    return fn_prefix.ops.autograd_ste_ops.ceil_ste_impl(x)


def floor_ste(x: Tensor) -> Tensor:
    if _tracing():
        return torch.floor(x)
    return fn_prefix.ops.autograd_ste_ops.floor_ste_impl(x)


def tensor_clamp_ste(x: Tensor, min_val: Tensor, max_val: Tensor) -> Tensor:
    if _tracing():
        return tensor_clamp(x, min_val, max_val)
    return fn_prefix.ops.autograd_ste_ops.tensor_clamp_ste_impl(x, min_val, max_val)


def tensor_clamp_ste_(x: Tensor, min_val: Tensor, max_val: Tensor) -> Tensor:
    if _tracing():
        return tensor_clamp_(x, min_val, max_val)
    return fn_prefix.ops.autograd_ste_ops.tensor_clamp_ste_impl_(x, min_val, max_val)


def scalar_clamp_ste(x: Tensor, min_val: float, max_val: float) -> Tensor:
    if _tracing():
        return torch.clamp(x, min_val, max_val)
    return fn_prefix.ops.autograd_ste_ops.scalar_clamp_ste_impl(x, min_val, max_val)


def scalar_clamp_min_ste(x: Tensor, min_val: float) -> Tensor:
    if _tracing():
        return torch.clamp_min(x, min_val)
    return fn_prefix.ops.autograd_ste_ops.scalar_clamp_min_ste_impl(x, min_val)


def binary_sign_ste(x: Tensor) -> Tensor:
    if _tracing():
        return binary_sign(x)
    return fn_prefix.ops.autograd_ste_ops.binary_sign_ste_impl(x)


def ternary_sign_ste(x: Tensor) -> Tensor:
    if _tracing():
        return torch.sign(x)
    return fn_prefix.ops.autograd_ste_ops.ternary_sign_ste_impl(x)


def round_to_zero_ste(x: Tensor) -> Tensor:
    if _tracing():
        return round_to_zero(x)
    return fn_prefix.ops.autograd_ste_ops.round_to_zero_ste_impl(x)


def dpu_round_ste(x: Tensor) -> Tensor:
    if _tracing():
        return dpu_round(x)
    return fn_prefix.ops.autograd_ste_ops.dpu_round_ste_impl(x)


def abs_binary_sign_grad(x: Tensor) -> Tensor:
    if _tracing():
        return torch.abs(x)
    return fn_prefix.ops.autograd_ste_ops.abs_binary_sign_grad_impl(x)
