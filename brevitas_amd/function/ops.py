"""Plain (non straight-through) quantization primitives -- mirror of B/function/ops.py:16-191.

Tensor math runs in libbvq.so (CPU tensors: the pure-torch route, brevitas_amd._aten); the integer-range
formulas are scalar arithmetic on the 0-dim bit-width tensor, exactly as in the reference.
"""
import torch
from torch import Tensor
from torch.autograd import Function

from .. import _aten
from .. import _native as nat

__all__ = ['binary_sign', 'round_to_zero', 'dpu_round', 'tensor_clamp', 'tensor_clamp_', 'identity', 'max_int',
           'min_int']


class _ZeroGradUnaryFn(Function):
    """piecewise-constant maps: the reference's op compositions have zero gradient everywhere"""

    @staticmethod
    def forward(ctx, x, op):
        return _aten.for_tensor(x).unary(op, x)

    @staticmethod
    def backward(ctx, grad_y):
        return torch.zeros_like(grad_y), None


class _TensorClampFn(Function):
    """tensor_clamp with the autograd of its two torch.where w.r.t. x (B/function/ops.py:98-100)"""

    @staticmethod
    def forward(ctx, x, min_val, max_val):
        ctx.save_for_backward(x, min_val, max_val)
        return _aten.for_tensor(x).tensor_clamp(x, min_val, max_val).reshape(x.shape)

    @staticmethod
    def backward(ctx, grad_y):
        x, min_val, max_val = ctx.saved_tensors
        return _aten.for_tensor(x).tensor_clamp_bwd(grad_y, x, min_val, max_val).reshape(x.shape), None, None


def binary_sign(x: Tensor) -> Tensor:
    """2-valued sign, +1 at 0 (B/function/ops.py:16-34)"""
    return _ZeroGradUnaryFn.apply(x, nat.OP_BINARY_SIGN) if x.requires_grad else _aten.for_tensor(x).unary(nat.OP_BINARY_SIGN, x)


def round_to_zero(x: Tensor) -> Tensor:
    """sign(x) * floor(|x|) (B/function/ops.py:37-53)"""
    return _ZeroGradUnaryFn.apply(x, nat.OP_ROUND_TO_ZERO) if x.requires_grad else _aten.for_tensor(x).unary(nat.OP_ROUND_TO_ZERO, x)


def dpu_round(x: Tensor) -> Tensor:
    """DPU rounding (B/function/ops.py:56-72)"""
    return _ZeroGradUnaryFn.apply(x, nat.OP_DPU_ROUND) if x.requires_grad else _aten.for_tensor(x).unary(nat.OP_DPU_ROUND, x)


def tensor_clamp(x: Tensor, min_val: Tensor, max_val: Tensor) -> Tensor:
    """Clamp with tensor bounds, differentiable w.r.t. x (B/function/ops.py:75-100).

    Gradients w.r.t. the bounds (learned bit-widths) are outside this engine's scope: in that case
    the same two torch.where run on the device as in the reference.
    """
    if min_val.requires_grad or max_val.requires_grad:
        out = torch.where(x > max_val, max_val.type_as(x), x)
        return torch.where(out < min_val, min_val.type_as(out), out)
    return _TensorClampFn.apply(x, min_val, max_val)


def tensor_clamp_(x: Tensor, min_val: Tensor, max_val: Tensor) -> Tensor:
    """In-place variant, not differentiable (B/function/ops.py:103-111)"""
    if not x.is_contiguous():
        raise nat.BvqError('tensor_clamp_: in-place clamp needs a contiguous tensor')
    _aten.for_tensor(x).tensor_clamp(x, min_val, max_val, out=x)
    return x


def identity(x: Tensor) -> Tensor:
    return x


def max_int(signed: bool, narrow_range: bool, bit_width: Tensor) -> Tensor:
    """Largest representable integer, as a tensor like bit_width (B/function/ops.py:132-161)"""
    if not signed and not narrow_range:
        return (2 ** bit_width) - 1
    if not signed and narrow_range:
        return (2 ** bit_width) - 2
    return (2 ** (bit_width - 1)) - 1


def min_int(signed: bool, narrow_range: bool, bit_width: Tensor) -> Tensor:
    """Smallest representable integer, as a tensor like bit_width (B/function/ops.py:164-191)"""
    if signed and narrow_range:
        return -(2 ** (bit_width - 1)) + 1
    if signed:
        return -(2 ** (bit_width - 1))
    return 0 * bit_width


def int_range_host(signed: bool, narrow_range: bool, bit_width: int):
    """(min_int, max_int) as python floats for a host-known bit width: no device work, no sync"""
    if signed:
        return float(-(2 ** (bit_width - 1)) + (1 if narrow_range else 0)), float(2 ** (bit_width - 1) - 1)
    return 0.0, float(2 ** bit_width - 1 - (1 if narrow_range else 0))
