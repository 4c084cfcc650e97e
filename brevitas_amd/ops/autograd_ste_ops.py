"""The `autograd_ste_ops` namespace, served by libbvq.so (seam 1).

Same 12 callables and the same argument meaning as the reference's two interchangeable backends
(Python: B/ops/autograd_ste_ops.py:385-431; C++: B/csrc/autograd_ste_ops.cpp:197-271): every forward
is one HIP elementwise kernel, every backward is the straight-through identity w.r.t. the first
argument (binary_sign(x) * grad for abs_binary_sign_grad).  A CPU tensor takes the pure-torch route
(brevitas_amd._aten: the reference's own op composition), as the reference's backends do.
"""
import torch
from torch.autograd import Function

from .. import _aten
from .. import _native as nat

__all__ = [
    'ScalarClampSteFn', 'ScalarClampMinSteFn', 'TensorClampSteFn', 'InplaceTensorClampSteFn',
    'RoundToZeroSteFn', 'CeilSteFn', 'FloorSteFn', 'BinarySignSteFn', 'TernarySignSteFn', 'RoundSteFn',
    'AbsBinarySignGradFn', 'DPURoundSteFn', 'round_ste_impl', 'binary_sign_ste_impl',
    'ternary_sign_ste_impl', 'floor_ste_impl', 'ceil_ste_impl', 'round_to_zero_ste_impl',
    'scalar_clamp_min_ste_impl', 'scalar_clamp_ste_impl', 'tensor_clamp_ste_impl', 'tensor_clamp_ste_impl_',
    'abs_binary_sign_grad_impl', 'dpu_round_ste_impl']


def _unary_ste(name, op, doc):
    def forward(ctx, x):
        return _aten.for_tensor(x).unary(op, x)

    def backward(ctx, grad_y):
        return grad_y

    return type(name, (Function,), {'forward': staticmethod(forward), 'backward': staticmethod(backward),
                                    '__doc__': doc})


RoundSteFn = _unary_ste('RoundSteFn', nat.OP_ROUND, 'torch.round forward, identity backward (B/ops/autograd_ste_ops.py:329-353)')
CeilSteFn = _unary_ste('CeilSteFn', nat.OP_CEIL, 'torch.ceil forward, identity backward (:219-242)')
FloorSteFn = _unary_ste('FloorSteFn', nat.OP_FLOOR, 'torch.floor forward, identity backward (:245-268)')
RoundToZeroSteFn = _unary_ste('RoundToZeroSteFn', nat.OP_ROUND_TO_ZERO, 'round_to_zero forward, identity backward (:161-187)')
DPURoundSteFn = _unary_ste('DPURoundSteFn', nat.OP_DPU_ROUND, 'dpu_round forward, identity backward (:190-216)')
BinarySignSteFn = _unary_ste('BinarySignSteFn', nat.OP_BINARY_SIGN, 'binary_sign forward, identity backward (:271-300)')
TernarySignSteFn = _unary_ste('TernarySignSteFn', nat.OP_TERNARY_SIGN, 'torch.sign forward, identity backward (:303-326)')


class ScalarClampSteFn(Function):
    """torch.clamp(x, min_val, max_val) forward, identity backward (B/ops/autograd_ste_ops.py:37-66)"""

    @staticmethod
    def forward(ctx, x, min_val, max_val):
        return _aten.for_tensor(x).scalar_clamp(x, min_val, max_val)

    @staticmethod
    def backward(ctx, grad_y):
        return grad_y, None, None


class ScalarClampMinSteFn(Function):
    """torch.clamp_min(x, min_val) forward, identity backward (B/ops/autograd_ste_ops.py:69-97)"""

    @staticmethod
    def forward(ctx, x, min_val):
        return _aten.for_tensor(x).scalar_clamp(x, min_val, None)

    @staticmethod
    def backward(ctx, grad_y):
        return grad_y, None


class TensorClampSteFn(Function):
    """tensor_clamp forward, gradient to x only (B/ops/autograd_ste_ops.py:100-128)"""

    @staticmethod
    def forward(ctx, x, min_val, max_val):
        return _aten.for_tensor(x).tensor_clamp(x, min_val, max_val)

    @staticmethod
    def backward(ctx, grad_y):
        return grad_y, None, None


class InplaceTensorClampSteFn(Function):
    """in-place tensor_clamp_ forward (x is overwritten), gradient to x only (:131-158)"""

    @staticmethod
    def forward(ctx, x, min_val, max_val):
        if not x.is_contiguous():
            raise nat.BvqError('tensor_clamp_ste_: in-place clamp needs a contiguous tensor')
        _aten.for_tensor(x).tensor_clamp(x, min_val, max_val, out=x)
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, grad_y):
        return grad_y, None, None


class AbsBinarySignGradFn(Function):
    """torch.abs forward; backward binary_sign(x) * grad, i.e. subgradient 1 at 0 (:356-382)"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _aten.for_tensor(x).unary(nat.OP_ABS, x)

    @staticmethod
    def backward(ctx, grad_y):
        x, = ctx.saved_tensors
        return _aten.for_tensor(x).abs_binary_sign_grad_bwd(grad_y, x).reshape(x.shape)


round_ste_impl = RoundSteFn.apply
binary_sign_ste_impl = BinarySignSteFn.apply
ternary_sign_ste_impl = TernarySignSteFn.apply
floor_ste_impl = FloorSteFn.apply
ceil_ste_impl = CeilSteFn.apply
round_to_zero_ste_impl = RoundToZeroSteFn.apply
dpu_round_ste_impl = DPURoundSteFn.apply
scalar_clamp_min_ste_impl = ScalarClampMinSteFn.apply
scalar_clamp_ste_impl = ScalarClampSteFn.apply
tensor_clamp_ste_impl = TensorClampSteFn.apply
tensor_clamp_ste_impl_ = InplaceTensorClampSteFn.apply
abs_binary_sign_grad_impl = AbsBinarySignGradFn.apply
