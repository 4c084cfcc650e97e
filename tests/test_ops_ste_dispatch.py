"""Seam 1 contract, restated from the reference's tests/brevitas/function/test_ops_ste.py:47-123:
`ops_ste.<f>(*args)` calls `<prefix>.ops.autograd_ste_ops.<f>_impl(*args)` exactly once with the same
arguments and returns its result; here the prefix is brevitas_amd and the impl is HIP-backed.
No GPU needed: the backend symbol is mocked, as in the reference."""
from unittest import mock

import pytest
import torch

import brevitas_amd
from brevitas_amd.function import ops_ste

AUTOGRAD_OPS_PREFIX = 'brevitas_amd.ops.autograd_ste_ops.'

ELEMWISE = ['round_ste', 'ceil_ste', 'floor_ste', 'binary_sign_ste', 'ternary_sign_ste', 'round_to_zero_ste',
            'dpu_round_ste', 'abs_binary_sign_grad']


def test_backend_flags():
    """B/function/ops_ste.py:38-43: a loaded native backend selects the non-Python prefix"""
    assert brevitas_amd.NATIVE_STE_BACKEND_LOADED
    assert ops_ste.fn_prefix is brevitas_amd
    assert set(ops_ste.__all__) == {
        'round_ste', 'ceil_ste', 'floor_ste', 'tensor_clamp_ste', 'tensor_clamp_ste_', 'scalar_clamp_ste',
        'scalar_clamp_min_ste', 'binary_sign_ste', 'ternary_sign_ste', 'round_to_zero_ste', 'dpu_round_ste',
        'abs_binary_sign_grad'}


def test_namespace_has_the_twelve_impls():
    """names registered by B/csrc/autograd_ste_ops.cpp:258-271 / aliased by B/ops/autograd_ste_ops.py:385-431"""
    ns = brevitas_amd.ops.autograd_ste_ops
    for name in ['round_ste_impl', 'ceil_ste_impl', 'floor_ste_impl', 'binary_sign_ste_impl',
                 'ternary_sign_ste_impl', 'round_to_zero_ste_impl', 'dpu_round_ste_impl', 'tensor_clamp_ste_impl',
                 'tensor_clamp_ste_impl_', 'scalar_clamp_ste_impl', 'scalar_clamp_min_ste_impl',
                 'abs_binary_sign_grad_impl']:
        assert callable(getattr(ns, name)), name


@pytest.mark.parametrize('name', ELEMWISE)
def test_elemwise_dispatch(name):
    impl = name + '_impl' if name != 'abs_binary_sign_grad' else 'abs_binary_sign_grad_impl'
    x = torch.randn(3, 4)
    with mock.patch(AUTOGRAD_OPS_PREFIX + impl) as m:
        m.return_value = sentinel = torch.zeros(1)
        out = getattr(ops_ste, name)(x)
        m.assert_called_once_with(x)
        assert out is sentinel


@pytest.mark.parametrize('name,impl', [('tensor_clamp_ste', 'tensor_clamp_ste_impl'),
                                        ('tensor_clamp_ste_', 'tensor_clamp_ste_impl_')])
def test_tensor_clamp_dispatch(name, impl):
    x, lo, hi = torch.randn(5), torch.tensor(-1.0), torch.tensor(1.0)
    with mock.patch(AUTOGRAD_OPS_PREFIX + impl) as m:
        m.return_value = sentinel = torch.zeros(1)
        out = getattr(ops_ste, name)(x, lo, hi)
        m.assert_called_once_with(x, lo, hi)
        assert out is sentinel


def test_scalar_clamp_dispatch():
    x = torch.randn(5)
    with mock.patch(AUTOGRAD_OPS_PREFIX + 'scalar_clamp_ste_impl') as m:
        m.return_value = sentinel = torch.zeros(1)
        assert ops_ste.scalar_clamp_ste(x, -2.0, 3.0) is sentinel
        m.assert_called_once_with(x, -2.0, 3.0)
    with mock.patch(AUTOGRAD_OPS_PREFIX + 'scalar_clamp_min_ste_impl') as m:
        m.return_value = sentinel = torch.zeros(1)
        assert ops_ste.scalar_clamp_min_ste(x, 1e-10) is sentinel
        m.assert_called_once_with(x, 1e-10)


def test_tracing_emits_plain_ops():
    """under torch.jit tracing the wrappers fall back to the plain torch op (B/function/ops_ste.py:65-66)"""
    traced = torch.jit.trace(lambda t: ops_ste.floor_ste(ops_ste.round_ste(t) + 0.25), torch.randn(4),
                             check_trace=False)
    x = torch.tensor([0.4, 1.5, -2.5, 3.7])
    assert torch.equal(traced(x), torch.floor(torch.round(x) + 0.25))


def test_cpu_tensors_take_the_pure_torch_route_and_the_hip_library_refuses_them():
    """a CPU tensor runs the reference's own op composition on ATen (brevitas_amd._aten; SURVEY 8b "Errors"); the
    HIP library itself never touches host memory: its wrappers refuse a CPU tensor loudly"""
    from brevitas_amd import _aten, _native
    x = torch.tensor([0.5, 1.5, -2.5, 2.4], requires_grad=True)
    y = ops_ste.round_ste(x)
    assert torch.equal(y.detach(), torch.tensor([0., 2., -2., 2.]))  # half to even, like torch.round
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones(4))                         # straight-through
    assert _aten.for_tensor(x) is _aten
    with pytest.raises(_native.BvqError, match='only runs on a ROCm'):
        _native.unary(_native.OP_ROUND, torch.randn(4))
