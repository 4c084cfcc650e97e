"""The pure-torch route for CPU tensors (brevitas_amd._aten; SURVEY 8b "Errors": a CPU-built layer or a CPU
evaluation pass must run through the same modules): the module-level parity tests of the device path, run
unchanged on CPU tensors against the same golden vectors produced by the reference -- the resolved graphs of
the named quantizers, IntQuant over every rounding / clamp / dtype / layout case, the STE ops with autograd,
the statistics, the shifted / fixed-point / sign / decoupled / truncating variants and learned bit widths.

The test bodies live in the tests/test_gpu_*.py modules (marked `gpu` there); here they are collected again with
the device constant of their module switched to 'cpu'.  No HIP kernel runs and nothing from oracle/ is involved
(`grep -r oracle brevitas_amd/` stays empty): the CPU route is the reference's own op composition on ATen.
"""
import pytest

import test_gpu_fixed_point as FP
import test_gpu_learned_bw as LB
import test_gpu_modules as M
import test_gpu_shifted as SH
import test_gpu_ste_stats_modules as SSM
import test_gpu_variants as V

_MODULES = (M, SSM, V, LB, SH, FP)


@pytest.fixture(autouse=True)
def on_cpu(monkeypatch):
    import brevitas_amd.config as config
    monkeypatch.setattr(config, 'SCALAR_OPERAND_MODE', 'cpu')
    for mod in _MODULES:
        monkeypatch.setattr(mod, 'DEV', 'cpu')


# resolved graphs of the named quantizers, IntQuant, scales, state dicts
from test_gpu_modules import (test_act_param_from_stats, test_act_parameter_scale, test_act_runtime_stats,  # noqa: E402,F401
                              test_const_scale_doctest, test_int_codes_emission,
                              test_int_quant_doctest_and_tensor_bit_width, test_int_quant_module_golden,
                              test_param_from_stats_state_dict_keys, test_qcdq_operand_package, test_weight_per_channel)
# the 12 STE ops with autograd, the plain ops' doctests, the statistics with autograd
from test_gpu_ste_stats_modules import (test_ops_ste_functions_with_autograd, test_plain_ops_doctests,  # noqa: E402,F401
                                        test_stats_modules_with_autograd,
                                        test_parameter_list_stats_concatenates_tracked_weights)
# 8f-4 variants
from test_gpu_variants import test_decoupled, test_doctests, test_sign_quantizers, test_trunc  # noqa: E402,F401
from test_gpu_learned_bw import (test_act_learned_bit_width, test_bit_width_modules,  # noqa: E402,F401
                                 test_weight_learned_bit_width)
from test_gpu_shifted import test_shifted_act, test_shifted_weight  # noqa: E402,F401
from test_gpu_fixed_point import (test_log_domain_learned_scale, test_pot_act, test_pot_bias, test_pot_max_init,  # noqa: E402,F401
                                  test_pot_weight)


def test_layers_built_on_cpu_run_and_train():
    """QuantConv2d / QuantLinear constructed on the CPU (the reference's default workflow) forward and backward"""
    import torch

    import brevitas_amd.quant as Q
    from brevitas_amd.nn import QuantConv2d, QuantLinear
    torch.manual_seed(0)
    conv = QuantConv2d(4, 6, 3, padding=1, weight_quant=Q.Int8WeightPerChannelFloat,
                       input_quant=Q.Int8ActPerTensorFloat(scaling_impl_type='stats', scaling_stats_op='max'))
    x = torch.randn(2, 4, 8, 8, requires_grad=True)
    y = conv(x)
    y.sum().backward()
    assert y.shape == (2, 6, 8, 8) and bool(torch.isfinite(y).all())
    assert conv.weight.grad is not None and x.grad is not None
    lin = QuantLinear(16, 8, weight_quant=Q.Int4WeightPerChannelFloat,
                      input_quant=Q.Int8ActPerTensorFloat(collect_stats_steps=2))
    for _ in range(4):  # percentile collection (torch.kthvalue on CPU), then the learned scale
        out = lin(torch.randn(5, 16))
        out.sum().backward()
    assert bool(torch.isfinite(out).all()) and lin.weight.grad is not None
    # the same modules moved to a device tensor later take the HIP kernels: nothing is cached per device
    w_q = lin.quant_weight()[0]
    codes = torch.round(w_q / lin.quant_weight()[1])
    assert float(codes.abs().max()) <= 7.0
