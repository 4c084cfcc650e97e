"""The one proxy-level module on the hot path: activation function fused with its quantizer.

FusedActivationQuantProxy (drop-in for B/proxy/runtime_quant.py:73-84) applies `activation_impl`
and then `tensor_quant`.  When the activation is ReLU (QuantReLU, the dominant pattern) and the
quantizer is brevitas_amd's RescalingIntQuant, the activation is folded into the statistic and
quantizer kernels: its own read+write pass and its backward pass disappear.  Every other
combination runs the two modules one after the other, like the reference.
"""
import torch

from brevitas_amd import _native as nat
from brevitas_amd.core.function_wrapper import Identity

__all__ = ['FusedActivationQuantProxy']


def _pre_op_of(activation_impl):
    """bvq_pre_op of an activation module, or None if it has no fused form"""
    if activation_impl is None or isinstance(activation_impl, (Identity, torch.nn.Identity)):
        return nat.PRE_NONE
    if type(activation_impl) is torch.nn.ReLU:
        return nat.PRE_RELU
    return None


class FusedActivationQuantProxy(torch.nn.Module):

    def __init__(self, activation_impl, tensor_quant):
        super().__init__()
        self.activation_impl = activation_impl
        self.tensor_quant = tensor_quant

    def forward(self, x):
        pre_op = _pre_op_of(self.activation_impl)
        fused_forward = getattr(self.tensor_quant, 'bvq_forward_pre', None)
        if pre_op is not None and fused_forward is not None and x.is_cuda:
            return fused_forward(x, pre_op)
        x = self.activation_impl(x)
        x, output_scale, output_zp, output_bit_width = self.tensor_quant(x)
        return x, output_scale, output_zp, output_bit_width
