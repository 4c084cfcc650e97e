"""Thin quantized layers: the call sequence of QuantWBIOL.forward_impl (B/nn/quant_layer.py:302-365)
reduced to what configs 4 and 5 need -- quantize the input, re-quantize the weight (every forward, as
the reference does in training), run the float conv / linear on the dequantized tensors.  Brevitas'
own layers (proxies, QuantTensor, export) are out of scope; these exist so that the engine can be
driven end to end without them.
"""
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from brevitas_amd.core.quant import RescalingIntQuant

__all__ = ['QuantConv2d', 'QuantLinear', 'QuantIdentity']

WeightQuantFactory = Callable[[torch.nn.Parameter], RescalingIntQuant]


class _QuantWeightMixin:

    def _init_quant(self, weight_quant: Optional[WeightQuantFactory], input_quant: Optional[torch.nn.Module],
                    bias_quant=None):
        self.weight_quant = weight_quant(self.weight) if weight_quant is not None else None
        self.input_quant = input_quant
        # bias_quant: a module `q(bias, scale)` fed the accumulator's scale (Int8Bias ...), or a factory taking
        # the bias parameter for quantizers with a scale of their own (Int8BiasPerTensorFloatInternalScaling)
        if bias_quant is not None and self.bias is not None and not isinstance(bias_quant, torch.nn.Module):
            bias_quant = bias_quant(self.bias)
        self.bias_quant = bias_quant if self.bias is not None else None

    def _quant_all(self, x):
        """-> (x, w, bias) as the float op consumes them (B/nn/quant_layer.py:302-333)"""
        in_scale = None
        if self.input_quant is not None and not getattr(self, 'bvq_disable_input_quant', False):  # bias correction
            x, in_scale, _, _ = self.input_quant(x)
        w, w_scale, _, _ = self.quant_weight()
        bias = self.bias
        if self.bias_quant is not None and not getattr(self, 'bvq_disable_weight_quant', False):
            from brevitas_amd.core.quant.int import PrescaledRestrictIntQuant
            if isinstance(self.bias_quant, PrescaledRestrictIntQuant):
                if in_scale is None or w_scale is None:
                    raise RuntimeError('Input scale required')  # B/proxy/parameter_quant.py: requires_input_scale
                # the accumulator's scale, one per output channel (or one in all): quant_weight.scale * quant_input.scale
                out_scale = (w_scale.reshape(-1) * in_scale.reshape(-1)).reshape(-1)
                bias = self.bias_quant(self.bias, out_scale)[0]
            else:
                bias = self.bias_quant(self.bias)[0]
        return x, w, bias

    def quant_weight(self):
        """-> (dequantized weight, scale, zero_point, bit_width); identity if no weight quantizer"""
        if self.weight_quant is None or getattr(self, 'bvq_disable_weight_quant', False):  # calibration
            return self.weight, None, None, None
        return self.weight_quant(self.weight)

    def quant_input(self, x):
        return self.input_quant(x)[0] if self.input_quant is not None else x


class QuantConv2d(_QuantWeightMixin, torch.nn.Conv2d):
    """torch.nn.Conv2d with an input quantizer and a per-forward weight quantizer
    (B/nn/quant_conv.py:116-206).  `weight_quant` is a factory taking the layer's weight,
    e.g. brevitas_amd.quant.Int8WeightPerChannelFloat; `input_quant` a tensor_quant module."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, weight_quant: Optional[WeightQuantFactory] = None,
                 input_quant: Optional[torch.nn.Module] = None, bias_quant=None, device=None, dtype=None):
        torch.nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                 bias, device=device, dtype=dtype)
        self._init_quant(weight_quant, input_quant, bias_quant)
        if device is not None:
            self.to(device)  # the quantizers' buffers / parameters follow the layer

    def forward(self, x):
        x, w, bias = self._quant_all(x)
        return F.conv2d(x, w, bias, self.stride, self.padding, self.dilation, self.groups)


class QuantLinear(_QuantWeightMixin, torch.nn.Linear):
    """torch.nn.Linear counterpart (B/nn/quant_linear.py:22-73)"""

    def __init__(self, in_features, out_features, bias=True, weight_quant: Optional[WeightQuantFactory] = None,
                 input_quant: Optional[torch.nn.Module] = None, bias_quant=None, device=None, dtype=None):
        torch.nn.Linear.__init__(self, in_features, out_features, bias, device=device, dtype=dtype)
        self._init_quant(weight_quant, input_quant, bias_quant)
        if device is not None:
            self.to(device)

    def forward(self, x):
        x, w, bias = self._quant_all(x)
        return F.linear(x, w, bias)


class QuantIdentity(torch.nn.Module):
    """an activation quantizer as a layer (B/nn/quant_activation.py)"""

    def __init__(self, act_quant: torch.nn.Module):
        super().__init__()
        self.act_quant = act_quant

    def forward(self, x):
        return self.act_quant(x)[0]
