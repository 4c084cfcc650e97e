"""Ternary quantizer (drop-in for B/core/quant/ternary.py:15-66): on a device tensor the dead-zone mask, the sign
and the scaling are ONE kernel, their autograd one more (include/bvq.h, bvq_variant_fwd / bvq_variant_bwd)."""
from typing import Tuple

import torch
from torch import Tensor
from torch.nn import Module

from brevitas_amd.core.bit_width import BitWidthConst
from brevitas_amd.core.quant.delay import DelayWrapper
from brevitas_amd.core.utils import StatelessBuffer
from brevitas_amd.function.ops_ste import ternary_sign_ste


class TernaryQuant(torch.nn.Module):
    """y = [|x| > threshold * scale] * ternary_sign_ste(x) * scale; bit width 2, zero-point 0

    Examples (B/core/quant/ternary.py:32-44):
        >>> out, scale, zero_point, bit_width = TernaryQuant(ConstScaling(1.0), 0.5)(torch.Tensor([0.04, -0.6, 3.3]))
        >>> out
        tensor([ 0., -1.,  1.])
    """

    def __init__(self, scaling_impl: Module, threshold: float, quant_delay_steps: int = None):
        super().__init__()
        self.scaling_impl = scaling_impl
        self.threshold = threshold
        self.bit_width = BitWidthConst(2)
        self.zero_point = StatelessBuffer(torch.tensor(0.0))
        self.delay_wrapper = DelayWrapper(quant_delay_steps)

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        scale = self.scaling_impl(x)
        from brevitas_amd import _native as nat
        from brevitas_amd.core.quant.binary import _sign_kernel
        # mask.float() * sign(x) * scale computes in float32 whatever x's dtype is (B/core/quant/ternary.py:64-66)
        fused = _sign_kernel(x, scale, nat.VAR_TERNARY, threshold=float(self.threshold), ct=torch.float32) \
            if scale.dtype != torch.float64 else None
        if fused is not None:
            return self.delay_wrapper(x, fused), scale, self.zero_point(), self.bit_width()
        # {-1, 0, +1}: zero inside the dead zone |x| <= threshold * scale, the sign outside (gradient passes through)
        outside = torch.abs(x) > self.threshold * scale
        y = self.delay_wrapper(x, outside.float() * ternary_sign_ste(x) * scale)
        return y, scale, self.zero_point(), self.bit_width()
