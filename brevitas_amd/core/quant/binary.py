"""Binary quantizers (drop-ins for B/core/quant/binary.py): sign(x) * scale through the HIP-backed
straight-through ops; same constructors, outputs and state."""
from typing import Tuple

import torch
from torch import Tensor
from torch.nn import Module

from brevitas_amd.core.bit_width import BitWidthConst
from brevitas_amd.core.function_wrapper import TensorClamp
from brevitas_amd.core.quant.delay import DelayWrapper
from brevitas_amd.core.utils import StatelessBuffer
from brevitas_amd.function.ops_ste import binary_sign_ste


class BinaryQuant(torch.nn.Module):
    """y = binary_sign_ste(x) * scale; bit width 1, zero-point 0 (weights)"""

    def __init__(self, scaling_impl: Module, quant_delay_steps: int = 0):
        super().__init__()
        self.scaling_impl = scaling_impl
        self.bit_width = BitWidthConst(1)
        self.zero_point = StatelessBuffer(torch.tensor(0.0))
        self.delay_wrapper = DelayWrapper(quant_delay_steps)

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        scale = self.scaling_impl(x)
        y = binary_sign_ste(x) * scale
        y = self.delay_wrapper(x, y)
        return y, scale, self.zero_point(), self.bit_width()


class ClampedBinaryQuant(torch.nn.Module):
    """as BinaryQuant, after clamping x to (-scale, scale): gradients outside that range are zeroed
    (activations)"""

    def __init__(self, scaling_impl: Module, tensor_clamp_impl: Module = TensorClamp(), quant_delay_steps: int = 0):
        super().__init__()
        self.scaling_impl = scaling_impl
        self.bit_width = BitWidthConst(1)
        self.zero_point = StatelessBuffer(torch.tensor(0.0))
        self.delay_wrapper = DelayWrapper(quant_delay_steps)
        self.tensor_clamp_impl = tensor_clamp_impl

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        scale = self.scaling_impl(x)
        y = self.tensor_clamp_impl(x, -scale, scale)
        y = binary_sign_ste(y) * scale
        y = self.delay_wrapper(x, y)
        return y, scale, self.zero_point(), self.bit_width()
