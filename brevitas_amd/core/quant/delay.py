"""Quantization that switches on only after a number of training steps (drop-in for
B/core/quant/delay.py:12-54): until then the wrapper hands back the float input instead of the quantized
value.  The step counter is plain module state, as in the reference (it is not saved)."""
from typing import Optional

import torch
from torch import Tensor

from brevitas_amd.core._state import TolerantLoad


class _NoDelay(torch.nn.Module):
    """quantization is on from the first step"""

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        return y


class _DelayQuant(TolerantLoad, torch.nn.Module):
    """passes x through for the first `quant_delay_steps` calls, y afterwards"""

    def __init__(self, quant_delay_steps):
        super().__init__()
        self.quant_delay_steps = quant_delay_steps

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        remaining = self.quant_delay_steps
        if remaining <= 0:
            return y
        self.quant_delay_steps = remaining - 1
        return x


class DelayWrapper(torch.nn.Module):

    def __init__(self, quant_delay_steps: Optional[int]):
        super().__init__()
        delayed = quant_delay_steps is not None and quant_delay_steps > 0
        self.delay_impl = _DelayQuant(quant_delay_steps) if delayed else _NoDelay()

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        return self.delay_impl(x, y)
