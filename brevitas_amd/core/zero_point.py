"""Zero-point implementations on the accelerated path (B/core/zero_point.py:27-35)."""
import torch
from torch import Tensor

from brevitas_amd.core.utils import StatelessBuffer

__all__ = ['ZeroZeroPoint']


class ZeroZeroPoint(torch.nn.Module):
    """constant 0. zero-point (symmetric quantization)"""

    def __init__(self) -> None:
        super().__init__()
        self.zero_point = StatelessBuffer(torch.tensor(0.0))

    def forward(self, x: Tensor, scale: Tensor, bit_width: Tensor) -> Tensor:
        return self.zero_point()
