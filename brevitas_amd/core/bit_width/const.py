"""Constant bit width (B/core/bit_width/const.py:14-40)."""
import torch
from torch import Tensor

from brevitas_amd.core.utils import StatelessBuffer


class BitWidthConst(torch.nn.Module):
    """Returns the bit width as a 0-dim float tensor; not part of the state dict.

    The returned tensor carries `bvq_host_value`, the python int it was built from, so that the fused
    quantizer can derive its integer clamp bounds on the host without a device->host sync.
    """

    def __init__(self, bit_width: int) -> None:
        super().__init__()
        assert isinstance(bit_width, int)
        self.bit_width = StatelessBuffer(torch.tensor(float(bit_width)))
        self._host_value = bit_width

    def forward(self) -> Tensor:
        t = self.bit_width()
        t.bvq_host_value = self._host_value
        return t
