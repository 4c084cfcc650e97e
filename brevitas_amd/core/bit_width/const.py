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


class MsbClampBitWidth(torch.nn.Module):
    """Bit width of an accumulator after dropping most-significant bits (drop-in for
    B/core/bit_width/const.py:43-66): |input_bit_width - bits_to_remove|, kept inside
    [min_overall_bit_width, max_overall_bit_width] by a straight-through clamp, so that a learned
    `bits_to_remove` (RemoveBitwidthParameter) keeps receiving a gradient at the bounds."""

    def __init__(self, bit_width_to_remove_impl: torch.nn.Module, min_overall_bit_width: int,
                 max_overall_bit_width: int) -> None:
        super().__init__()
        self.bit_width_to_remove_impl = bit_width_to_remove_impl
        self.min_overall_bit_width = BitWidthConst(min_overall_bit_width)
        self.max_overall_bit_width = BitWidthConst(max_overall_bit_width)

    def forward(self, input_bit_width: Tensor) -> Tensor:
        from brevitas_amd.function.ops_ste import tensor_clamp_ste
        kept = torch.abs(input_bit_width - self.bit_width_to_remove_impl())
        lo, hi = self.min_overall_bit_width(), self.max_overall_bit_width()
        return tensor_clamp_ste(kept, lo, hi)
