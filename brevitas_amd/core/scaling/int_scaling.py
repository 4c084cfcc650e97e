"""Integer threshold a float threshold is divided by to obtain the scale
(B/core/scaling/int_scaling.py:11-37)."""
import torch
from torch import Tensor

from brevitas_amd.function.ops import int_range_host, max_int, min_int


class IntScaling(torch.nn.Module):

    def __init__(self, signed: bool, narrow_range: bool):
        super().__init__()
        self.signed = signed
        self.narrow_range = narrow_range
        self._cache = {}

    def forward(self, bit_width: Tensor) -> Tensor:
        bw = getattr(bit_width, 'bvq_host_value', None)
        if bw is not None and not bit_width.requires_grad:
            # same value as the tensor arithmetic below, from a per-device cache: no kernel launches
            key = (bit_width.device, bit_width.dtype, bw)
            cached = self._cache.get(key)
            if cached is None:
                cached = torch.tensor(self.host_value(bw), dtype=bit_width.dtype, device=bit_width.device)
                self._cache[key] = cached
            return cached
        if self.signed:
            return -min_int(self.signed, self.narrow_range, bit_width)
        return max_int(self.signed, self.narrow_range, bit_width)

    def host_value(self, bit_width: int) -> float:
        """same number for a host-known bit width, without touching the device"""
        lo, hi = int_range_host(self.signed, self.narrow_range, bit_width)
        return -lo if self.signed else hi


class PowerOfTwoIntScaling(torch.nn.Module):

    def __init__(self, signed: bool):
        super().__init__()
        self.signed = signed

    def forward(self, bit_width: Tensor) -> Tensor:
        return max_int(self.signed, False, bit_width) + 1
