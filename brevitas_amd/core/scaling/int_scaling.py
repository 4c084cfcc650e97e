"""Integer threshold a float threshold is divided by to obtain the scale
(B/core/scaling/int_scaling.py:11-37)."""
import torch
from torch import Tensor

from brevitas_amd.function.ops import int_range_host, max_int, min_int


class _HostCachedIntScaling(torch.nn.Module):
    """for a host-known bit width (BitWidthConst) the threshold is a cached per-device constant:
    same value as the tensor arithmetic, no kernel launches"""

    def __init__(self):
        super().__init__()
        self._cache = {}

    def _cached(self, bit_width: Tensor):
        bw = getattr(bit_width, 'bvq_host_value', None)
        if bw is None or bit_width.requires_grad:
            return None
        key = (bit_width.device, bit_width.dtype, bw)
        cached = self._cache.get(key)
        if cached is None:
            cached = torch.tensor(self.host_value(bw), dtype=bit_width.dtype, device=bit_width.device)
            self._cache[key] = cached
        return cached


class IntScaling(_HostCachedIntScaling):

    def __init__(self, signed: bool, narrow_range: bool):
        super().__init__()
        self.signed = signed
        self.narrow_range = narrow_range

    def forward(self, bit_width: Tensor) -> Tensor:
        cached = self._cached(bit_width)
        if cached is not None:
            return cached
        if self.signed:
            return -min_int(self.signed, self.narrow_range, bit_width)
        return max_int(self.signed, self.narrow_range, bit_width)

    def host_value(self, bit_width: int) -> float:
        """same number for a host-known bit width, without touching the device"""
        lo, hi = int_range_host(self.signed, self.narrow_range, bit_width)
        return -lo if self.signed else hi


class PowerOfTwoIntScaling(_HostCachedIntScaling):
    """2^(b-1) (signed) or 2^b (unsigned): keeps threshold / int_threshold a power of two when the
    threshold is one (B/core/scaling/int_scaling.py:28-37)"""

    def __init__(self, signed: bool):
        super().__init__()
        self.signed = signed

    def forward(self, bit_width: Tensor) -> Tensor:
        cached = self._cached(bit_width)
        if cached is not None:
            return cached
        return max_int(self.signed, False, bit_width) + 1

    def host_value(self, bit_width: int) -> float:
        return int_range_host(self.signed, False, bit_width)[1] + 1
