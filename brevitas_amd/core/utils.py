"""In-place update helpers and the buffer type the constant-valued modules are built on (drop-ins for
B/core/utils.py:14-63).  The in-place ops keep the reference's op sequence: every one of them rounds to the
buffer's dtype, and the running statistics are compared bit for bit."""
from typing import Optional

import torch

from brevitas_amd.core._state import TolerantLoad

VALUE_ATTR_NAME = 'value'


def inplace_tensor_add(tensor: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
    return tensor.add_(value)


def inplace_tensor_mul(tensor: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
    return tensor.mul_(value)


def inplace_momentum_update(tensor: torch.Tensor, update: torch.Tensor, momentum: Optional[float],
                            counter: int, new_counter: int) -> torch.Tensor:
    """tensor <- mean of the `new_counter` updates seen so far (momentum None), or the exponential moving
    average tensor * (1 - momentum) + momentum * update; two in-place ops either way (B/core/utils.py:27-38)"""
    if momentum is None:
        keep, contribution = counter / new_counter, update / new_counter
    else:
        keep, contribution = 1 - momentum, momentum * update
    tensor.mul_(keep)
    tensor.add_(contribution)
    return tensor


class StatelessBuffer(TolerantLoad, torch.nn.Module):
    """A constant that moves with .to() / .cuda() like a buffer but is not part of the checkpoint: it is neither
    written to a state dict nor expected in one (B/core/utils.py:41-63).  forward() returns it detached."""

    bvq_never_saved = (VALUE_ATTR_NAME,)

    def __init__(self, value: torch.Tensor):
        super().__init__()
        self.register_buffer(VALUE_ATTR_NAME, value)

    def forward(self):
        return self.value.detach()

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        entries = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        entries.pop(prefix + VALUE_ATTR_NAME, None)
        return entries
