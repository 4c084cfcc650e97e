"""Small stateful helpers -- mirror of B/core/utils.py."""
from typing import Optional

import torch

VALUE_ATTR_NAME = 'value'


def inplace_tensor_add(tensor: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
    tensor.add_(value)
    return tensor


def inplace_tensor_mul(tensor: torch.Tensor, value: torch.Tensor) -> torch.Tensor:
    tensor.mul_(value)
    return tensor


def inplace_momentum_update(tensor: torch.Tensor, update: torch.Tensor, momentum: Optional[float],
                            counter: int, new_counter: int) -> torch.Tensor:
    """running average (momentum None) or exponential moving average (B/core/utils.py:27-38)"""
    if momentum is None:
        tensor.mul_(counter / new_counter)
        tensor.add_(update / new_counter)
    else:
        tensor.mul_(1 - momentum)
        tensor.add_(momentum * update)
    return tensor


class StatelessBuffer(torch.nn.Module):
    """A buffer that follows .to()/.cuda() but is never written to or required from a state dict
    (B/core/utils.py:41-63)."""

    def __init__(self, value: torch.Tensor):
        super().__init__()
        self.register_buffer(VALUE_ATTR_NAME, value)

    def forward(self):
        return self.value.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        key = prefix + VALUE_ATTR_NAME
        if key in missing_keys:
            missing_keys.remove(key)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        out = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        out.pop(prefix + VALUE_ATTR_NAME, None)
        return out
