"""Optional delay of quantization by a number of training steps (B/core/quant/delay.py:12-54)."""
from typing import Optional

import torch
from torch import Tensor


class _NoDelay(torch.nn.Module):

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        return y


class _DelayQuant(torch.nn.Module):

    def __init__(self, quant_delay_steps):
        super().__init__()
        self.quant_delay_steps = quant_delay_steps

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        if self.quant_delay_steps > 0:
            self.quant_delay_steps = self.quant_delay_steps - 1
            return x
        return y

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        training_key = prefix + 'training'
        if training_key in missing_keys:
            missing_keys.remove(training_key)


class DelayWrapper(torch.nn.Module):

    def __init__(self, quant_delay_steps: Optional[int]):
        super().__init__()
        if quant_delay_steps is None or quant_delay_steps <= 0:
            self.delay_impl = _NoDelay()
        else:
            self.delay_impl = _DelayQuant(quant_delay_steps)

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        return self.delay_impl(x, y)
