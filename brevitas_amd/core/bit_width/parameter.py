"""Learned bit widths (B/core/bit_width/parameter.py:22-146).  The bit width is a 0-dim tensor in the
autograd graph; the quantizers then take their integer bounds from tensors (min_int / max_int) and run
op by op on the HIP-backed elementwise ops, so the bit width receives its gradient through the scale's
integer threshold and through the clamp bounds, as in the reference."""
import torch
from torch import Tensor
from torch.nn import Module, Parameter

import brevitas_amd.config as config
from brevitas_amd.core.function_wrapper import RoundSte
from brevitas_amd.core.restrict_val import IntRestrictValue
from brevitas_amd.function.ops_ste import abs_binary_sign_grad

MIN_INT_BIT_WIDTH = 2
NON_ZERO_EPSILON = 1e-6
REMOVE_ZERO_BIT_WIDTH = 0.1


class BitWidthParameter(torch.nn.Module):
    """learnable bit width = restrict(|offset| + min_bit_width)

    Examples (B/core/bit_width/parameter.py:39-41):
        >>> BitWidthParameter(8)()
        tensor(8., grad_fn=...)
    """

    def __init__(self, bit_width: int, min_bit_width: int = MIN_INT_BIT_WIDTH,
                 restrict_bit_width_impl: Module = None, override_pretrained_bit_width: bool = False) -> None:
        super().__init__()
        if restrict_bit_width_impl is None:
            restrict_bit_width_impl = IntRestrictValue(RoundSte())
        if bit_width < MIN_INT_BIT_WIDTH:
            raise RuntimeError("Int bit width has to be at least {}, instead is {}.".format(
                MIN_INT_BIT_WIDTH, bit_width))
        if min_bit_width < MIN_INT_BIT_WIDTH:
            raise RuntimeError("Min int bit width has to be at least {}, instead is {}.".format(
                MIN_INT_BIT_WIDTH, min_bit_width))
        if bit_width < min_bit_width:
            raise RuntimeError("Int bit width has to be at least {}, instead is {}.".format(
                min_bit_width, bit_width))
        bit_width = float(int(bit_width))
        min_bit_width = float(int(min_bit_width))
        bit_width_base = restrict_bit_width_impl.restrict_init_float(min_bit_width)
        bit_width = restrict_bit_width_impl.restrict_init_float(bit_width)
        self.bit_width_offset = Parameter(torch.tensor(bit_width - bit_width_base))
        self.bit_width_base = bit_width_base
        self.restrict_bit_width_impl = restrict_bit_width_impl
        self.override_pretrained = override_pretrained_bit_width

    def forward(self) -> Tensor:
        bit_width = abs_binary_sign_grad(self.bit_width_offset) + self.bit_width_base
        return self.restrict_bit_width_impl(bit_width)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + 'bit_width_offset'
        if self.override_pretrained and key in state_dict:
            del state_dict[key]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        if config.IGNORE_MISSING_KEYS and key in missing_keys:
            missing_keys.remove(key)


class RemoveBitwidthParameter(torch.nn.Module):
    """learnable number of bits to drop = 1 / (eps + |coeff|)  (B/core/bit_width/parameter.py:104-146)"""

    def __init__(self, bit_width_to_remove: int, override_pretrained_bit_width: bool = False,
                 non_zero_epsilon: float = NON_ZERO_EPSILON, remove_zero_bit_width=REMOVE_ZERO_BIT_WIDTH):
        super().__init__()
        if bit_width_to_remove < 0:
            raise RuntimeError("Bit width to clamp has to be >= 0.".format(bit_width_to_remove))
        elif bit_width_to_remove == 0:
            bit_width_coeff_init = 1 / remove_zero_bit_width
        else:
            bit_width_coeff_init = 1 / bit_width_to_remove
        self.bit_width_coeff = Parameter(torch.tensor(bit_width_coeff_init))
        self.non_zero_epsilon = non_zero_epsilon
        self.override_pretrained = override_pretrained_bit_width

    def forward(self) -> Tensor:
        return 1.0 / (self.non_zero_epsilon + torch.abs(self.bit_width_coeff))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + 'bit_width_coeff'
        if self.override_pretrained and key in state_dict:
            del state_dict[key]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        if config.IGNORE_MISSING_KEYS and key in missing_keys:
            missing_keys.remove(key)
