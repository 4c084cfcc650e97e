"""Views that turn a tensor into the 1-D / [C, -1] input of a statistic
(B/core/function_wrapper/shape.py:19-118).

`bvq_channel_dim()` lets the fused statistics kernels skip the permuted contiguous copy: it names
the channel dimension of the ORIGINAL tensor that ends up as dim 0 of the view, so the kernel can
reduce x in place as [outer, channels, inner].
"""
from typing import Optional, Tuple

import torch

from brevitas_amd.function.shape import (over_batch_over_output_channels, over_batch_over_tensor,
                                         over_output_channels, over_tensor)

from .misc import Identity


class PermuteDims(torch.nn.Module):

    def __init__(self, permute_dims: Tuple[int, ...]) -> None:
        super().__init__()
        self.permute_dims = permute_dims

    def forward(self, x: torch.Tensor):
        return x.permute(*self.permute_dims).contiguous()


class OverTensorView(torch.nn.Module):

    def forward(self, x: torch.Tensor):
        return x.reshape(over_tensor(x))

    def bvq_channel_dim(self, ndim: int):
        return None  # whole-tensor statistic


class OverOutputChannelView(torch.nn.Module):

    def __init__(self, permute_dims: Optional[Tuple[int, ...]]) -> None:
        super().__init__()
        if permute_dims is not None:
            self.permute_impl = PermuteDims(permute_dims)
        else:
            self.permute_impl = Identity()

    def forward(self, x: torch.Tensor):
        y = self.permute_impl(x)
        return y.reshape(over_output_channels(y))

    def bvq_channel_dim(self, ndim: int):
        """dimension of the unpermuted tensor that becomes dim 0, if every other dimension keeps its
        relative order (then the [C, -1] view orders each channel's elements exactly like the
        in-place [outer, C, inner] traversal); otherwise -1 (permuted copy needed)."""
        if isinstance(self.permute_impl, Identity):
            return 0
        dims = [d % ndim for d in self.permute_impl.permute_dims]
        if len(dims) != ndim:
            return -1
        rest = dims[1:]
        return dims[0] if rest == sorted(rest) else -1


class OverBatchOverTensorView(torch.nn.Module):

    def forward(self, x: torch.Tensor):
        return x.reshape(over_batch_over_tensor(x))


class OverBatchOverOutputChannelView(torch.nn.Module):

    def forward(self, x: torch.Tensor):
        return x.reshape(over_batch_over_output_channels(x))


class StatsInputViewShapeImpl(object):
    OVER_TENSOR = OverTensorView
    OVER_OUTPUT_CHANNELS = OverOutputChannelView
    OVER_BATCH_OVER_TENSOR = OverBatchOverTensorView
    OVER_BATCH_OVER_OUTPUT_CHANNELS = OverBatchOverOutputChannelView
