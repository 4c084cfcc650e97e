"""View-shape helpers -- mirror of B/function/shape.py."""
from typing import Tuple

from torch import Tensor

__all__ = ['over_tensor', 'over_output_channels', 'over_batch_over_tensor', 'over_batch_over_output_channels']


def over_tensor(x: Tensor) -> int:
    return -1


def over_output_channels(x: Tensor) -> Tuple[int, int]:
    return x.shape[0], -1


def over_batch_over_tensor(x: Tensor) -> Tuple[int, int]:
    return x.shape[0], -1


def over_batch_over_output_channels(x: Tensor):
    return x.shape[0], x.shape[1], -1
