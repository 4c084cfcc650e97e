"""Module wrappers of trivial tensor maps (B/core/function_wrapper/misc.py).  These only ever see
scale-shaped tensors (1..C elements) and stay plain torch ops."""
import torch


class Identity(torch.nn.Module):

    def forward(self, x: torch.Tensor):
        return x


class PowerOfTwo(torch.nn.Module):

    def forward(self, x: torch.Tensor):
        return 2.0 ** x


class LogTwo(torch.nn.Module):

    def forward(self, x: torch.Tensor):
        return torch.log2(x)


class InplaceLogTwo(torch.nn.Module):
    """not differentiable"""

    def forward(self, x: torch.Tensor):
        x.log2_()
        return x
