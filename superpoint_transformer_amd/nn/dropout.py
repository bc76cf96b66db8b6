"""Stochastic depth (src/nn/dropout.py): drop the whole residual branch."""
import torch
from torch import nn

__all__ = ["DropPath"]


class DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p <= 0:
            return x
        keep = 1.0 - self.p
        mask = torch.rand(1, device=x.device) < keep
        return x * mask.to(x.dtype) / keep
