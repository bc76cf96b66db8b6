"""Stochastic depth (src/nn/dropout.py:7-39): every ROW of the residual branch
is kept with probability ``1 - drop_prob`` independently (one Bernoulli draw per
node, broadcast over the feature dims) and, with ``scale_by_keep``, the kept
rows are scaled by ``1 / keep``."""
from torch import nn

__all__ = ["DropPath"]


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:0.3f}"
