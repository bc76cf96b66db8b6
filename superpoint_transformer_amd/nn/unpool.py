"""Parent -> child broadcast (src/nn/unpool.py:7-13): HIP row gather whose
backward is a segment sum over the CSR view of the index."""
from torch import nn

from .. import ops

__all__ = ["IndexUnpool"]


class IndexUnpool(nn.Module):
    def forward(self, x, idx):
        return ops.gather_rows(x, idx)
