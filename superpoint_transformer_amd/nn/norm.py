"""Normalisations of the SPT operator surface on the HIP kernels.

Mirrors src/nn/norm.py: ``UnitSphereNorm`` (norm.py:53-138) and ``GraphNorm``
(torch_geometric.nn.norm.GraphNorm, re-exported at norm.py:5; parameters
``weight``, ``bias``, ``mean_scale`` keep their names so reference checkpoints
load)."""
import torch
from torch import nn

from .. import ops

__all__ = ["UnitSphereNorm", "GraphNorm", "INDEX_BASED_NORMS"]


class UnitSphereNorm(nn.Module):
    """Normalise the positions of same-segment nodes into a sphere of diameter
    1 (src/nn/norm.py:53-138); one fused HIP pass instead of three scatters,
    two gathers and the elementwise tail."""

    def __init__(self, log_diameter=False):
        super().__init__()
        self.log_diameter = log_diameter

    def forward(self, pos, idx, w=None, num_super=None):
        pos, diameter = ops.unit_sphere_norm(pos, idx, w=w, num_super=num_super)
        if self.log_diameter:
            diameter = torch.log(diameter + 1)
        return pos, diameter


class GraphNorm(nn.Module):
    """PyG GraphNorm: ``weight * (x - mean_scale*mean_g) / sqrt(var_g + eps) +
    bias`` per graph of ``batch``.  ``batch_size`` (number of graphs) may be
    given to avoid the ``batch.max()`` host sync."""

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels = in_channels
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(in_channels))
        self.bias = nn.Parameter(torch.empty(in_channels))
        self.mean_scale = nn.Parameter(torch.empty(in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)
        nn.init.ones_(self.mean_scale)

    def forward(self, x, batch=None, batch_size=None, act_slope=1.0):
        return ops.graph_norm(x, batch, self.weight, self.bias, self.mean_scale,
                              eps=self.eps, num_graphs=batch_size, act_slope=act_slope)

    def extra_repr(self):
        return f"{self.in_channels}"


INDEX_BASED_NORMS = (GraphNorm,)
