"""Normalisations of the SPT operator surface on the HIP kernels.

Mirrors src/nn/norm.py: ``UnitSphereNorm`` (norm.py:53-138) and the index-based norms it
re-exports from torch_geometric 2.3.0 (norm.py:5): ``GraphNorm`` (parameters ``weight``,
``bias``, ``mean_scale`` keep their names so reference checkpoints load), ``LayerNorm`` and
``InstanceNorm``."""
import torch
from torch import nn

from .. import ops
from ..csr import csr_of

__all__ = ["UnitSphereNorm", "GraphNorm", "LayerNorm", "InstanceNorm", "INDEX_BASED_NORMS"]


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


def _group_mean(x, csr):
    """Per-group mean of the rows of x, gathered back to the rows ([n, C]); segment kernels."""
    return ops.gather_rows(ops.segment_reduce(x, csr, None, "mean"), csr.idx)


class GraphNorm(nn.Module):
    """PyG GraphNorm: ``weight * (x - mean_scale*mean_g) / sqrt(var_g + eps) +
    bias`` per graph of ``batch``.  ``batch_size`` (number of graphs) may be
    given to avoid the ``batch.max()`` host sync.

    ``generic=True``: ``batch`` is ANY group index (unsorted, up to one group per row - what
    ``Data.norm_index`` hands out for ``norm_mode='node' | 'segment'``, src/data/data.py:103-130):
    the statistics then run on the segment-CSR kernels (mean / gather / mean) instead of the
    few-sorted-graphs kernel."""

    def __init__(self, in_channels, eps=1e-5):
        super().__init__()
        self.in_channels = in_channels
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(in_channels))
        self.bias = nn.Parameter(torch.empty(in_channels))
        self.mean_scale = nn.Parameter(torch.empty(in_channels))
        self.generic = False
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)
        nn.init.ones_(self.mean_scale)

    def forward(self, x, batch=None, batch_size=None, act_slope=1.0):
        if self.generic and batch is not None:
            csr = csr_of(batch, None)          # (batch_size counts clouds, not groups, here)
            out = x - _group_mean(x, csr) * self.mean_scale
            var = _group_mean(out * out, csr)
            out = self.weight * out / (var + self.eps).sqrt() + self.bias
            return out if act_slope == 1.0 else torch.nn.functional.leaky_relu(out, act_slope)
        return ops.graph_norm(x, batch, self.weight, self.bias, self.mean_scale,
                              eps=self.eps, num_graphs=batch_size, act_slope=act_slope)

    def extra_repr(self):
        return f"{self.in_channels}"


class LayerNorm(nn.Module):
    """torch_geometric.nn.norm.LayerNorm (2.3.0), imported by src/nn/norm.py:5: ``mode='graph'``
    normalises over all nodes AND channels of each graph, ``mode='node'`` is
    ``F.layer_norm`` over the channels of each row.  Group statistics on the segment kernels."""

    def __init__(self, in_channels, eps=1e-5, affine=True, mode="graph"):
        super().__init__()
        if mode not in ("graph", "node"):
            raise ValueError(f"Unknown normalization mode: {mode}")
        self.in_channels, self.eps, self.affine, self.mode = in_channels, eps, affine, mode
        if affine:
            self.weight = nn.Parameter(torch.empty(in_channels))
            self.bias = nn.Parameter(torch.empty(in_channels))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.ones_(self.weight)
            nn.init.zeros_(self.bias)

    def forward(self, x, batch=None, batch_size=None):
        if self.mode == "node":
            return torch.nn.functional.layer_norm(x, (self.in_channels,), self.weight, self.bias,
                                                  self.eps)
        if batch is None:
            x = x - x.mean()
            out = x / (x.std(unbiased=False) + self.eps)
        else:
            csr = csr_of(batch, batch_size)
            # mean over (rows of the graph) x channels = group mean of the row means
            mean = _group_mean(x.mean(dim=-1, keepdim=True), csr)
            x = x - mean
            var = _group_mean((x * x).mean(dim=-1, keepdim=True), csr)
            out = x / (var + self.eps).sqrt()
        if self.weight is not None:
            out = out * self.weight + self.bias
        return out

    def extra_repr(self):
        return f"{self.in_channels}, affine={self.affine}, mode={self.mode}"


class InstanceNorm(nn.Module):
    """torch_geometric.nn.norm.InstanceNorm (2.3.0): per graph and channel, biased variance;
    ``affine=False`` and no running statistics by default (the running-statistics variant is
    not built: no shipped config turns it on)."""

    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=False,
                 track_running_stats=False):
        super().__init__()
        if track_running_stats:
            raise NotImplementedError("InstanceNorm(track_running_stats=True) is not built")
        self.in_channels, self.eps, self.affine = in_channels, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(in_channels))
            self.bias = nn.Parameter(torch.zeros(in_channels))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x, batch=None, batch_size=None):
        if batch is None:
            mean = x.mean(dim=0, keepdim=True)
            out = x - mean
            var = (out * out).mean(dim=0, keepdim=True)
        else:
            csr = csr_of(batch, batch_size)
            out = x - _group_mean(x, csr)
            var = _group_mean(out * out, csr)
        out = out / (var + self.eps).sqrt()
        if self.weight is not None:
            out = out * self.weight + self.bias
        return out

    def extra_repr(self):
        return f"{self.in_channels}"


INDEX_BASED_NORMS = (LayerNorm, InstanceNorm, GraphNorm)   # src/nn/norm.py:140
