"""The ``torch_geometric`` symbols the reference's hot path imports
(src/nn/attention.py:4, pool.py:3-9, norm.py:5-7, utils/edge.py:1), on the HIP ops."""
import torch
from torch import nn

from .. import ops
from ..csr import csr_of
# same parameters as PyG's (GraphNorm: weight, bias, mean_scale; LayerNorm / InstanceNorm: weight, bias)
from ..nn.norm import GraphNorm, InstanceNorm, LayerNorm  # noqa: F401
from . import scatter_shim


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    """torch_geometric.utils.softmax: per-group softmax along dim 0."""
    if ptr is not None or dim != 0:
        raise NotImplementedError("HIP shim: softmax(src, index, dim=0) only")
    csr = csr_of(index, num_nodes)
    mx = ops.segment_reduce(src.detach(), csr, None, "max")
    e = (src - ops.gather_rows(mx, csr.idx)).exp()
    z = ops.segment_reduce(e, csr, None, "sum") + 1e-16
    return e / ops.gather_rows(z, csr.idx)


def degree(index, num_nodes=None, dtype=None):
    n = int(num_nodes) if num_nodes is not None else int(index.max()) + 1
    return csr_of(index, n).counts().to(dtype or torch.float)


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    return scatter_shim.scatter(src, index, dim, None, dim_size, reduce)


class _Aggregation(nn.Module):
    reduce = None

    def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
        return scatter_shim.scatter(x, index, 0, None, dim_size, self.reduce)


class SumAggregation(_Aggregation):
    reduce = "sum"


class MeanAggregation(_Aggregation):
    reduce = "mean"


class MaxAggregation(_Aggregation):
    reduce = "max"


class MinAggregation(_Aggregation):
    reduce = "min"


class StdAggregation(nn.Module):
    def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
        mean = scatter_shim.scatter_mean(x, index, 0, None, dim_size)
        mean2 = scatter_shim.scatter_mean(x * x, index, 0, None, dim_size)
        out = (mean2 - mean * mean).clamp(min=1e-5).sqrt()
        return out.masked_fill(out <= 1e-5 ** 0.5, 0.0)     # PyG: a clamped variance reads as 0


def ones(t):
    if t is not None:
        t.data.fill_(1.0)


def zeros(t):
    if t is not None:
        t.data.fill_(0.0)


def consecutive_cluster(src):
    unique, inv = torch.unique(src, sorted=True, return_inverse=True)
    perm = torch.arange(inv.size(0), dtype=inv.dtype, device=inv.device)
    perm = inv.new_empty(unique.size(0)).scatter_(0, inv, perm)
    return inv, perm


_MISSING = "???"   # PyG's sentinel: "edge_attr not passed" differs from "edge_attr=None"


def coalesce(edge_index, edge_attr=_MISSING, num_nodes=None, reduce="sum", **unused):
    """torch_geometric.utils.coalesce: sort edges by (row, col) and merge
    duplicates (imported by src/utils/scatter.py:6 and neighbors.py:7 for the
    'next' rows; plain torch - not on the per-step path)."""
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max()) + 1 if edge_index.numel() else 0
    key = edge_index[0] * max(n, 1) + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    ei = torch.stack([uniq // max(n, 1), uniq % max(n, 1)])
    if isinstance(edge_attr, str) and edge_attr == _MISSING:
        return ei
    if edge_attr is None:
        return ei, None
    red = "sum" if reduce in ("add", "sum") else reduce
    return ei, scatter_shim.scatter(edge_attr, inv, 0, None, uniq.numel(), red)


def remove_self_loops(edge_index, edge_attr=None):
    """torch_geometric.utils.remove_self_loops (src/utils/graph.py:8-10, 497): always a pair."""
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], None if edge_attr is None else edge_attr[keep]


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    """torch_geometric.utils.add_self_loops (src/transforms/graph.py:7, 1445-1449): one (i, i)
    edge per node appended; their attributes are ``fill_value`` (a number or a row; 1 when
    omitted) or, for a reduction name, that reduction of each node's incoming edge attributes."""
    n = int(num_nodes) if num_nodes is not None else (int(edge_index.max()) + 1 if edge_index.numel() else 0)
    loops = torch.arange(n, device=edge_index.device, dtype=edge_index.dtype).repeat(2, 1)
    if edge_attr is not None:
        shape = (n,) + tuple(edge_attr.shape[1:])
        if fill_value is None:
            fill = edge_attr.new_full(shape, 1.0)
        elif isinstance(fill_value, (int, float)):
            fill = edge_attr.new_full(shape, fill_value)
        elif torch.is_tensor(fill_value):
            fill = fill_value.to(edge_attr.device, edge_attr.dtype)
            fill = fill.expand(shape).contiguous() if fill.dim() != edge_attr.dim() or fill.shape[0] != n \
                else fill
        elif isinstance(fill_value, str):
            fill = scatter_shim.scatter(edge_attr, edge_index[1], 0, None, n, fill_value)
        else:
            raise AttributeError("No valid 'fill_value' provided")
        edge_attr = torch.cat([edge_attr, fill], dim=0)
    return torch.cat([edge_index, loops], dim=1), edge_attr


def to_undirected(edge_index, edge_attr=_MISSING, num_nodes=None, reduce="add"):
    """torch_geometric.utils.to_undirected (src/transforms/sampling.py:5): both directions of
    every edge, duplicates merged with ``reduce``."""
    missing = isinstance(edge_attr, str) and edge_attr == _MISSING
    ei = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if missing or edge_attr is None:
        out = coalesce(ei, None, num_nodes, reduce)[0]
        return out if missing else (out, None)
    return coalesce(ei, torch.cat([edge_attr, edge_attr], dim=0), num_nodes, reduce)
