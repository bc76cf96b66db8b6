"""Child -> parent pooling (src/nn/pool.py:24-330) on the segment-CSR kernels.

Aggregation pools are one segment reduce.  The attentive pools (parents query their children:
softmax over each parent's children of <q_parent, k_child>, weighted sum of the children's
values) are a chain of the same kernels: gather of the parents' queries, segment max / sum for
the softmax, segment sum of the weighted values."""
import torch
from torch import nn

from .. import ops
from ..csr import csr_of
from .attention import _qk_scale_spec

__all__ = ["pool_factory", "SumPool", "MeanPool", "MaxPool", "MinPool", "StdPool",
           "BaseAttentivePool", "AttentivePool", "AttentivePoolWithLearntQueries"]


class _AggregationPool(nn.Module):
    reduce = None

    def forward(self, x_child, x_parent, index, edge_attr=None, num_pool=None):
        return ops.segment_reduce(x_child, index, num_pool, self.reduce)


class SumPool(_AggregationPool):
    reduce = "sum"


class MeanPool(_AggregationPool):
    reduce = "mean"


class MaxPool(_AggregationPool):
    reduce = "max"


class MinPool(_AggregationPool):
    reduce = "min"


class StdPool(_AggregationPool):
    """PyG's StdAggregation (pool.py:80-81): sqrt(clamp(E[x^2] - E[x]^2, 1e-5)), a clamped
    variance reading as 0."""

    def forward(self, x_child, x_parent, index, edge_attr=None, num_pool=None):
        csr = csr_of(index, num_pool)
        mean = ops.segment_reduce(x_child, csr, None, "mean")
        mean2 = ops.segment_reduce(x_child * x_child, csr, None, "mean")
        out = (mean2 - mean * mean).clamp(min=1e-5).sqrt()
        return out.masked_fill(out <= 1e-5 ** 0.5, 0.0)


def pool_factory(pool, *args, **kwargs):
    if isinstance(pool, (_AggregationPool, BaseAttentivePool)):
        return pool
    table = {"max": MaxPool, "min": MinPool, "mean": MeanPool, "sum": SumPool, "std": StdPool}
    if isinstance(pool, str):
        if pool not in table:
            raise NotImplementedError(f"pool='{pool}' is not built on the HIP path")
        return table[pool]()
    return pool(*args, **kwargs)


class BaseAttentivePool(nn.Module):
    """pool.py:84-254; child classes provide ``_get_query``.  Same parameter names as the
    reference (``kv``, ``k_rpe``, ``q_rpe``, ``in_proj``, ``out_proj``)."""

    def __init__(self, dim=None, num_heads=1, in_dim=None, out_dim=None, qkv_bias=True,
                 qk_dim=8, qk_scale=None, attn_drop=None, drop=None, in_rpe_dim=9,
                 k_rpe=False, q_rpe=False, v_rpe=False, heads_share_rpe=False):
        super().__init__()
        assert dim % num_heads == 0, "dim must be a multiple of num_heads"
        self.dim, self.num_heads, self.qk_dim = dim, num_heads, qk_dim
        self.scale_mode, self.scale_a = _qk_scale_spec(dim, num_heads, qk_scale)
        self.heads_share_rpe = heads_share_rpe
        self.kv = nn.Linear(dim, qk_dim * num_heads + dim, bias=qkv_bias)
        rpe_dim = qk_dim if heads_share_rpe else qk_dim * num_heads

        def enc(flag):
            if not isinstance(flag, bool):
                return flag
            return nn.Linear(in_rpe_dim, rpe_dim) if flag else None

        self.k_rpe, self.q_rpe = enc(k_rpe), enc(q_rpe)
        if v_rpe:
            raise NotImplementedError
        self.in_proj = nn.Linear(in_dim, dim) if in_dim is not None else None
        self.out_proj = nn.Linear(dim, out_dim) if out_dim is not None else None
        self.attn_drop = nn.Dropout(attn_drop) if attn_drop is not None and attn_drop > 0 else None
        self.out_drop = nn.Dropout(drop) if drop is not None and drop > 0 else None

    def _rpe(self, lin, edge_attr, n):
        rpe = ops.linear(edge_attr, lin.weight, lin.bias)
        if self.heads_share_rpe:
            rpe = rpe.repeat(1, self.num_heads)
        return rpe.view(n, self.num_heads, -1)

    def forward(self, x_child, x_parent, index, edge_attr=None, num_pool=None):
        """x_child [Nc, Cc], x_parent [Np, Cp], index [Nc] = parent of each child,
        edge_attr [Nc, F] -> [Np, C] (or out_dim)."""
        nc = x_child.shape[0]
        n_parent = x_parent.shape[0] if num_pool is None else num_pool
        H, D = self.num_heads, self.qk_dim
        if self.in_proj is not None:
            x_child = ops.linear(x_child, self.in_proj.weight, self.in_proj.bias)
        csr = csr_of(index, n_parent)
        q = ops.gather_rows(self._get_query(x_parent), csr.idx).view(nc, H, D)
        kv = ops.linear(x_child, self.kv.weight, self.kv.bias)
        k = kv[:, :D * H].reshape(nc, H, D)
        v = kv[:, D * H:].reshape(nc, H, -1)

        # src/utils/nn.py:75-127 with s = index: D = (dim / heads)^-1/2, G = children^-1/2
        if self.scale_mode == 2:
            q = q * self.scale_a
        else:
            g = (csr.counts().float() ** -0.5)[csr.idx].view(-1, 1, 1)
            q = q * (self.scale_a * g if self.scale_mode == 0 else self.scale_a + g)
        if self.k_rpe is not None:
            k = k + self._rpe(self.k_rpe, edge_attr, nc)
        if self.q_rpe is not None:
            q = q + self._rpe(self.q_rpe, edge_attr, nc)

        compat = (q * k).sum(dim=-1)                                   # [Nc, H]
        # torch_geometric.utils.softmax over each parent's children (pool.py:224)
        mx = ops.segment_reduce(compat.detach(), csr, None, "max")
        e = (compat - ops.gather_rows(mx, csr.idx)).exp()
        attn = e / ops.gather_rows(ops.segment_reduce(e, csr, None, "sum") + 1e-16, csr.idx)
        if self.attn_drop is not None:
            attn = self.attn_drop(attn)
        x = ops.segment_reduce((v * attn.unsqueeze(-1)).reshape(nc, self.dim), csr, None, "sum")
        if self.out_proj is not None:
            x = ops.linear(x, self.out_proj.weight, self.out_proj.bias)
        if self.out_drop is not None:
            x = self.out_drop(x)
        return x

    def _get_query(self, x_parent):
        raise NotImplementedError

    def extra_repr(self):
        return f"dim={self.dim}, num_heads={self.num_heads}"


class AttentivePool(BaseAttentivePool):
    """Queries = a Linear of the parents' own features (pool.py:257-303)."""

    def __init__(self, dim=None, q_in_dim=None, num_heads=1, in_dim=None, out_dim=None,
                 qkv_bias=True, qk_dim=8, qk_scale=None, attn_drop=None, drop=None,
                 in_rpe_dim=9, k_rpe=False, q_rpe=False, v_rpe=False, heads_share_rpe=False):
        super().__init__(dim=dim, num_heads=num_heads, in_dim=in_dim, out_dim=out_dim,
                         qkv_bias=qkv_bias, qk_dim=qk_dim, qk_scale=qk_scale,
                         attn_drop=attn_drop, drop=drop, in_rpe_dim=in_rpe_dim, k_rpe=k_rpe,
                         q_rpe=q_rpe, v_rpe=v_rpe, heads_share_rpe=heads_share_rpe)
        self.q = nn.Linear(q_in_dim, qk_dim * num_heads, bias=qkv_bias)

    def _get_query(self, x_parent):
        return ops.linear(x_parent, self.q.weight, self.q.bias)


class AttentivePoolWithLearntQueries(BaseAttentivePool):
    """One learnt query per head, shared by all parents (pool.py:307-360); truncated-normal
    initialisation (std 0.02) like ``init_weights`` gives a ``LearnableParameter``."""

    def __init__(self, dim=None, num_heads=1, in_dim=None, out_dim=None, qkv_bias=True,
                 qk_dim=8, qk_scale=None, attn_drop=None, drop=None, in_rpe_dim=18,
                 k_rpe=False, q_rpe=False, v_rpe=False, heads_share_rpe=False):
        super().__init__(dim=dim, num_heads=num_heads, in_dim=in_dim, out_dim=out_dim,
                         qkv_bias=qkv_bias, qk_dim=qk_dim, qk_scale=qk_scale,
                         attn_drop=attn_drop, drop=drop, in_rpe_dim=in_rpe_dim, k_rpe=k_rpe,
                         q_rpe=q_rpe, v_rpe=v_rpe, heads_share_rpe=heads_share_rpe)
        self.q = nn.Parameter(torch.zeros(qk_dim * num_heads))
        nn.init.trunc_normal_(self.q, std=0.02)

    def _get_query(self, x_parent):
        return self.q.repeat(x_parent.shape[0], 1)
