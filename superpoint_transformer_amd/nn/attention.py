"""SelfAttentionBlock (src/nn/attention.py:11-328) on the fused HIP kernel.

Same constructor, parameter names (``qkv``, ``k_rpe``, ``q_rpe``, ``v_rpe``,
``in_proj``, ``out_proj``) and forward signature as the reference, so its
checkpoints load and ``TransformerBlock`` / ``Stage`` call it unchanged.  The
dense Linears (qkv, out_proj) stay on rocBLAS through PyTorch; everything
between them is ONE kernel launch (``ops.edge_attention``).

Two options no shipped configuration turns on - RPE from node-feature differences
(``k_delta_rpe`` / ``q_delta_rpe``) and dropout on the attention weights while training - take
the reference's op-by-op route instead, every op on the segment-CSR kernels
(``_forward_composed``)."""
import torch
from torch import nn

from .. import ops
from ..csr import EdgeCSR, csr_of

__all__ = ["SelfAttentionBlock"]


def _qk_scale_spec(dim, num_heads, qk_scale):
    """(mode, a) of spt_edge_attn_*: src/utils/nn.py:75-127."""
    d = (dim // num_heads) ** -0.5
    if qk_scale is None:
        return 0, d
    if not isinstance(qk_scale, str):
        return 2, float(qk_scale)
    s = qk_scale.lower().replace(" ", "")
    if s in ("d+g", "g+d"):
        return 1, d
    if s in ("dg", "gd", "d*g", "g*d", "d.g", "g.d"):
        return 0, d
    if s == "d":
        return 2, d
    if s == "g":
        return 0, 1.0
    raise ValueError(f"Unable to build QK scaling scheme for qk_scale='{qk_scale}'")


class SelfAttentionBlock(nn.Module):
    def __init__(self, dim, num_heads=1, in_dim=None, out_dim=None, qkv_bias=True,
                 qk_dim=8, qk_scale=None, attn_drop=None, drop=None, in_rpe_dim=18,
                 k_rpe=False, q_rpe=False, v_rpe=False, k_delta_rpe=False,
                 q_delta_rpe=False, qk_share_rpe=False, q_on_minus_rpe=False,
                 heads_share_rpe=False):
        super().__init__()
        assert dim % num_heads == 0, "dim must be a multiple of num_heads"
        self.dim = dim
        self.num_heads = num_heads
        self.qk_dim = qk_dim
        self.scale_mode, self.scale_a = _qk_scale_spec(dim, num_heads, qk_scale)
        self.heads_share_rpe = heads_share_rpe
        self.qk_share_rpe = qk_share_rpe
        self.q_on_minus_rpe = q_on_minus_rpe

        self.qkv = nn.Linear(dim, qk_dim * 2 * num_heads + dim, bias=qkv_bias)

        qk_rpe_dim = qk_dim if heads_share_rpe else qk_dim * num_heads
        v_rpe_dim = dim // num_heads if heads_share_rpe else dim

        def enc(flag, out):
            if not isinstance(flag, bool):
                return flag                      # a shared encoder module
            return nn.Linear(in_rpe_dim, out) if flag else None

        self.k_rpe = enc(k_rpe, qk_rpe_dim)
        if not isinstance(q_rpe, bool):
            self.q_rpe = q_rpe
        else:
            self.q_rpe = nn.Linear(in_rpe_dim, qk_rpe_dim) \
                if q_rpe and not (k_rpe and qk_share_rpe) else None
        self.v_rpe = enc(v_rpe, v_rpe_dim)
        # attention.py:138-150: encoders of x[target] - x[source]
        if not isinstance(k_delta_rpe, bool):
            self.k_delta_rpe = k_delta_rpe
        else:
            self.k_delta_rpe = nn.Linear(dim, qk_rpe_dim) if k_delta_rpe else None
        if not isinstance(q_delta_rpe, bool):
            self.q_delta_rpe = q_delta_rpe
        else:
            self.q_delta_rpe = nn.Linear(dim, qk_rpe_dim) \
                if q_delta_rpe and not (k_delta_rpe and qk_share_rpe) else None

        self.in_proj = nn.Linear(in_dim, dim) if in_dim is not None else None
        self.out_proj = nn.Linear(dim, out_dim) if out_dim is not None else None
        self.attn_drop = nn.Dropout(attn_drop) if attn_drop is not None and attn_drop > 0 else None
        self.out_drop = nn.Dropout(drop) if drop is not None and drop > 0 else None

    def _expand(self, lin, negate=False):
        """(weight, bias) of an RPE encoder in the kernel's per-(head, dim) row
        layout; ``heads_share_rpe`` tiles the rows over heads
        (attention.py:228-230)."""
        if lin is None:
            return None
        w, b = lin.weight, lin.bias
        if negate:
            w = -w                                # Linear(-x) = -W x + b
        if self.heads_share_rpe:
            w = w.repeat(self.num_heads, 1)
            b = None if b is None else b.repeat(self.num_heads)
        return w, b

    def _composed(self):
        return (self.k_delta_rpe is not None or self.q_delta_rpe is not None
                or (self.attn_drop is not None and self.training))

    def forward_prenorm_residual(self, x, norm, norm_index, num_graphs, edge_index,
                                 edge_attr=None, ea_grad=None):
        """``x + self(norm(x), ...)`` - the pre-norm self-attention branch of a TransformerBlock
        (src/nn/transformer.py:231-234) - with the GraphNorm applied inside the qkv Linear's read of
        x and the residual added in the out_proj Linear's epilogue (``ops.norm_linear`` /
        ``ops.linear_residual``: bitwise the unfused values, one apply pass and two elementwise adds
        fewer per block).  Returns None when the fused route does not apply (the caller then runs
        norm, block and add one after the other)."""
        if (self.in_proj is not None or self.out_proj is None or self._composed()
                or (self.out_drop is not None and self.training)
                or getattr(norm, "generic", False)
                or not ops.norm_linear_ok(x, norm_index, num_graphs, self.qkv.weight)
                or not ops.linear_residual_ok(x, self.out_proj.weight)):
            return None
        qkv, x_res = ops.norm_linear(x, norm_index, num_graphs, norm.weight, norm.bias,
                                     norm.mean_scale, norm.eps, self.qkv.weight, self.qkv.bias)
        return self._attend(qkv, edge_index, edge_attr, ea_grad, residual=x_res)

    def forward(self, x, edge_index, edge_attr=None, ea_grad=None):
        """x [N, Cx]; edge_index [2, E] (row 0 = querying source, row 1 = key
        target; any order) or an ``EdgeCSR``; edge_attr [E, in_rpe_dim]; ``ea_grad``: the
        stage's shared edge_attr gradient buffer (``ops.EdgeAttrGradShare``) or None."""
        if self.in_proj is not None:
            x = ops.linear(x, self.in_proj.weight, self.in_proj.bias)
        if self._composed():
            return self._forward_composed(x, edge_index, edge_attr)
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        return self._attend(qkv, edge_index, edge_attr, ea_grad)

    def _attend(self, qkv, edge_index, edge_attr, ea_grad, residual=None):
        k_rpe = q_rpe = v_rpe = None
        if edge_attr is not None:
            k_rpe = self._expand(self.k_rpe)
            if self.q_rpe is not None:
                q_rpe = self._expand(self.q_rpe, self.q_on_minus_rpe)
            elif self.k_rpe is not None and self.qk_share_rpe:
                q_rpe = self._expand(self.k_rpe, self.q_on_minus_rpe)
            v_rpe = self._expand(self.v_rpe)
        x = ops.edge_attention(
            qkv, edge_index, edge_attr if (k_rpe or q_rpe or v_rpe) else None,
            k_rpe=k_rpe, q_rpe=q_rpe, v_rpe=v_rpe, num_heads=self.num_heads,
            qk_dim=self.qk_dim, scale_mode=self.scale_mode, scale_a=self.scale_a,
            ea_grad=ea_grad)
        if residual is not None:                  # forward_prenorm_residual: out_proj exists
            return ops.linear_residual(x, self.out_proj.weight, self.out_proj.bias, residual)
        if self.out_proj is not None:
            x = ops.linear(x, self.out_proj.weight, self.out_proj.bias)
        if self.out_drop is not None:
            x = self.out_drop(x)
        return x

    def _forward_composed(self, x, edge_index, edge_attr):
        """attention.py:197-326 op by op (``x`` already through ``in_proj``): gathers of q / k / v
        to the edges, the RPE Linears, the per-source softmax as segment max / sum, the weighted
        values as a segment sum.  An ``EdgeCSR`` is walked in its own (source-sorted) order."""
        H, D, n = self.num_heads, self.qk_dim, x.shape[0]
        if isinstance(edge_index, EdgeCSR):
            counts = (edge_index.erowptr[1:] - edge_index.erowptr[:-1]).long()
            s = torch.arange(n, device=x.device).repeat_interleave(counts)
            t = edge_index.tgt_sorted.long()
            if edge_attr is not None:
                edge_attr = ops.gather_rows(edge_attr, edge_index.eperm.long())
        else:
            s, t = edge_index[0].contiguous(), edge_index[1].contiguous()
        E = s.numel()
        csr = csr_of(s, n)
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        q = ops.gather_rows(qkv[:, :D * H], s).view(E, H, D)
        k = ops.gather_rows(qkv[:, D * H:2 * D * H], t).view(E, H, D)
        v = ops.gather_rows(qkv[:, 2 * D * H:], t).view(E, H, -1)
        if self.scale_mode == 2:                                   # src/utils/nn.py:75-127
            q = q * self.scale_a
        else:
            g = (csr.counts().float() ** -0.5)[s].view(-1, 1, 1)
            q = q * (self.scale_a * g if self.scale_mode == 0 else self.scale_a + g)

        def rpe(lin, feat):
            r = ops.linear(feat, lin.weight, lin.bias)
            return (r.repeat(1, H) if self.heads_share_rpe else r).view(E, H, -1)

        sign = -1.0 if self.q_on_minus_rpe else 1.0
        if edge_attr is not None:
            if self.k_rpe is not None:
                k = k + rpe(self.k_rpe, edge_attr)
            if self.q_rpe is not None:
                q = q + rpe(self.q_rpe, sign * edge_attr)
            elif self.k_rpe is not None and self.qk_share_rpe:
                q = q + rpe(self.k_rpe, sign * edge_attr)
        if self.k_delta_rpe is not None or self.q_delta_rpe is not None:
            delta = ops.gather_rows(x, t) - ops.gather_rows(x, s)     # attention.py:258, 271
            if self.k_delta_rpe is not None:
                k = k + rpe(self.k_delta_rpe, delta)
            if self.q_delta_rpe is not None:
                q = q + rpe(self.q_delta_rpe, sign * delta)
            elif self.k_delta_rpe is not None and self.qk_share_rpe and edge_attr is not None:
                q = q + rpe(self.k_delta_rpe, sign * delta)
        if self.v_rpe is not None and edge_attr is not None:
            v = v + rpe(self.v_rpe, edge_attr)

        compat = (q * k).sum(dim=-1)                                   # [E, H]
        mx = ops.segment_reduce(compat.detach(), csr, None, "max")
        e = (compat - ops.gather_rows(mx, s)).exp()
        attn = e / ops.gather_rows(ops.segment_reduce(e, csr, None, "sum") + 1e-16, s)
        if self.attn_drop is not None:
            attn = self.attn_drop(attn)
        x = ops.segment_reduce((v * attn.unsqueeze(-1)).reshape(E, self.dim), csr, None, "sum")
        if self.out_proj is not None:
            x = ops.linear(x, self.out_proj.weight, self.out_proj.bias)
        if self.out_drop is not None:
            x = self.out_drop(x)
        return x

    def extra_repr(self):
        return f"dim={self.dim}, num_heads={self.num_heads}"
