"""SelfAttentionBlock (src/nn/attention.py:11-328) on the fused HIP kernel.

Same constructor, parameter names (``qkv``, ``k_rpe``, ``q_rpe``, ``v_rpe``,
``in_proj``, ``out_proj``) and forward signature as the reference, so its
checkpoints load and ``TransformerBlock`` / ``Stage`` call it unchanged.  The
dense Linears (qkv, out_proj) stay on rocBLAS through PyTorch; everything
between them is ONE kernel launch (``ops.edge_attention``)."""
import torch
from torch import nn

from .. import ops

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
        if (not isinstance(k_delta_rpe, bool)) or k_delta_rpe \
                or (not isinstance(q_delta_rpe, bool)) or q_delta_rpe:
            raise NotImplementedError(
                "k_delta_rpe / q_delta_rpe are not built on the HIP path (they are "
                "off in every shipped SPT config, configs/model/semantic/_attention.yaml:22-23)")
        self.k_delta_rpe = None
        self.q_delta_rpe = None

        self.in_proj = nn.Linear(in_dim, dim) if in_dim is not None else None
        self.out_proj = nn.Linear(dim, out_dim) if out_dim is not None else None
        if attn_drop is not None and attn_drop > 0:
            raise NotImplementedError(
                "attention dropout is not built on the HIP path (null in every "
                "shipped SPT config, configs/model/semantic/_down.yaml:15-17)")
        self.attn_drop = None
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

    def forward(self, x, edge_index, edge_attr=None, ea_grad=None):
        """x [N, Cx]; edge_index [2, E] (row 0 = querying source, row 1 = key
        target; any order) or an ``EdgeCSR``; edge_attr [E, in_rpe_dim]; ``ea_grad``: the
        stage's shared edge_attr gradient buffer (``ops.EdgeAttrGradShare``) or None."""
        if self.in_proj is not None:
            x = ops.linear(x, self.in_proj.weight, self.in_proj.bias)
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
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
        if self.out_proj is not None:
            x = ops.linear(x, self.out_proj.weight, self.out_proj.bias)
        if self.out_drop is not None:
            x = self.out_drop(x)
        return x

    def extra_repr(self):
        return f"dim={self.dim}, num_heads={self.num_heads}"
