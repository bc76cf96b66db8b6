"""TransformerBlock (src/nn/transformer.py:14-265): pre/post-norm residual
self-attention and FFN branches."""
from torch import nn

from .attention import SelfAttentionBlock
from .dropout import DropPath
from .mlp import FFN
from .norm import INDEX_BASED_NORMS, GraphNorm

__all__ = ["TransformerBlock", "VersionHolder"]


class VersionHolder:
    """Code version that fixes the FFN residual definition
    (transformer.py:240-244; src/utils/version.py)."""

    def __init__(self, value="3.0.0", commit_hash=None):
        self.value = value
        self.commit_hash = commit_hash

    @property
    def major(self):
        return int(str(self.value).split(".")[0])

    @property
    def minor(self):
        return int(str(self.value).split(".")[1])


class TransformerBlock(nn.Module):
    def __init__(self, dim, num_heads=1, qkv_bias=True, qk_dim=8, qk_scale=None,
                 in_rpe_dim=18, ffn_ratio=4, attn_drop=None, residual_drop=None,
                 drop_path=None, activation=nn.LeakyReLU(), norm=GraphNorm, pre_norm=True,
                 no_sa=False, no_ffn=False, k_rpe=False, q_rpe=False, v_rpe=False,
                 k_delta_rpe=False, q_delta_rpe=False, qk_share_rpe=False,
                 q_on_minus_rpe=False, heads_share_rpe=False, version_holder=None):
        super().__init__()
        self.dim = dim
        self.pre_norm = pre_norm
        self.version_holder = version_holder or VersionHolder()
        self.no_sa = no_sa
        if not no_sa:
            self.sa_norm = norm(dim)
            self.sa = SelfAttentionBlock(
                dim, num_heads=num_heads, in_dim=None, out_dim=dim, qkv_bias=qkv_bias,
                qk_dim=qk_dim, qk_scale=qk_scale, in_rpe_dim=in_rpe_dim,
                attn_drop=attn_drop, drop=residual_drop, k_rpe=k_rpe, q_rpe=q_rpe,
                v_rpe=v_rpe, k_delta_rpe=k_delta_rpe, q_delta_rpe=q_delta_rpe,
                qk_share_rpe=qk_share_rpe, q_on_minus_rpe=q_on_minus_rpe,
                heads_share_rpe=heads_share_rpe)
        self.no_ffn = no_ffn
        if not no_ffn:
            self.ffn_norm = norm(dim)
            self.ffn_ratio = ffn_ratio
            self.ffn = FFN(dim, hidden_dim=int(dim * ffn_ratio), activation=activation,
                           drop=residual_drop)
        self.drop_path = DropPath(drop_path) if drop_path is not None and drop_path > 0 \
            else nn.Identity()

    def forward(self, x, norm_index, edge_index=None, edge_attr=None, num_graphs=None,
                ea_grad=None):
        shortcut = x
        has_edges = edge_index is not None and \
            (getattr(edge_index, "e", None) or getattr(edge_index, "shape", (0, 0))[1]) > 0
        if self.no_sa or not has_edges:
            pass
        elif self.pre_norm:
            out = None
            if isinstance(self.sa_norm, GraphNorm) and (
                    isinstance(self.drop_path, nn.Identity) or not self.training):
                # norm inside the qkv Linear's read, residual in the out_proj Linear's epilogue
                out = self.sa.forward_prenorm_residual(x, self.sa_norm, norm_index, num_graphs,
                                                       edge_index, edge_attr=edge_attr,
                                                       ea_grad=ea_grad)
            if out is None:
                x = self._forward_norm(self.sa_norm, x, norm_index, num_graphs)
                x = self.sa(x, edge_index, edge_attr=edge_attr, ea_grad=ea_grad)
                out = shortcut + self.drop_path(x)
            x = out
        else:
            x = self.sa(x, edge_index, edge_attr=edge_attr, ea_grad=ea_grad)
            x = self.drop_path(x)
            x = self._forward_norm(self.sa_norm, shortcut + x, norm_index, num_graphs)

        vh = self.version_holder
        if vh.major >= 3 or (vh.major == 2 and vh.minor >= 2):
            shortcut = x

        if not self.no_ffn and self.pre_norm:
            out = None
            if isinstance(self.ffn_norm, GraphNorm) and (
                    isinstance(self.drop_path, nn.Identity) or not self.training):
                out = self.ffn.forward_prenorm_residual(x, self.ffn_norm, norm_index, num_graphs)
            if out is None:
                x = self._forward_norm(self.ffn_norm, x, norm_index, num_graphs)
                x = self.ffn(x)
                out = shortcut + self.drop_path(x)
            x = out
        if not self.no_ffn and not self.pre_norm:
            x = self.ffn(x)
            x = self.drop_path(x)
            x = self._forward_norm(self.ffn_norm, shortcut + x, norm_index, num_graphs)
        return x, norm_index, edge_index

    @staticmethod
    def _forward_norm(norm, x, norm_index, num_graphs=None):
        if isinstance(norm, GraphNorm):
            return norm(x, batch=norm_index, batch_size=num_graphs)
        if isinstance(norm, INDEX_BASED_NORMS):
            return norm(x, batch=norm_index)
        return norm(x)
