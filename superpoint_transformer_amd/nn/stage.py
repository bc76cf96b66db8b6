"""Stage / DownNFuseStage / UpNFuseStage / PointStage (src/nn/stage.py).

Orchestration only: position injection (UnitSphereNorm), in_mlp, N
TransformerBlocks, out_mlp; pooling before a down stage, unpooling before an
up stage.  Same constructor arguments, module names and forward signatures as
the reference; the extra keyword ``num_graphs`` (number of clouds in the batch)
lets GraphNorm skip its ``batch.max()`` host sync."""
import torch
from torch import nn

from .. import ops
from ..csr import edge_csr_of
from .fusion import CatFusion, fusion_factory
from .mlp import MLP
from .norm import GraphNorm, UnitSphereNorm
from .pool import pool_factory
from .transformer import TransformerBlock
from .unpool import IndexUnpool

__all__ = ["Stage", "DownNFuseStage", "UpNFuseStage", "PointStage"]


def _shared_rpe_encoders(rpe, num_blocks, num_heads, in_dim, out_dim, blocks_share,
                         heads_share):
    """stage.py:289-318: one encoder per block, or one shared by all blocks."""
    if not isinstance(rpe, bool):
        assert blocks_share, \
            "a module passed as RPE encoder is shared by all blocks: set blocks_share_rpe"
        return [rpe] * num_blocks
    if not heads_share:
        out_dim = out_dim * num_heads
    if blocks_share and rpe:
        return [nn.Linear(in_dim, out_dim)] * num_blocks
    return [rpe] * num_blocks


class Stage(nn.Module):
    def __init__(self, dim, num_blocks=1, num_heads=1, in_mlp=None, out_mlp=None,
                 mlp_activation=nn.LeakyReLU(), mlp_norm=GraphNorm, mlp_drop=None,
                 use_pos=True, use_diameter=False, use_diameter_parent=False, qk_dim=8,
                 k_rpe=False, q_rpe=False, k_delta_rpe=False, q_delta_rpe=False,
                 qk_share_rpe=False, q_on_minus_rpe=False, blocks_share_rpe=False,
                 heads_share_rpe=False, version_holder=None, **transformer_kwargs):
        super().__init__()
        self.version_holder = version_holder
        self.dim = dim
        self.num_blocks = num_blocks
        self.num_heads = num_heads
        if in_mlp is not None:
            assert in_mlp[-1] == dim
            self.in_mlp = MLP(in_mlp, activation=mlp_activation, norm=mlp_norm, drop=mlp_drop)
        else:
            self.in_mlp = None
        if out_mlp is not None:
            assert out_mlp[0] == dim
            self.out_mlp = MLP(out_mlp, activation=mlp_activation, norm=mlp_norm, drop=mlp_drop)
        else:
            self.out_mlp = None
        if num_blocks > 0:
            # the reference hard-codes 18 input channels for block-shared edge
            # encoders (stage.py:150-156)
            k_blocks = _shared_rpe_encoders(k_rpe, num_blocks, num_heads, 18, qk_dim,
                                            blocks_share_rpe, heads_share_rpe)
            q_blocks = _shared_rpe_encoders(q_rpe and not (k_rpe and qk_share_rpe),
                                            num_blocks, num_heads, 18, qk_dim,
                                            blocks_share_rpe, heads_share_rpe)
            self.transformer_blocks = nn.ModuleList(
                TransformerBlock(dim, num_heads=num_heads, qk_dim=qk_dim, k_rpe=kb, q_rpe=qb,
                                 k_delta_rpe=k_delta_rpe, q_delta_rpe=q_delta_rpe,
                                 qk_share_rpe=qk_share_rpe, q_on_minus_rpe=q_on_minus_rpe,
                                 heads_share_rpe=heads_share_rpe,
                                 version_holder=self.version_holder, **transformer_kwargs)
                for kb, qb in zip(k_blocks, q_blocks))
        else:
            self.transformer_blocks = None
        self.pos_norm = UnitSphereNorm()
        self.feature_fusion = CatFusion()
        self.use_pos = use_pos
        self.use_diameter = use_diameter
        self.use_diameter_parent = use_diameter_parent

    @property
    def out_dim(self):
        if self.out_mlp is not None:
            return self.out_mlp.out_dim
        if self.transformer_blocks is not None:
            return self.transformer_blocks[-1].dim
        if self.in_mlp is not None:
            return self.in_mlp.out_dim
        return self.dim

    def _inject(self, x, pos, diameter, node_size, super_index, num_super, n, dtype, device):
        parts = [x]
        if pos is not None:                                   # stage.py:249-254
            normalized_pos, diameter_parent = self.pos_norm(
                pos, super_index, w=node_size, num_super=num_super)
            if self.use_pos:
                parts.insert(0, normalized_pos)
        else:
            diameter_parent = None
        if self.use_diameter:                                 # stage.py:257-261
            parts.insert(0, diameter if diameter is not None else
                         torch.zeros((n, 1), dtype=dtype, device=device))
        if self.use_diameter_parent:                          # stage.py:263-271
            if diameter_parent is None:
                diam = torch.zeros((n, 1), dtype=dtype, device=device)
            elif super_index is None:
                diam = diameter_parent.repeat(n, 1)
            else:
                diam = diameter_parent[super_index]
            parts.insert(0, diam)
        parts = [p for p in parts if p is not None]
        if len(parts) == 1:
            x = parts[0]
        elif parts:
            x = torch.cat(parts, dim=1)
        return x, diameter_parent

    def forward(self, x, norm_index, pos=None, diameter=None, node_size=None,
                super_index=None, edge_index=None, edge_attr=None, num_super=None,
                num_graphs=None, pool_to_parent=None, ea_grad=None):
        """``pool_to_parent`` = (parent batch vector or None): when the stage is only an
        in_mlp (PointStage) whose output feeds nothing but the max-pool to the parents, the
        pooled features are returned instead (``PooledToParent``) and the MLP's last
        norm + activation run inside the pool's read."""
        ref = x if x is not None else pos if pos is not None else diameter
        if ref is None:
            if super_index is None:
                raise ValueError("Could not infer basic info from input arguments")
            n, dtype, device = super_index.shape[0], torch.float, super_index.device
        else:
            n, dtype, device = ref.shape[0], ref.dtype, ref.device

        # position / diameter injection (stage.py:249-271): every fusion prepends its
        # columns, so [diameter_parent | diameter | normalized_pos | x] is built by ONE
        # concatenation instead of up to three passes over the level's rows
        if (self.use_pos and self.use_diameter_parent and not self.use_diameter
                and not self.pos_norm.log_diameter and ops.unit_sphere_assemble_ok(x, pos)):
            # [diameter_parent | normalized_pos | x] written by the normalisation's own pass
            x, diameter_parent = ops.unit_sphere_assemble(x, pos, super_index, w=node_size,
                                                          num_super=num_super)
        else:
            x, diameter_parent = self._inject(x, pos, diameter, node_size, super_index, num_super,
                                              n, dtype, device)

        if (pool_to_parent is not None and self.in_mlp is not None and super_index is not None
                and self.transformer_blocks is None and self.out_mlp is None):
            pooled = self.in_mlp.forward_max_pooled(
                x, super_index, num_super, batch=norm_index, batch_size=num_graphs,
                seg_graph=pool_to_parent[0])
            if pooled is not None:
                return PooledToParent(pooled), diameter_parent
        if self.in_mlp is not None:
            x = self.in_mlp(x, batch=norm_index, batch_size=num_graphs)
        if self.transformer_blocks is not None:
            if edge_index is not None and not hasattr(edge_index, "erowptr") \
                    and edge_index.shape[1] > 0:
                edge_index = edge_csr_of(edge_index, x.shape[0])   # once per stage
            # the blocks all read the same edge_attr: one shared gradient buffer
            # (``ea_grad``: a share handed down by SPT.forward, common to the down and the up
            # stage of a level, which read the same edge_attr too)
            share = ea_grad
            if share is None and (ops.share_edge_attr_grad() and edge_attr is not None
                                  and len(self.transformer_blocks) > 1
                                  and torch.is_grad_enabled() and edge_attr.requires_grad):
                share = ops.EdgeAttrGradShare()
            for block in self.transformer_blocks:
                x, norm_index, edge_index = block(
                    x, norm_index, edge_index=edge_index, edge_attr=edge_attr,
                    num_graphs=num_graphs, ea_grad=share)
        if self.out_mlp is not None:
            x = self.out_mlp(x, batch=norm_index, batch_size=num_graphs)
        return x, diameter_parent


class PooledToParent:
    """Child features already max-pooled to the parent level (see Stage.forward)."""

    __slots__ = ("x",)

    def __init__(self, x):
        self.x = x


class DownNFuseStage(Stage):
    """pool child features to the parents, fuse with the parents' own
    features, then a Stage (stage.py:321-444)."""

    def __init__(self, *args, pool="max", fusion="cat", **kwargs):
        super().__init__(*args, **kwargs)
        self.down_pool_block = pool_factory(pool)
        self.fusion = fusion_factory(fusion)

    def forward(self, x_parent, x_child, norm_index, pool_index, pos=None, diameter=None,
                node_size=None, super_index=None, edge_index=None, edge_attr=None,
                v_edge_attr=None, num_super=None, num_graphs=None, num_super_parent=None,
                ea_grad=None):
        if isinstance(x_child, PooledToParent):
            x_pooled = x_child.x
        else:
            x_pooled = self.down_pool_block(x_child, x_parent, pool_index,
                                            edge_attr=v_edge_attr, num_pool=num_super)
        x_fused = self.fusion(x_parent, x_pooled)
        return super().forward(x_fused, norm_index, pos=pos, node_size=node_size,
                               super_index=super_index, edge_index=edge_index,
                               edge_attr=edge_attr, num_super=num_super_parent,
                               num_graphs=num_graphs, ea_grad=ea_grad)


class UpNFuseStage(Stage):
    """unpool parent features to the children, fuse with the skip features,
    then a Stage (stage.py:447-571)."""

    def __init__(self, *args, unpool="index", fusion="cat", **kwargs):
        super().__init__(*args, **kwargs)
        if unpool != "index":
            raise NotImplementedError(f"Unknown unpool={unpool} mode")
        self.unpool = IndexUnpool()
        self.fusion = fusion_factory(fusion)

    def forward(self, x_child, x_parent, norm_index, unpool_index, pos=None, diameter=None,
                node_size=None, super_index=None, edge_index=None, edge_attr=None,
                num_super=None, num_graphs=None, ea_grad=None):
        x_unpool = self.unpool(x_parent, unpool_index)
        x_fused = self.fusion(x_child, x_unpool)
        return super().forward(x_fused, norm_index, pos=pos, node_size=node_size,
                               super_index=super_index, edge_index=edge_index,
                               edge_attr=edge_attr, num_super=num_super,
                               num_graphs=num_graphs, ea_grad=ea_grad)


class PointStage(Stage):
    """Level-0 stage: position injection + point MLP, no attention
    (stage.py:574-806, ``cnn_blocks=False`` branch; the torchsparse CNN encoder
    of EZ-SP is outside the hot path)."""

    def __init__(self, in_mlp, mlp_activation=nn.LeakyReLU(), mlp_norm=GraphNorm,
                 mlp_drop=None, use_pos=True, use_diameter_parent=False, cnn_blocks=False,
                 version_holder=None, **unused_cnn_kwargs):
        if cnn_blocks:
            raise NotImplementedError("the sparse-CNN point encoder is not on the HIP path")
        assert in_mlp is None or len(in_mlp) > 1
        super().__init__(in_mlp[-1] if in_mlp is not None else None, num_blocks=0,
                         in_mlp=in_mlp, out_mlp=None, mlp_activation=mlp_activation,
                         mlp_norm=mlp_norm, mlp_drop=mlp_drop, use_pos=use_pos,
                         use_diameter=False, use_diameter_parent=use_diameter_parent,
                         version_holder=version_holder)

    def forward(self, x, norm_index, pos=None, diameter=None, node_size=None,
                super_index=None, edge_index=None, edge_attr=None, coords=None, batch=None,
                x_mlp=None, num_super=None, num_graphs=None, pool_to_parent=None):
        return super().forward(x, norm_index, pos, diameter, node_size, super_index,
                               edge_index, edge_attr, num_super=num_super,
                               num_graphs=num_graphs, pool_to_parent=pool_to_parent)
