"""SPT backbone driver over NAG levels (src/models/components/spt.py:14-981).

U-Net over the hierarchy: PointStage on level 0, DownNFuseStages (pool ->
fuse -> transformer blocks) going up, UpNFuseStages (unpool -> fuse ->
blocks) coming back down to level 1.  Constructor arguments and module names
(``first_stage``, ``down_stages``, ``up_stages``, ``node_mlps``,
``h_edge_mlps``, ``v_edge_mlps``) follow the reference so that its state
dicts load.  ``nano=True`` (spt.py:486-521, 786-797: no level-0 stage - the NAG starts at level
1, whose handcrafted features go through ``node_mlps[0]`` / ``h_edge_mlps[0]`` into a full
``Stage`` with transformer blocks) is built; the sparse-CNN point encoder is outside the hot
path and raises."""
import torch
from torch import nn

from .. import ops
from ..csr import adopt_csr, verify_adopted
from .fusion import CatFusion
from .mlp import MLP
from .norm import GraphNorm
from .pool import BaseAttentivePool, MaxPool, pool_factory
from .stage import DownNFuseStage, PointStage, Stage, UpNFuseStage
from .transformer import VersionHolder

__all__ = ["SPT"]


def _listify(*args):
    """spt.py `listify_with_reference`: broadcast scalars to the length of the
    first (list) argument."""
    ref = args[0]
    if ref is None:
        return [[] for _ in args]
    if not isinstance(ref, (list, tuple)):
        ref = [ref]
    n = len(ref)
    out = [list(ref)]
    for a in args[1:]:
        if isinstance(a, (list, tuple)) and len(a) == n:
            out.append(list(a))
        elif isinstance(a, (list, tuple)) and n == 0:
            out.append([])
        else:
            out.append([a] * n)
    return out


def _mlps(layers, num_stage, activation, norm, shared):
    if layers is None:
        return [None] * num_stage
    if shared:
        return nn.ModuleList([MLP(layers, activation=activation, norm=norm)] * num_stage)
    return nn.ModuleList([MLP(layers, activation=activation, norm=norm)
                          for _ in range(num_stage)])


def _rpe_per_stage(rpe, num_stages, in_dim, out_dim, stages_share):
    if not isinstance(rpe, bool):
        assert stages_share
        return [rpe] * num_stages
    if stages_share and rpe:
        return [nn.Linear(in_dim, out_dim)] * num_stages
    return [rpe] * num_stages


def _share_for(edge_attr):
    """A gradient-buffer share for every attention block that will read ``edge_attr`` in this
    forward pass, or None (no gradient wanted / sharing off)."""
    import torch
    from .. import ops
    if (edge_attr is not None and ops.share_edge_attr_grad() and torch.is_grad_enabled()
            and edge_attr.requires_grad):
        return ops.EdgeAttrGradShare()
    return None


def _get(data, key, default=None):
    if isinstance(data, dict):
        return data.get(key, default)
    return getattr(data, key, default)


def _adopt_sub_views(levels):
    """A NAG level stores its partition twice: ``super_index`` on the children and the same
    clusters as a CSR on the parents (``nag[i+1].sub``, src/data/cluster.py:19-77).  Hand the
    stored CSR to the segment kernels as the view of ``super_index`` (the pool, UnitSphereNorm,
    the unpool's backward all group by it) instead of sorting the index again every batch."""
    for lo, hi in zip(levels[:-1], levels[1:]):
        si, sub = _get(lo, "super_index"), _get(hi, "sub")
        if si is None or sub is None or not torch.is_tensor(si) or not si.is_cuda:
            continue
        pointers, points = getattr(sub, "pointers", None), getattr(sub, "points", None)
        if pointers is None or points is None or points.device != si.device:
            continue
        asc = getattr(sub, "_ascending", None)
        if asc is False:
            # a consistent `sub` whose clusters do not ascend (h5io.load_nag of a reference file:
            # Cluster sorts with a non-stable sort, src/data/cluster.py:19-77): legitimate, but not
            # the stable sort's view - the level goes to the device sort
            continue
        # (adopt_csr checks pointers, point range, membership and the ascending order with the
        # kernel that writes the clamped int32 view; the verdict is read a step later without a
        # host round trip - csr.verify_adopted: a stale `sub` raises StaleCSRError and loses its
        # view, an order-different one just goes back to the sort)
        adopt_csr(si, pointers.numel() - 1, pointers, points, ascending=asc, verify="deferred")


class SPT(nn.Module):
    def __init__(self, point_mlp=None, point_drop=None, nano=False, down_dim=None,
                 down_pool_dim=None, down_in_mlp=None, down_out_mlp=None,
                 down_mlp_drop=None, down_num_heads=1, down_num_blocks=0, down_ffn_ratio=4,
                 down_residual_drop=None, down_attn_drop=None, down_drop_path=None,
                 up_dim=None, up_in_mlp=None, up_out_mlp=None, up_mlp_drop=None,
                 up_num_heads=1, up_num_blocks=0, up_ffn_ratio=4, up_residual_drop=None,
                 up_attn_drop=None, up_drop_path=None, node_mlp=None, h_edge_mlp=None,
                 v_edge_mlp=None, mlp_activation=nn.LeakyReLU(), mlp_norm=GraphNorm,
                 qk_dim=8, qkv_bias=True, qk_scale=None, in_rpe_dim=18,
                 activation=nn.LeakyReLU(), norm=GraphNorm, pre_norm=True, no_sa=False,
                 no_ffn=False, k_rpe=False, q_rpe=False, v_rpe=False, k_delta_rpe=False,
                 q_delta_rpe=False, qk_share_rpe=False, q_on_minus_rpe=False,
                 share_hf_mlps=False, stages_share_rpe=False, blocks_share_rpe=False,
                 heads_share_rpe=False, use_pos=True, use_node_hf=True, use_diameter=False,
                 use_diameter_parent=False, pool="max", unpool="index", fusion="cat",
                 norm_mode="graph", output_stage_wise=False, version="3.0.0",
                 matrix_precision=None, **ignored):
        super().__init__()
        # not in the reference: "f32" | "bf16" | "f32-exact" pins this model's GEMM precision
        # (precision.matrix_precision, per call); None = whatever is active around the call
        self.matrix_precision = matrix_precision
        if norm_mode not in ("graph", "node", "segment"):
            raise NotImplementedError(f"Unknown mode='{norm_mode}'")      # data.py:130
        self.norm_mode = norm_mode
        self.nano = nano = bool(nano)
        self.use_pos, self.use_node_hf = use_pos, use_node_hf
        self.use_diameter, self.use_diameter_parent = use_diameter, use_diameter_parent
        self.output_stage_wise = output_stage_wise
        self.version_holder = VersionHolder(version)

        (down_dim, down_pool_dim, down_in_mlp, down_out_mlp, down_mlp_drop, down_num_heads,
         down_num_blocks, down_ffn_ratio, down_residual_drop, down_attn_drop,
         down_drop_path, pool) = _listify(
            down_dim, down_pool_dim, down_in_mlp, down_out_mlp, down_mlp_drop,
            down_num_heads, down_num_blocks, down_ffn_ratio, down_residual_drop,
            down_attn_drop, down_drop_path, pool)
        (up_dim, up_in_mlp, up_out_mlp, up_mlp_drop, up_num_heads, up_num_blocks,
         up_ffn_ratio, up_residual_drop, up_attn_drop, up_drop_path) = _listify(
            up_dim, up_in_mlp, up_out_mlp, up_mlp_drop, up_num_heads, up_num_blocks,
            up_ffn_ratio, up_residual_drop, up_attn_drop, up_drop_path)
        # in_mlp entries are themselves lists: _listify must not broadcast them
        # spt.py:450-483: with nano the first entry of every down_* list configures the first
        # Stage, and the handcrafted-feature MLPs get one more entry (run before that Stage)
        num_down, num_up = len(down_dim) - nano, len(up_dim)

        needs_h_edge = any(b > 0 for b in down_num_blocks + up_num_blocks)
        self.node_mlps = _mlps(node_mlp if use_node_hf else None, num_down + nano,
                               mlp_activation, mlp_norm, share_hf_mlps)
        self.h_edge_mlps = _mlps(h_edge_mlp if needs_h_edge else None, num_down + nano,
                                 mlp_activation, mlp_norm, share_hf_mlps)
        # vertical (child -> parent) edge features only feed attentive pools (spt.py:453-483)
        needs_v_edge = num_down > 0 and isinstance(
            pool_factory(pool[0], down_pool_dim[0]), BaseAttentivePool)
        self.v_edge_mlps = _mlps(v_edge_mlp if needs_v_edge else None, num_down, mlp_activation,
                                 mlp_norm, share_hf_mlps)
        self.feature_fusion = CatFusion()

        common = dict(mlp_activation=mlp_activation, mlp_norm=mlp_norm, qk_dim=qk_dim,
                      qkv_bias=qkv_bias, qk_scale=qk_scale, in_rpe_dim=in_rpe_dim,
                      activation=activation, norm=norm, pre_norm=pre_norm, no_sa=no_sa,
                      no_ffn=no_ffn, v_rpe=v_rpe, k_delta_rpe=k_delta_rpe,
                      q_delta_rpe=q_delta_rpe, qk_share_rpe=qk_share_rpe,
                      q_on_minus_rpe=q_on_minus_rpe, use_pos=use_pos,
                      use_diameter=use_diameter, use_diameter_parent=use_diameter_parent,
                      blocks_share_rpe=blocks_share_rpe, heads_share_rpe=heads_share_rpe,
                      version_holder=self.version_holder)
        if nano:                                              # spt.py:486-521
            self.first_stage = Stage(
                down_dim[0], num_blocks=down_num_blocks[0], in_mlp=down_in_mlp[0],
                out_mlp=down_out_mlp[0], mlp_drop=down_mlp_drop[0],
                num_heads=down_num_heads[0], ffn_ratio=down_ffn_ratio[0],
                residual_drop=down_residual_drop[0], attn_drop=down_attn_drop[0],
                drop_path=down_drop_path[0], k_rpe=k_rpe, q_rpe=q_rpe, **common)
        else:
            self.first_stage = PointStage(
                point_mlp, mlp_activation=mlp_activation, mlp_norm=mlp_norm,
                mlp_drop=point_drop, use_pos=use_pos, use_diameter_parent=use_diameter_parent,
                version_holder=self.version_holder)
        if num_down > 0:
            # the per-stage lists keep the length of down_dim: entry 0 is the nano Stage's
            dk = [None] * nano + _rpe_per_stage(k_rpe, num_down, 18, qk_dim, stages_share_rpe)
            dq = [None] * nano + _rpe_per_stage(q_rpe and not (k_rpe and qk_share_rpe), num_down,
                                                18, qk_dim, stages_share_rpe)
            self.down_stages = nn.ModuleList([
                DownNFuseStage(
                    down_dim[i], num_blocks=down_num_blocks[i], in_mlp=down_in_mlp[i],
                    out_mlp=down_out_mlp[i], mlp_drop=down_mlp_drop[i],
                    num_heads=down_num_heads[i], ffn_ratio=down_ffn_ratio[i],
                    residual_drop=down_residual_drop[i], attn_drop=down_attn_drop[i],
                    drop_path=down_drop_path[i], k_rpe=dk[i], q_rpe=dq[i],
                    pool=pool_factory(pool[i], down_pool_dim[i]), fusion=fusion, **common)
                for i in range(nano, num_down + nano)])
        else:
            self.down_stages = None
        if num_up > 0:
            uk = _rpe_per_stage(k_rpe, num_up, 18, qk_dim, stages_share_rpe)
            uq = _rpe_per_stage(q_rpe and not (k_rpe and qk_share_rpe), num_up, 18, qk_dim,
                                stages_share_rpe)
            self.up_stages = nn.ModuleList([
                UpNFuseStage(
                    up_dim[i], num_blocks=up_num_blocks[i], in_mlp=up_in_mlp[i],
                    out_mlp=up_out_mlp[i], mlp_drop=up_mlp_drop[i], num_heads=up_num_heads[i],
                    ffn_ratio=up_ffn_ratio[i], residual_drop=up_residual_drop[i],
                    attn_drop=up_attn_drop[i], drop_path=up_drop_path[i], k_rpe=uk[i],
                    q_rpe=uq[i], unpool=unpool, fusion=fusion, **common)
                for i in range(num_up)])
        else:
            self.up_stages = None

    @property
    def num_down_stages(self):
        return len(self.down_stages) if self.down_stages is not None else 0

    @property
    def num_up_stages(self):
        return len(self.up_stages) if self.up_stages is not None else 0

    @property
    def out_dim(self):
        if self.output_stage_wise:
            return [s.out_dim for s in self.up_stages][::-1] + [self.down_stages[-1].out_dim]
        if self.up_stages is not None:
            return self.up_stages[-1].out_dim
        if self.down_stages is not None:
            return self.down_stages[-1].out_dim
        return self.first_stage.out_dim

    def forward(self, nag):
        if self.norm_mode != "graph":
            # group index = one group per node / per (segment, cloud): the statistics of every
            # GraphNorm run on the segment-CSR kernels, the fused few-sorted-graphs routes are off
            if self.nano:
                raise NotImplementedError("norm_mode != 'graph' is not built for nano models")
            for m in self.modules():
                if isinstance(m, GraphNorm):
                    m.generic = True
        if self.matrix_precision is None:
            out = self._forward(nag)
        else:
            from .. import precision
            with precision.matrix_precision(self.matrix_precision):
                out = self._forward(nag)
        if not (self.training and torch.is_grad_enabled()) and not torch.cuda.is_current_stream_capturing():
            # results are about to leave the device (eval / inference / a single forward): the
            # adopted views' verdicts are read NOW - a stale `sub` raises here, not never.  A
            # training loop reads them a step later without waiting (csr.verify_adopted) and
            # calls csr.verify_adopted(block=True) after its last batch.
            verify_adopted(block=True)
        return out

    def _forward(self, nag):
        """``nag[i]`` exposes pos, x, super_index, node_size, batch, edge_index,
        edge_attr (attributes or dict keys).  ``nag.num_clouds`` (optional) is
        the number of clouds in the batch.  Returns level-1 features (or the
        stage-wise list), like spt.py:760-879."""
        B = _get(nag, "num_clouds", None) if not isinstance(nag, (list, tuple)) else None
        if self.nano:
            return self._forward_nano(nag, B)
        levels = [nag[i] for i in range(self.num_down_stages + 1)]
        sizes = [_get(lv, "pos").shape[0] for lv in levels]
        _adopt_sub_views(levels)

        def norm_index(lv):                                  # Data.norm_index(mode), data.py:103-130
            batch = _get(lv, "batch")
            if self.norm_mode == "graph":
                return batch
            n = _get(lv, "pos").shape[0]
            dev = _get(lv, "pos").device
            if self.norm_mode == "node":
                return torch.arange(n, device=dev)
            sup = _get(lv, "super_index")
            if sup is None:
                sup = torch.zeros(n, dtype=torch.long, device=dev)
            if batch is None:
                return sup
            nb = B if B is not None else int(batch.max()) + 1
            return sup * nb + batch

        d0 = levels[0]
        # the point features feed nothing but the max-pool of the first down stage: let the
        # first stage hand over pooled features (its last norm then runs inside the pool)
        fuse_pool = (self.num_down_stages > 0 and
                     isinstance(getattr(self.down_stages[0], "down_pool_block", None), MaxPool))
        x, diameter = self.first_stage(                       # spt.py:894-913
            _get(d0, "x") if self.use_node_hf else None, norm_index(d0),
            pos=_get(d0, "pos"), node_size=_get(d0, "node_size"),
            super_index=_get(d0, "super_index"), num_super=sizes[1] if len(sizes) > 1 else None,
            num_graphs=B,
            pool_to_parent=(norm_index(levels[1]),) if fuse_pool else None)

        down_outputs, node_x, edge_attrs, ea_shares = [], {}, {}, {}
        for i_stage in range(self.num_down_stages):
            i_level = i_stage + 1
            lv = levels[i_level]
            stage = self.down_stages[i_stage]
            ni = norm_index(lv)
            xh = _get(lv, "x")
            if self.node_mlps[i_stage] is not None and xh is not None:     # spt.py:823-826
                xh = self.node_mlps[i_stage](xh, batch=ni, batch_size=B)
            ea = _get(lv, "edge_attr")
            ei = _get(lv, "edge_index")
            if self.h_edge_mlps[i_stage] is not None and ea is not None:   # spt.py:827-835
                one = B == 1 and self.norm_mode == "graph"              # one cloud: one graph
                eb = None if (ni is None or one) else ni[ei[0]]
                if eb is not None and B is not None and self.norm_mode == "graph" and eb.is_cuda:
                    # sorted inside each third of [i<j | j>i | loops]: a handful of runs
                    ops.graph_runs_via(eb, B, ei, ei, ni)
                ea = self.h_edge_mlps[i_stage](ea, batch=eb, batch_size=B)
            node_x[i_level], edge_attrs[i_level] = xh, ea
            # the down stage and, later, the up stage of this level read the same edge_attr:
            # one gradient buffer for all their blocks (the first block of the down stage - the
            # last one autograd reaches - hands it over)
            ea_shares[i_level] = _share_for(ea)
            v_ea = self._v_edge_attr(i_stage, levels[i_level - 1], B)
            is_last = i_level == len(levels) - 1
            x, diameter = stage(                                            # spt.py:915-930
                xh if self.use_node_hf else None, x, ni, _get(levels[i_level - 1], "super_index"),
                pos=_get(lv, "pos"), node_size=_get(lv, "node_size"),
                super_index=None if is_last else _get(lv, "super_index"),
                edge_index=ei, edge_attr=ea, v_edge_attr=v_ea, num_super=sizes[i_level],
                num_graphs=B, num_super_parent=None if is_last else sizes[i_level + 1],
                ea_grad=ea_shares[i_level])
            down_outputs.append(x)

        up_outputs = []
        for i_stage in range(self.num_up_stages):                           # spt.py:860-868
            i_level = self.num_down_stages - i_stage - 1
            lv = levels[i_level]
            x_skip = down_outputs[-(2 + i_stage)]
            xh = node_x.get(i_level) if self.use_node_hf else None
            x, _ = self.up_stages[i_stage](
                self.feature_fusion(x_skip, xh), x, norm_index(lv), _get(lv, "super_index"),
                pos=_get(lv, "pos"), node_size=_get(lv, "node_size"),
                super_index=_get(lv, "super_index"), edge_index=_get(lv, "edge_index"),
                edge_attr=edge_attrs.get(i_level), num_super=sizes[i_level + 1], num_graphs=B,
                ea_grad=ea_shares.get(i_level))
            up_outputs.append(x)

        if self.output_stage_wise:
            return [x] + up_outputs[::-1][1:] + [down_outputs[-1]]
        return x

    def _v_edge_attr(self, i_stage, child, B):
        """The children's vertical edge features for the pool of down stage ``i_stage``, through
        their MLP when there is one (spt.py:836-841, 929).  The reference reads the features
        from the PARENT level there while normalising with the children's index - shapes that
        only agree by accident; the children's own ``v_edge_attr`` (what
        ``_on_the_fly_vertical_edge_features`` writes, transforms/graph.py:1411-1414, and what
        spt.py:929 hands to the stage when there is no MLP) is used here."""
        v_ea = _get(child, "v_edge_attr")
        mlp = self.v_edge_mlps[i_stage]
        if mlp is None or v_ea is None:
            return v_ea
        if self.norm_mode != "graph":
            raise NotImplementedError("vertical edge MLPs with norm_mode != 'graph' are not built")
        ni = _get(child, "batch")
        return mlp(v_ea, batch=None if (ni is None or B == 1) else ni, batch_size=B)

    def _forward_nano(self, nag, B):
        """spt.py:760-879 with ``nano=True``: ``nag[1]`` is the first level the model sees
        (``nag.start_i_level == 1``); stage i works on level i + 1."""
        nd = self.num_down_stages
        levels = {i: nag[i] for i in range(1, nd + 2)}
        sizes = {i: _get(lv, "pos").shape[0] for i, lv in levels.items()}
        _adopt_sub_views([levels[i] for i in range(1, nd + 2)])
        top = nd + 1

        def hf(i_level, k):
            """node / edge handcrafted features of a level through their MLPs (spt.py:786-797,
            823-835)"""
            lv = levels[i_level]
            ni, ei = _get(lv, "batch"), _get(lv, "edge_index")
            xh, ea = _get(lv, "x"), _get(lv, "edge_attr")
            if self.node_mlps[k] is not None and xh is not None:
                xh = self.node_mlps[k](xh, batch=ni, batch_size=B)
            if self.h_edge_mlps[k] is not None and ea is not None:
                ea = self.h_edge_mlps[k](ea, batch=None if (ni is None or B == 1) else ni[ei[0]],
                                         batch_size=B)
            return xh, ea

        node_x, edge_attrs = {}, {}
        lv = levels[1]
        node_x[1], edge_attrs[1] = hf(1, 0)
        x, _ = self.first_stage(                              # spt.py:893-913
            node_x[1] if self.use_node_hf else None, _get(lv, "batch"), pos=_get(lv, "pos"),
            node_size=_get(lv, "node_size"),
            super_index=None if top == 1 else _get(lv, "super_index"),
            edge_index=_get(lv, "edge_index"), edge_attr=edge_attrs[1],
            num_super=None if top == 1 else sizes[2], num_graphs=B)
        down_outputs = [x]
        for i_stage in range(nd):
            i_level = i_stage + 2
            lv = levels[i_level]
            node_x[i_level], edge_attrs[i_level] = hf(i_level, i_stage + 1)
            is_last = i_level == top
            x, _ = self.down_stages[i_stage](                 # spt.py:915-930
                node_x[i_level] if self.use_node_hf else None, x, _get(lv, "batch"),
                _get(levels[i_level - 1], "super_index"), pos=_get(lv, "pos"),
                node_size=_get(lv, "node_size"),
                super_index=None if is_last else _get(lv, "super_index"),
                edge_index=_get(lv, "edge_index"), edge_attr=edge_attrs[i_level],
                v_edge_attr=self._v_edge_attr(i_stage, levels[i_level - 1], B),
                num_super=sizes[i_level], num_graphs=B,
                num_super_parent=None if is_last else sizes[i_level + 1])
            down_outputs.append(x)
        up_outputs = []
        for i_stage in range(self.num_up_stages):             # spt.py:860-868
            i_level = nd - i_stage
            lv = levels[i_level]
            x_skip = down_outputs[-(2 + i_stage)]
            xh = node_x.get(i_level) if self.use_node_hf else None
            x, _ = self.up_stages[i_stage](
                self.feature_fusion(x_skip, xh), x, _get(lv, "batch"), _get(lv, "super_index"),
                pos=_get(lv, "pos"), node_size=_get(lv, "node_size"),
                super_index=_get(lv, "super_index"), edge_index=_get(lv, "edge_index"),
                edge_attr=edge_attrs.get(i_level), num_super=sizes[i_level + 1], num_graphs=B)
            up_outputs.append(x)
        if self.output_stage_wise:
            return [x] + up_outputs[::-1][1:] + [down_outputs[-1]]
        return x
