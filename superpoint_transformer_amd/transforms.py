"""Per-batch on-device graph transforms of the hot path's "next" rows
(SURVEY.md 8f): on-the-fly horizontal edge features fused with the edge
symmetrisation and the self loops."""
import ctypes

import torch

from . import _lib

__all__ = ["horizontal_edge_features", "EDGE_FEATURE_COLUMNS", "NodeSize", "SampleSubNodes",
           "SampleSegments", "SampleEdges", "OnTheFlyHorizontalEdgeFeatures",
           "SampleRadiusSubgraphs", "OnTheFlyInstanceGraph", "segment_sampling_weights",
           "NAGRestrictSize", "MortonOrder", "morton_code"]

EDGE_FEATURE_COLUMNS = [
    "mean_off_x", "mean_off_y", "mean_off_z", "std_off_x", "std_off_y", "std_off_z",
    "mean_dist", "angle_source", "angle_target", "normal_angle", "log_length",
    "log_surface", "log_volume", "log_size", "centroid_dir_x", "centroid_dir_y",
    "centroid_dir_z", "centroid_dist"]


def horizontal_edge_features(edge_index, edge_attr, pos, normal, log_length, log_surface,
                             log_volume, log_size, add_self_loops=True):
    """``OnTheFlyHorizontalEdgeFeatures`` (all default keys) + ``NAGAddSelfLoops``
    of one level (src/transforms/graph.py:1135-1277, 1419-1452) in one kernel.

    ``edge_index`` [2,E] trimmed edges, ``edge_attr`` [E,7] (f16 on disk: cast
    here like the reference's ``.float()``), node attributes of the level.
    Returns (edge_index [2, 2E(+N)], edge_attr [2E(+N), 18])."""
    _lib.require_cuda(edge_index, edge_attr, pos)
    dev = pos.device
    se = edge_index.long().contiguous()
    e, n = se.shape[1], pos.shape[0]

    def f32(t, cols=None):
        t = t.detach().float().contiguous()
        return t.view(n, cols) if cols else t.view(-1)

    ea7 = edge_attr.detach().float().contiguous()
    if ea7.shape != (e, 7):
        raise ValueError(f"edge_attr must be [E,7], got {tuple(ea7.shape)}")
    etot = 2 * e + (n if add_self_loops else 0)
    ei_out = torch.empty((2, etot), dtype=torch.int64, device=dev)
    ea_out = torch.empty((etot, 18), dtype=torch.float32, device=dev)
    args = [f32(pos, 3), f32(normal, 3), f32(log_length), f32(log_surface), f32(log_volume),
            f32(log_size)]
    with torch.cuda.device(dev):
        st = _lib.lib.spt_horizontal_edge_features_f32(
            _lib.ptr(se), e, n, _lib.ptr(ea7), *[_lib.ptr(a) for a in args],
            int(add_self_loops), _lib.ptr(ei_out), _lib.ptr(ea_out), _lib.stream_ptr(dev))
    _lib.check(st, "spt_horizontal_edge_features_f32")
    # host knowledge of the layout just written: [i<j | j>i (| loops)] with edge i mirrored at
    # i + e - the attention backward takes its by-target stream from it instead of a second sort
    # (csr.EdgeCSR; checked on the device).  A later transform that thins the list
    # (SampleEdges with n >= 0, NAGRestrictSize) builds a new tensor without the attribute.
    ei_out._spt_mirror_pairs = e
    return ei_out, ea_out


def vertical_edge_features(child, parent):
    """``_on_the_fly_vertical_edge_features`` with all default keys
    (src/transforms/graph.py:1335-1416): ``child`` / ``parent`` are level objects (Data or
    dicts) holding pos, normal, log_length, log_surface, log_volume, log_size, and
    ``child.super_index``.  Returns ``v_edge_attr`` [num_child, 9] (one kernel)."""
    def get(d, k):
        v = d[k] if isinstance(d, dict) else getattr(d, k)
        return v
    sup = get(child, "super_index").long().contiguous()
    _lib.require_cuda(sup)
    n = sup.numel()
    dev = sup.device
    keys = ("pos", "normal", "log_length", "log_surface", "log_volume", "log_size")
    cargs = [get(child, k).detach().float().contiguous() for k in keys]
    pargs = [get(parent, k).detach().float().contiguous() for k in keys]
    out = torch.empty((n, 9), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_vertical_edge_features_f32(
            _lib.ptr(sup), n, *[_lib.ptr(a) for a in cargs], *[_lib.ptr(a) for a in pargs],
            _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(st, "spt_vertical_edge_features_f32")
    return out


# ---------------------------------------------------------------------------
# Per-batch on-device transforms of the training pipeline
# (configs/datamodule/semantic/default.yaml:206-290), on the NAG mirror of data.py
# ---------------------------------------------------------------------------


class NodeSize:
    """``node_size`` of every level above ``low`` = number of ``low``-level nodes it
    holds (src/transforms/graph.py:1475-1498)."""

    def __init__(self, low=0):
        self.low = low

    def __call__(self, nag):
        for i in range(self.low + 1, nag.num_levels):
            nag[i].node_size = nag.get_sub_size(i, low=self.low)
        return nag


class SampleSubNodes:
    """Keep ``n_min``..``n_max`` random ``low``-level nodes of every ``high``-level
    segment (src/transforms/sampling.py:656-715)."""

    def __init__(self, high=1, low=0, n_max=32, n_min=16, mask=None, seed=None):
        self.high, self.low, self.n_max, self.n_min, self.mask, self.seed = \
            high, low, n_max, n_min, mask, seed

    def __call__(self, nag):
        if self.low == self.high:
            return nag
        idx = nag.get_sampling(high=self.high, low=self.low, n_max=self.n_max, n_min=self.n_min,
                               mask=self.mask, return_pointers=False, seed=self.seed)
        return nag.select(self.low, idx)


def segment_sampling_weights(nag, i_level, by_size=False, by_class=False):
    """Probability of drawing each level-``i_level`` segment (sampling.py:771-798 and
    895-921, the same lines twice): uniform, plus - ``by_size`` - the cube root of the number
    of points it holds, plus - ``by_class`` - the rarity of the rarest class it contains
    (``y`` = per-segment label histogram), each term normalised to sum 1 before it is added."""
    data = nag[i_level]
    w = torch.ones(data.num_nodes, device=nag.device)
    if by_size:
        sw = nag.get_sub_size(i_level, low=0) ** 0.333
        w = w + sw / sw.sum()
    if by_class and "y" in data:
        scores = 1 / (data.y.sum(dim=0).sqrt() + 1)
        scores = scores / scores.sum()
        cw = (data.y.gt(0) * scores.view(1, -1)).max(dim=1).values
        w = w + (cw / cw.sum()).squeeze()
    return w / w.sum()


class SampleSegments:
    """Drop a ``ratio`` of the segments of every level >= 1, from the top level down
    (src/transforms/sampling.py:718-807)."""

    def __init__(self, ratio=0.2, by_size=False, by_class=False):
        assert isinstance(ratio, list) and all(0 <= r < 1 for r in ratio) or (0 <= ratio < 1)
        self.ratio, self.by_size, self.by_class = ratio, by_size, by_class

    def __call__(self, nag):
        L = nag.num_levels
        ratio = self.ratio if isinstance(self.ratio, list) else [self.ratio] * (L - 1)
        for i in range(L - 1, 0, -1):
            if ratio[i - 1] <= 0:
                continue
            n = nag[i].num_nodes
            keep = n - int(n * ratio[i - 1])
            w = segment_sampling_weights(nag, i, self.by_size, self.by_class)
            idx = torch.multinomial(w, keep, replacement=False)
            nag = nag.select(i, idx)
        return nag


class SampleEdges:
    """Keep ``n_min``..``n_max`` random outgoing edges per source node
    (src/transforms/sampling.py:1234-1312) on the levels ``levels``."""

    def __init__(self, levels=(1, 2), n_min=16, n_max=32, seed=None):
        self.levels, self.n_min, self.n_max, self.seed = tuple(levels), n_min, n_max, seed

    def __call__(self, nag):
        from .segment import sparse_sample
        for i in self.levels:
            if i >= nag.num_levels or self.n_min < 0 or self.n_max < 0:
                continue
            d = nag[i]
            if not d.has_edges:
                continue
            idx = sparse_sample(d.edge_index[0], n_max=self.n_max, n_min=self.n_min,
                                seed=self.seed, num_segments=d.num_nodes)
            d.edge_index = d.edge_index[:, idx]
            for key in ["edge_attr"] + d.edge_keys:
                if key in d:
                    d[key] = d[key][idx]
        return nag


class OnTheFlyHorizontalEdgeFeatures:
    """18-D edge features from the stored 7-D ones + node attributes, edges doubled, and
    (``add_self_loops``) ``NAGAddSelfLoops`` fused (src/transforms/graph.py:1063-1277,
    1419-1452)."""

    def __init__(self, add_self_loops=True, use_mean_normal=False):
        self.add_self_loops = add_self_loops
        self.normal_key = "mean_normal" if use_mean_normal else "normal"

    def __call__(self, nag):
        for i in range(1, nag.num_levels):
            d = nag[i]
            if not d.has_edges:
                continue
            d.edge_index, d.edge_attr = horizontal_edge_features(
                d.edge_index, d.edge_attr, d.pos, d[self.normal_key], d.log_length, d.log_surface,
                d.log_volume, d.log_size, add_self_loops=self.add_self_loops)
        return nag


class SampleRadiusSubgraphs:
    """Keep the level-``i_level`` nodes within ``r`` of ``k`` random seed nodes (at most
    the ``k_max`` nearest per seed), each neighbourhood as its own batch item when
    ``disjoint`` (src/transforms/sampling.py:810-1000, 1094-1230).  Seeds are drawn like
    the reference (``torch.multinomial`` over ``segment_sampling_weights``, spread over the
    batch items when the level carries a ``batch``); ``idx_seed`` overrides the draw."""

    def __init__(self, r=2, k_max=10000, i_level=1, k=1, use_batch=True, disjoint=False,
                 cylindrical=False, idx_seed=None, by_size=False, by_class=False):
        self.r, self.k_max, self.i_level, self.k = r, k_max, i_level, k
        self.use_batch, self.disjoint, self.cylindrical, self.idx_seed = \
            use_batch, disjoint, cylindrical, idx_seed
        self.by_size, self.by_class = by_size, by_class

    def _seeds(self, data, k, w):
        batch = data.batch if "batch" in data else None
        if batch is None or not self.use_batch:
            return torch.multinomial(w, k, replacement=False)
        ids = batch.unique()
        ids = ids[torch.randperm(ids.numel())]
        kb = max(k // ids.numel(), 1)
        out, done = [], 0
        for step, b in enumerate(ids):
            if step >= ids.numel() - 1:
                kb = k - done
            m = torch.where(batch == b)[0]
            out.append(m[torch.multinomial(w[m], kb, replacement=False)])
            done += kb
            if done >= k:
                break
        return torch.cat(out)

    def _ball(self, data, seed):
        from .ops import _workspace
        dev = data.device
        pos = data.pos.detach().float().contiguous()
        n = pos.shape[0]
        batch = data.batch.long().contiguous() if "batch" in data else None
        c = pos[seed].cpu()
        center = (ctypes.c_float * 3)(float(c[0]), float(c[1]), float(c[2]))
        out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        nbytes = _lib.lib.spt_radius_ball_workspace_bytes(n)
        ws = _workspace(nbytes, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_radius_ball_f32(
                _lib.ptr(pos), n, ctypes.cast(center, ctypes.c_void_p), float(self.r),
                int(self.cylindrical), _lib.ptr(batch), int(batch[seed]) if batch is not None else 0,
                _lib.ptr(out), _lib.ptr(count), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
        _lib.check(st, "spt_radius_ball_f32")
        idx = out[:int(count)]
        if idx.numel() > self.k_max:                      # keep the k_max nearest (rare)
            w = torch.tensor([1.0, 1.0, 0.0 if self.cylindrical else 1.0], device=dev)
            d = ((pos[idx] - pos[seed]) * w).norm(dim=1)
            idx = idx[torch.topk(d, self.k_max, largest=False).indices].sort().values
        return idx

    def __call__(self, nag):
        from .data import NAG
        if self.i_level is None or self.k <= 0 or self.r is None or self.r <= 0:
            return nag
        i_level = nag.num_levels - 1 if self.i_level == -1 else self.i_level
        data = nag[i_level]
        k = self.k if self.k < data.num_nodes else 1
        seeds = self.idx_seed if self.idx_seed is not None else self._seeds(
            data, k, segment_sampling_weights(nag, i_level, self.by_size, self.by_class))
        balls = [self._ball(data, int(s)) for s in seeds]
        if self.disjoint:
            return NAG.from_nag_list([nag.select(i_level, idx) for idx in balls])
        return nag.select(i_level, torch.unique(torch.cat(balls)))


def _per_level(level, default, value, num_levels, start=0):
    """``fill_list_with_string_indexing`` (src/utils/list.py:46-91): ``value`` on the levels
    ``level`` names - an int, 'all', 'i+' (level i and above) or 'i-' (the levels BELOW i) - and
    ``default`` elsewhere."""
    out = [default] * num_levels
    if isinstance(level, int):
        out[level] = value
    elif level == "all":
        out[start:] = [value] * (num_levels - start)
    elif level[-1] == "+":
        i = int(level[:-1])
        out[i:] = [value] * (num_levels - i)
    elif level[-1] == "-":
        i = int(level[:-1])
        out[:i] = [value] * i
    else:
        raise ValueError(f"Unsupported level={level}")
    return out


class NAGRestrictSize:
    """Cap the number of nodes and / or edges of the levels ``level`` by uniform random removal
    (src/transforms/sampling.py:1346-1423; in the training pipeline after the samplers, with
    ``level='1+'``): surplus nodes go through ``NAG.select`` (their descendants and ancestors
    follow), surplus edges are dropped with their attributes."""

    def __init__(self, level="1+", num_nodes=0, num_edges=0):
        self.level, self.num_nodes, self.num_edges = level, num_nodes, num_edges

    def __call__(self, nag):
        L = nag.num_levels
        nodes = _per_level(self.level, -1, self.num_nodes, L)
        edges = _per_level(self.level, -1, self.num_edges, L)
        for i in range(L):
            nag = self._restrict_level(nag, i, nodes[i], edges[i])
        return nag

    @staticmethod
    def _restrict_level(nag, i_level, num_nodes, num_edges):
        d = nag[i_level]
        if d.num_nodes > num_nodes and num_nodes > 0:
            idx = torch.multinomial(torch.ones(d.num_nodes, device=nag.device), num_nodes,
                                    replacement=False)
            nag = nag.select(i_level, idx)
            d = nag[i_level]
        if d.num_edges > num_edges and num_edges > 0:
            idx = torch.multinomial(torch.ones(d.num_edges, device=nag.device), num_edges,
                                    replacement=False)
            keys = [k for k in ["edge_attr"] + d.edge_keys if k in d]      # before the cut
            d.edge_index = d.edge_index[:, idx]
            for key in keys:
                d[key] = d[key][idx]
        return nag


class OnTheFlyInstanceGraph:
    """Targets of the panoptic heads, built every batch (src/transforms/instance.py:18-257):
    the trimmed graph between the level-``level`` segments (``obj_edge_index``: the existing
    edges, or segments with points / centroids within ``radius``), the affinity of each edge
    from the segments' overlaps with the annotated objects (``obj_edge_affinity``,
    ``InstanceData.instance_graph``) and the position of each segment's target object
    (``obj_pos``)."""

    _ADJACENCY_MODES = ["available", "radius-atomic", "radius-centroid"]
    _CENTROID_MODES = ["iou", "ratio-product"]

    def __init__(self, level=1, num_classes=None, adjacency_mode="radius-atomic", k_max=30,
                 radius=1, use_batch=True, centroid_mode="iou", centroid_level=1,
                 smooth_affinity=True):
        assert adjacency_mode.lower() in self._ADJACENCY_MODES, \
            f"Expected 'mode' to be one of {self._ADJACENCY_MODES}"
        assert centroid_mode.lower() in self._CENTROID_MODES, \
            f"Expected 'mode' to be one of {self._CENTROID_MODES}"
        self.level, self.num_classes = level, num_classes
        self.adjacency_mode, self.k_max, self.radius = adjacency_mode.lower(), k_max, radius
        self.use_batch, self.centroid_mode = use_batch, centroid_mode.lower()
        self.centroid_level, self.smooth_affinity = centroid_level, smooth_affinity

    def __call__(self, nag):
        from .graph import to_trimmed
        from .instance import _consecutive
        from .neighbors import cluster_radius_nn_graph, knn_1_graph
        if self.level is None or self.level < 0:
            return nag
        if not 0 <= self.level < nag.num_levels:
            raise AssertionError(f"level {self.level} is not in the NAG")
        data = nag[self.level]
        batch = data.batch if (self.use_batch and "batch" in data) else None
        if self.adjacency_mode == "available":
            obj_edge_index = data.edge_index if data.has_edges else None
        elif self.adjacency_mode == "radius-atomic":
            obj_edge_index, _ = cluster_radius_nn_graph(
                nag[0].pos, nag.get_super_index(self.level, low=0), k_max=self.k_max,
                gap=self.radius, batch=batch)
        else:
            obj_edge_index, _ = knn_1_graph(data.pos, self.k_max, r_max=self.radius, batch=batch)
        if obj_edge_index is None:
            obj_edge_index = torch.empty(2, 0, dtype=torch.long, device=data.device)

        if "obj" not in data:                               # no annotations: graph only
            data.obj_edge_index = to_trimmed(obj_edge_index)
            data.obj_edge_affinity = None
            data.obj_pos = None
            return nag

        data.obj_edge_index, data.obj_edge_affinity = data.obj.instance_graph(
            obj_edge_index, num_classes=self.num_classes, smooth_affinity=self.smooth_affinity)

        # position of each segment's target object: objects' centroids estimated at
        # ``centroid_level``, looked up through a common dense numbering of the object ids
        i_level = min(self.centroid_level, nag.num_levels - 1)
        # 'ratio-product' is the transform's name (transforms/instance.py:101-119) for what
        # InstanceData.estimate_centroid implements as 'product-iou' (data/instance.py:337): the
        # reference forwards the string unchanged and raises there; it is mapped here
        mode = "product-iou" if self.centroid_mode == "ratio-product" else self.centroid_mode
        obj_pos, obj_idx = nag[i_level].estimate_instance_centroid(mode=mode)
        sp_obj_idx = data.obj.major(num_classes=self.num_classes)[0]
        joint = torch.cat((sp_obj_idx, obj_idx))
        dense = _consecutive(joint)[0]
        data.obj_pos = obj_pos[dense[:sp_obj_idx.numel()]]
        return nag


def morton_code(pos, bits=10):
    """``bits``-per-axis Morton (Z-order) code of [n, 3] positions inside their bounding box (int64)."""
    lo, hi = pos.min(0).values, pos.max(0).values
    top = (1 << bits) - 1
    q = ((pos - lo) / (hi - lo).clamp_min(1e-9) * float(top)).long().clamp_(0, top)
    code = torch.zeros(pos.shape[0], dtype=torch.int64, device=pos.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


class MortonOrder:
    """Load-time re-ordering of a NAG for MEMORY LOCALITY (round 5; not a transform of the
    reference - its datasets store nodes in the order the partition emitted them, spatial neighbours
    ~0.23 N apart in the demo room).  The top level is sorted by (cloud, Morton code of its
    position); every level below is sorted by its parent (in the parent's new numbering) and,
    inside a parent, by its own Morton code - each step one ``NAG.select`` with a permutation
    (src/data/nag.py:306-399), which re-indexes edges, ``super_index`` and ``sub`` consistently.  Afterwards the nodes a superpoint-graph edge
    joins (built by radius search, src/transforms/graph.py:193-321: spatial neighbours) are close
    in memory at every level, the children of a superpoint are contiguous (the pool's CSR view is
    the identity: coalesced streams instead of gathers), and the attention's k / v / record gathers
    hit rows that are still in L2.

    Purely a renumbering: edges, ``super_index``, ``sub`` and every per-node attribute follow.
    ``nag[i].morton_origin`` [n_i] = the ORIGINAL index of every node, so per-node outputs go
    back with ``MortonOrder.restore`` (``out_original[origin] = out``); results are those of the
    un-ordered NAG up to the summation order of segment / attention reductions."""

    KEY = "morton_origin"

    def __init__(self, bits=10):
        self.bits = int(bits)

    def __call__(self, nag):
        nag = nag.clone()
        L = nag.num_levels
        for i in range(L):
            n = nag[i].num_nodes
            nag[i][self.KEY] = torch.arange(n, device=nag.device)
        top = L - 1
        d = nag[top]
        key = morton_code(d.pos, self.bits)
        if d.batch is not None:
            key = key + (d.batch.long() << (3 * self.bits))
        nag = nag.select(top, torch.argsort(key, stable=True))
        # (NAG.select re-indexes the levels below a re-ordered level but leaves their nodes where
        # they are: every level is sorted explicitly - by parent, then by its own Morton code;
        # the points of level 0 by parent only, in their stored order)
        for i in range(top - 1, -1, -1):
            d = nag[i]
            key = d.super_index.long() << (3 * self.bits)
            if i > 0:
                key = key + morton_code(d.pos, self.bits)
            nag = nag.select(i, torch.argsort(key, stable=True))
        return nag

    @classmethod
    def restore(cls, nag, out, i_level):
        """Per-node tensor ``out`` [n_i, ...] of the re-ordered level ``i_level`` back in the
        original node order."""
        origin = nag[i_level][cls.KEY]
        back = torch.empty_like(out)
        back[origin] = out
        return back
