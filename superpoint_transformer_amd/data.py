"""Device-side NAG / Data / Cluster containers with the reference's selection
mechanism.

Light mirrors of ``src.data.{Cluster, Data, NAG}`` holding only what the hot path
and its neighbours need - attributes, the cluster CSR, edges - with the same
``select`` contract, argument names and return tuples
(src/data/cluster.py:79-140, src/data/data.py:286-470, src/data/nag.py:306-399,
672-711), executed by the kernels of ``csrc/select.hip`` and ``csrc/sampling.hip``.
Instance labels (``obj``: an ``instance.InstanceData``) follow selection and batching.
HDF5 reading / writing lives in ``h5io``.  Not mirrored: visualisation.
"""
import copy

import numpy as np
import torch

from . import _lib
from .csr import build_csr
from .ops import _workspace, segment_sum_i64
from .segment import sparse_sample

__all__ = ["Cluster", "Data", "NAG", "consecutive_cluster", "tensor_idx"]


def tensor_idx(idx, device):
    """int / list / slice-free numpy / bool mask / tensor -> 1-D LongTensor on ``device``
    (src/utils/tensor.py:tensor_idx); ``None`` stays ``None``."""
    if idx is None:
        return None
    if isinstance(idx, int):
        idx = [idx]
    if isinstance(idx, np.ndarray):
        idx = torch.from_numpy(idx)
    idx = torch.as_tensor(idx, device=device)
    if idx.dtype == torch.bool:
        idx = torch.where(idx)[0]
    return idx.long().view(-1).contiguous()


def _is_arange(idx, n):
    return idx.numel() == n and bool((idx == torch.arange(n, device=idx.device)).all())


def _count(t):
    return int(t.item())


def _is_instance_data(item):
    from .instance import InstanceData                 # instance.py imports this module
    return isinstance(item, InstanceData)


def consecutive_cluster(src, num_labels=None, gather=None):
    """``torch_geometric.nn.pool.consecutive.consecutive_cluster`` for labels in
    ``[0, num_labels)``: ``(inv, uniques)`` with ``inv`` the dense relabelling in sorted
    order and ``uniques`` the sorted labels present.  (PyG returns as second output the
    position of one representative per label; every call site of the reference only
    uses it as ``src[perm]`` = ``uniques``, cluster.py:130-131, data.py:404-405.)
    ``gather``: relabel ``src[gather]`` without materialising it."""
    _lib.require_cuda(src)
    src = src.long().contiguous()
    dev = src.device
    k = src.numel() if gather is None else gather.numel()
    if num_labels is None:
        num_labels = int(src.max()) + 1 if src.numel() else 0
    inv = torch.empty(k, dtype=torch.int64, device=dev)
    uniq = torch.empty(max(min(k, num_labels), 1), dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    nbytes = _lib.lib.spt_relabel_consecutive_workspace_bytes(num_labels)
    ws = _workspace(nbytes, dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_relabel_consecutive(
            _lib.ptr(src), _lib.ptr(gather), k, num_labels, _lib.ptr(inv), _lib.ptr(uniq),
            _lib.ptr(count), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
    _lib.check(st, "spt_relabel_consecutive")
    return inv, uniq[:_count(count)]


class Cluster:
    """CSR of the points of each cluster: ``points[pointers[c]:pointers[c+1]]``
    (src/data/cluster.py:19-58).  ``dense=True``: build it from a per-point cluster
    index (CSRData.__init__, src/data/csr.py:85-88)."""

    def __init__(self, pointers, points, dense=False, ascending=None):
        # ascending: the points of every cluster are in increasing order (True / False / None =
        # not known yet, checked on first use) - what lets the segment kernels take this CSR
        # as the view of the level's super_index instead of sorting it (csr.adopt_csr)
        self._ascending = ascending
        if dense:
            index = pointers.long()
            n = int(index.max()) + 1 if index.numel() else 0
            csr = build_csr(index, max(n, 1))
            self.pointers = csr.rowptr.long()[:n + 1] if n else torch.zeros(
                1, dtype=torch.int64, device=index.device)
            self.points = points[csr.perm.long()]
        else:
            self.pointers = pointers.long().contiguous()
            self.points = points.long().contiguous()

    @property
    def ascending(self):
        """True when the points of every cluster are stored in increasing order.  Checked once
        per object (one device pass + a host read); ``select`` and ``clone`` hand it on."""
        if self._ascending is None:
            m = self.points.numel()
            if m < 2:
                self._ascending = True
            else:
                ok = self.points[1:] > self.points[:-1]
                starts = self.pointers[1:-1]
                starts = starts[(starts > 0) & (starts < m)]
                ok[starts - 1] = True
                self._ascending = bool(ok.all())
        return self._ascending

    @property
    def device(self):
        return self.pointers.device

    @property
    def num_clusters(self):
        return self.pointers.numel() - 1

    @property
    def num_points(self):
        return self.points.numel()

    @property
    def sizes(self):
        return self.pointers[1:] - self.pointers[:-1]

    def to_super_index(self):
        """Per-point cluster index (cluster.py:67-77)."""
        out = torch.empty(self.num_points, dtype=torch.int64, device=self.device)
        out[self.points] = torch.arange(self.num_clusters, device=self.device).repeat_interleave(
            self.sizes)
        return out

    def clone(self):
        return Cluster(self.pointers.clone(), self.points.clone(), ascending=self._ascending)

    def to(self, device):
        return Cluster(self.pointers.to(device), self.points.to(device), ascending=self._ascending)

    def select(self, idx, update_sub=True, num_sub=None):
        """New Cluster made of the clusters ``idx`` (no duplicates), and - when
        ``update_sub`` - ``(idx_sub, sub_super)``: the surviving points (to index the
        level below) and their new cluster (the level below's new ``super_index``),
        with ``points`` relabelled to dense ids (cluster.py:79-140)."""
        idx = tensor_idx(idx, self.device)
        if idx is None or _is_arange(idx, self.num_clusters):
            return self.clone(), (None, None)
        dev = self.device
        k, m = idx.numel(), self.num_points
        if num_sub is None:
            num_sub = int(self.points.max()) + 1 if m else 0
        new_ptr = torch.empty(k + 1, dtype=torch.int64, device=dev)
        new_pts = torch.empty(max(m, 1), dtype=torch.int64, device=dev)
        idx_sub = torch.empty(max(num_sub, 1), dtype=torch.int64, device=dev)
        sub_super = torch.empty(max(num_sub, 1), dtype=torch.int64, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        nbytes = _lib.lib.spt_cluster_select_workspace_bytes(k, m, num_sub)
        ws = _workspace(nbytes, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_cluster_select(
                _lib.ptr(self.pointers), _lib.ptr(self.points), m, _lib.ptr(idx), k, num_sub,
                _lib.ptr(new_ptr), _lib.ptr(new_pts), _lib.ptr(idx_sub), _lib.ptr(sub_super),
                _lib.ptr(count), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
        _lib.check(st, "spt_cluster_select")
        kept = _count(count)                          # == new_ptr[-1]: idx has no duplicates
        if not update_sub:
            # CSRData.select only: same clusters, ORIGINAL point ids
            old = idx_sub[:kept][new_pts[:kept]]
            return Cluster(new_ptr, old, ascending=self._ascending), (None, None)
        # the kernel copies every kept cluster in its stored order and relabels the points
        # by rank: an ascending cluster stays ascending
        return (Cluster(new_ptr, new_pts[:kept], ascending=self._ascending),
                (idx_sub[:kept], sub_super[:kept]))


class Data:
    """Attribute store of one NAG level (subset of src/data/data.py): node tensors
    (first dim = num_nodes), ``edge_index`` [2,E] + ``edge_*`` tensors, ``super_index``
    (cluster of each node in the level above), ``sub`` (Cluster over the level below)."""

    _NODE_HINT = ("pos", "x", "rgb", "y", "super_index", "node_size", "normal")
    # attributes that read as None when absent: the reference's Data answers None for its own
    # optional properties (src/data/data.py:64-101) and PyG's Data for x / y / edge_index /
    # edge_attr, and callers test `data.obj_pos is None` (a transform that assigns None to a
    # key leaves it absent here: __setattr__ pops it)
    _OPTIONAL = frozenset((
        "pos", "rgb", "obj", "semantic_pred", "neighbor_index", "sub", "super_index",
        "v_edge_attr", "x", "y", "edge_index", "edge_attr", "batch", "obj_pos",
        "obj_edge_index", "obj_edge_affinity", "node_size", "normal"))

    def __init__(self, **attrs):
        object.__setattr__(self, "_store", {})
        object.__setattr__(self, "_num_nodes", None)
        for k, v in attrs.items():
            setattr(self, k, v)

    def __getattr__(self, key):
        store = object.__getattribute__(self, "_store")
        if key in store:
            return store[key]
        if key in Data._OPTIONAL:
            return None
        raise AttributeError(key)

    def __setattr__(self, key, value):
        if key == "num_nodes":
            object.__setattr__(self, "_num_nodes", value)
        elif value is None:
            self._store.pop(key, None)
        else:
            self._store[key] = value

    def __getitem__(self, key):
        """Strict: a key that is not stored raises ``KeyError`` (only ATTRIBUTE access answers
        None for the reference's optional properties)."""
        store = object.__getattribute__(self, "_store")
        if key in store:
            return store[key]
        raise KeyError(key)

    def get(self, key, default=None):
        return object.__getattribute__(self, "_store").get(key, default)

    __setitem__ = __setattr__

    def __contains__(self, key):
        return key in self._store

    def __iter__(self):
        return iter(list(self._store.items()))

    @property
    def keys(self):
        return list(self._store)

    @property
    def device(self):
        for v in self._store.values():
            if torch.is_tensor(v):
                return v.device
            if isinstance(v, Cluster) or _is_instance_data(v):
                return v.device
        return torch.device("cpu")

    @property
    def num_nodes(self):
        if self._num_nodes is not None:
            return self._num_nodes
        for k in self._NODE_HINT:
            if k in self._store:
                return self._store[k].shape[0]
        if "sub" in self._store:
            return self._store["sub"].num_clusters
        raise ValueError("cannot infer num_nodes: set data.num_nodes")

    @property
    def has_edges(self):
        return "edge_index" in self._store

    @property
    def num_edges(self):
        return self.edge_index.shape[1] if self.has_edges else 0

    @property
    def is_super(self):
        return "sub" in self._store

    @property
    def is_sub(self):
        return "super_index" in self._store

    @property
    def edge_keys(self):
        return [k for k in self._store if k.startswith("edge_") and k not in ("edge_index", "edge_attr")]

    @property
    def v_edge_keys(self):
        return [k for k in self._store if k.startswith("v_edge_")]

    def clone(self):
        out = Data()
        for k, v in self:
            out[k] = v.clone() if hasattr(v, "clone") else copy.deepcopy(v)
        out.num_nodes = self._num_nodes
        return out

    def select(self, idx, update_sub=True, update_super=True, num_sub=None, num_super=None):
        """``(data, (idx_sub, sub_super), (idx_super, super_sub))`` - data.py:286-470.
        ``num_sub`` / ``num_super``: sizes of the levels below / above when known (else
        read back from the index maxima)."""
        dev = self.device
        idx = tensor_idx(idx, dev)
        n = self.num_nodes
        if idx is None or _is_arange(idx, n):
            return self.clone(), (None, None), (None, None)
        k = idx.numel()
        data = Data()
        idx_edge = None
        if self.has_edges:                                            # data.py:360-373
            ei = self.edge_index.long().contiguous()
            E = ei.shape[1]
            inv = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
            out_e = torch.empty((2, max(E, 1)), dtype=torch.int64, device=dev)
            idx_e = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
            count = torch.zeros(1, dtype=torch.int64, device=dev)
            nbytes = _lib.lib.spt_select_edges_workspace_bytes(E)
            ws = _workspace(nbytes, dev)
            with torch.cuda.device(dev):
                st = _lib.lib.spt_index_inverse(_lib.ptr(idx), k, n, _lib.ptr(inv), _lib.stream_ptr(dev))
                _lib.check(st, "spt_index_inverse")
                st = _lib.lib.spt_select_edges(
                    _lib.ptr(ei), E, E, _lib.ptr(inv), n, _lib.ptr(out_e), out_e.shape[1],
                    _lib.ptr(idx_e), _lib.ptr(count), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
            _lib.check(st, "spt_select_edges")
            kept = _count(count)
            idx_edge = idx_e[:kept]
            data.edge_index = out_e[:, :kept]

        out_sub = (None, None)
        if self.is_super:                                             # data.py:379-388
            data.sub, out_sub = self.sub.select(idx, update_sub=update_sub, num_sub=num_sub)

        out_super = (None, None)
        if self.is_sub and not update_super:
            data.super_index = self.super_index[idx]
        if self.is_sub and update_super:                              # data.py:399-416
            new_si, idx_super = consecutive_cluster(self.super_index, num_super, gather=idx)
            data.super_index = new_si
            super_sub = Cluster(new_si, torch.arange(k, device=dev), dense=True, ascending=True)
            out_super = (idx_super, super_sub)

        skip = {"edge_index", "sub", "super_index", "neighbor_index", "neighbor_distance"}
        n_e = self.num_edges
        for key, item in self:                                        # data.py:419-462
            if key in skip:
                continue
            if isinstance(item, Cluster):
                data[key] = item.select(idx, update_sub=False)[0]
                continue
            if _is_instance_data(item):                               # data.py:437-439
                data[key] = item.select(idx)
                continue
            is_tensor = torch.is_tensor(item)
            node_sized = is_tensor and item.dim() > 0 and item.shape[0] == n
            edge_sized = is_tensor and item.dim() > 0 and item.shape[0] == n_e
            if node_sized and key in self.v_edge_keys:
                data[key] = item[idx]
            elif self.has_edges and edge_sized and key in ["edge_attr"] + self.edge_keys:
                data[key] = item[idx_edge]
            elif node_sized:
                data[key] = item[idx]
            else:
                data[key] = copy.deepcopy(item)
        data.num_nodes = k
        return data, out_sub, out_super


    def estimate_instance_centroid(self, mode="iou"):
        """``(obj_pos, obj_idx)`` of the target objects from the positions of the clusters
        overlapping them (data.py:941-974)."""
        if "obj" not in self._store:
            return None, None
        return self.obj.estimate_centroid(self.pos, mode=mode)


class NAG:
    """Nested hierarchy of ``Data`` levels (src/data/nag.py), level 0 = points."""

    def __init__(self, data_list):
        self._list = list(data_list)

    def __getitem__(self, i):
        return self._list[i]

    def __len__(self):
        return len(self._list)

    @property
    def num_levels(self):
        return len(self._list)

    @property
    def num_points(self):
        return [d.num_nodes for d in self._list]

    @property
    def device(self):
        return self._list[0].device

    def clone(self):
        return NAG([d.clone() for d in self._list])

    def get_super_index(self, high, low=0):
        """Index of the ``high``-level cluster of every ``low``-level node
        (nag.py:113-131)."""
        si = self[low].super_index
        for i in range(low + 1, high):
            si = self[i].super_index[si]
        return si

    def get_sub_size(self, high, low=0):
        """Number of ``low``-level nodes in every ``high``-level cluster
        (nag.py:59-110)."""
        size = None
        for i in range(low, high):
            si = self[i].super_index
            w = torch.ones_like(si) if size is None else size
            size = segment_sum_i64(w, si, self[i + 1].num_nodes)
        return size

    def get_sampling(self, high=1, low=0, n_max=32, n_min=1, mask=None, return_pointers=False,
                     seed=None):
        """nag.py:672-711."""
        return sparse_sample(self.get_super_index(high, low), n_max=n_max, n_min=n_min, mask=mask,
                             return_pointers=return_pointers, seed=seed,
                             num_segments=self[high].num_nodes)

    @classmethod
    def from_nag_list(cls, nag_list):
        """One NAG out of several (NAGBatch.from_nag_list, nag.py:878-898 +
        Batch.from_data_list): per level, attributes concatenated, ``super_index`` /
        ``edge_index`` / ``sub`` shifted by the node counts of the preceding items, and a
        ``batch`` vector telling which item each node came from."""
        L = nag_list[0].num_levels
        dev = nag_list[0].device
        counts = [[n[l].num_nodes for n in nag_list] for l in range(L)]
        out = []
        for l in range(L):
            items = [n[l] for n in nag_list]
            off = [0]
            for c in counts[l]:
                off.append(off[-1] + c)
            d = Data()
            for key in items[0].keys:
                vals = [it[key] for it in items]
                if key == "super_index":
                    up = [0]
                    for c in counts[l + 1]:
                        up.append(up[-1] + c)
                    d[key] = torch.cat([v + up[j] for j, v in enumerate(vals)])
                elif key == "edge_index":
                    d[key] = torch.cat([v + off[j] for j, v in enumerate(vals)], dim=1)
                elif key == "sub":
                    lo = [0]
                    for c in counts[l - 1]:
                        lo.append(lo[-1] + c)
                    ptr = [vals[0].pointers]
                    base = int(vals[0].pointers[-1])
                    for v in vals[1:]:
                        ptr.append(v.pointers[1:] + base)
                        base += int(v.pointers[-1])
                    asc = [v._ascending for v in vals]
                    d[key] = Cluster(torch.cat(ptr),
                                     torch.cat([v.points + lo[j] for j, v in enumerate(vals)]),
                                     ascending=True if all(a is True for a in asc) else
                                     (False if any(a is False for a in asc) else None))
                elif _is_instance_data(vals[0]):
                    d[key] = type(vals[0]).from_list(vals)
                elif torch.is_tensor(vals[0]):
                    d[key] = torch.cat(vals, dim=0)
                else:
                    d[key] = copy.deepcopy(vals[0])
            d.batch = torch.repeat_interleave(
                torch.arange(len(nag_list), device=dev),
                torch.tensor(counts[l], device=dev))
            # host knowledge of the batch layout (Batch.ptr in the reference): the fused layers'
            # run tables come from it without reading the device tensor back (ops.graph_runs)
            d.batch._spt_host_ptr = list(off)
            d.num_nodes = off[-1]
            out.append(d)
        return cls(out)

    def select(self, i_level, idx):
        """New NAG keeping the nodes ``idx`` (no duplicates) of level ``i_level``, their
        descendants and their ancestors, every level re-indexed consistently
        (nag.py:306-399)."""
        idx = tensor_idx(idx, self.device)
        sizes = self.num_points
        if idx is None or _is_arange(idx, sizes[i_level]):
            return self.clone()
        L = self.num_levels
        out = [None] * L

        def below(i):
            return sizes[i - 1] if i > 0 else None

        def above(i):
            return sizes[i + 1] if i + 1 < L else None

        out[i_level], out_sub, out_super = self[i_level].select(
            idx, update_sub=True, update_super=True, num_sub=below(i_level),
            num_super=above(i_level))
        for i in range(i_level - 1, -1, -1):                          # nag.py:357-368
            idx_sub, sub_super = out_sub
            out[i], out_sub, _ = self[i].select(idx_sub, update_sub=True, update_super=False,
                                                num_sub=below(i))
            out[i].super_index = sub_super
        for i in range(i_level + 1, L):                               # nag.py:371-382
            idx_super, super_sub = out_super
            out[i], _, out_super = self[i].select(idx_super, update_sub=False, update_super=True,
                                                  num_super=above(i))
            # when every node of the level below survived in place (idx_super is None), the
            # existing cluster CSR is still valid: keep it (the reference assigns None here)
            if super_sub is not None:
                out[i].sub = super_sub
        return NAG(out)
