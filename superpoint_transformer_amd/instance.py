"""Cluster-object overlaps of a NAG level: the label side of the panoptic path
(src/data/instance.py).

``InstanceData`` is a CSR over the clusters (segments) of a level: for every cluster the
objects it overlaps (``obj``), the number of points in each overlap (``count``) and the
objects' semantic label (``y``).  ``OnTheFlyInstanceGraph`` (transforms.py) runs on it every
training batch of the panoptic configuration: ``major`` (the target object of each cluster),
``instance_graph`` (trimmed cluster graph + edge affinities, the target of the edge-affinity
head) and ``estimate_centroid``.  All of it is integer segment work over the segment-CSR
kernels (``spt_segcsr_sum_i64``, the arg-max reduce) and index plumbing; nothing here runs
without the HIP library.
"""
import torch

from .data import tensor_idx, _is_arange
from .graph import to_trimmed
from .shims.scatter_shim import scatter_max, scatter_sum

__all__ = ["InstanceData"]


def _consecutive(src):
    """``consecutive_cluster`` for arbitrary (sparse, large) int64 keys: ``(inv, perm)`` with
    ``inv`` the dense relabelling in sorted key order and ``perm[u]`` the LAST position holding
    key u (what PyG's CPU scatter_ leaves; every call site reads properties shared by all
    holders of a key)."""
    uniq, inv, counts = torch.unique(src, sorted=True, return_inverse=True, return_counts=True)
    order = torch.argsort(inv, stable=True)
    return inv, order[counts.cumsum(0) - 1]


class InstanceData:
    """``obj / count / y [pointers[c]:pointers[c+1]]`` = the overlaps of cluster c
    (instance.py:15-101).  ``dense=True``: ``pointers`` is the cluster index of each
    (possibly duplicated) cluster-object pair; duplicates are merged and their counts summed
    (instance.py:70-90), clusters without a pair get an empty row."""

    def __init__(self, pointers, obj, count, y, dense=False, **kwargs):
        pointers, obj, count, y = (t.long().contiguous() for t in (pointers, obj, count, y))
        if dense:
            index = pointers
            if index.numel() == 0:                       # sparse.py:27
                raise AssertionError("At least one group index is required.")
            key = index * (obj.max() + 1) + obj
            inv, perm = _consecutive(key)
            count = scatter_sum(count, inv, 0, None, perm.numel())
            # pairs now sorted by (cluster, obj), i.e. already grouped by cluster (csr.py:85-88)
            index, obj, y = index[perm], obj[perm], y[perm]
            sizes = torch.bincount(index, minlength=int(index.max()) + 1)
            if not bool((sizes > 0).all()):              # sparse.py:28
                raise AssertionError("Indices must be dense")
            pointers = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
        self.pointers, self.obj, self.count, self.y = pointers, obj, count, y
        self.pair_cropped_count = None

    # -- CSRData surface (src/data/csr.py:225-250) ----------------------------------------
    @property
    def device(self):
        return self.pointers.device

    @property
    def values(self):
        return [self.obj, self.count, self.y]

    @property
    def num_groups(self):
        return self.pointers.numel() - 1

    num_clusters = num_groups

    @property
    def num_items(self):
        return self.obj.numel()

    num_overlaps = num_items

    @property
    def num_obj(self):
        return self.obj.unique().numel()

    @property
    def sizes(self):
        return self.pointers[1:] - self.pointers[:-1]

    @property
    def indices(self):
        """cluster of each pair (csr.py:244-248)"""
        return torch.arange(self.num_groups, device=self.device).repeat_interleave(self.sizes)

    def __len__(self):
        return self.num_groups

    def clone(self):
        out = InstanceData(self.pointers.clone(), self.obj.clone(), self.count.clone(),
                           self.y.clone())
        if self.pair_cropped_count is not None:
            out.pair_cropped_count = self.pair_cropped_count.clone()
        return out

    def to(self, device):
        out = InstanceData(*(t.to(device) for t in (self.pointers, self.obj, self.count, self.y)))
        if self.pair_cropped_count is not None:
            out.pair_cropped_count = self.pair_cropped_count.to(device)
        return out

    def select(self, idx, **kwargs):
        """The clusters ``idx`` (no duplicates), in that order (csr.py:328-408).  An identity
        selection returns a copy (the reference's ``__getitem__`` returns an EMPTY object
        there, csr.py:371-378, which ``Data.select`` never reaches: data.py:331-337)."""
        idx = tensor_idx(idx, self.device)
        if idx is None or _is_arange(idx, self.num_groups):
            return self.clone()
        sizes = self.pointers[idx + 1] - self.pointers[idx]
        ptr = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
        total = int(ptr[-1])
        val = torch.arange(total, device=self.device)
        val = val - ptr[:-1].repeat_interleave(sizes) + self.pointers[idx].repeat_interleave(sizes)
        return InstanceData(ptr, self.obj[val], self.count[val], self.y[val])

    __getitem__ = select

    @classmethod
    def from_list(cls, items):
        """Batch of several InstanceData: clusters stacked, ``obj`` shifted past the largest
        object index of the preceding items so that objects of different items never collide
        (CSRBatch.from_list with ``is_index_value=[True, False, False]``, csr.py:676-745)."""
        dev = items[0].device
        ptr, obj = [torch.zeros(1, dtype=torch.int64, device=dev)], []
        base, shift = 0, 0
        for it in items:
            ptr.append(it.pointers[1:] + base)
            base += it.num_items
            obj.append(it.obj + shift)
            shift += int(it.obj.max()) + 1 if it.num_items else 0
        return cls(torch.cat(ptr), torch.cat(obj), torch.cat([it.count for it in items]),
                   torch.cat([it.y for it in items]))

    # -- instance.py:160-236 -----------------------------------------------------------------
    def major(self, num_classes=None):
        """``(obj, count, y)`` of the object each cluster overlaps most; a cluster whose
        largest overlap is void but holds <= 50 % void points gets its largest non-void
        overlap.  Ties -> first pair of the cluster."""
        num_classes = num_classes if num_classes else self.y.max() + 1
        cluster_idx = self.indices
        pair_is_void = (self.y < 0) | (self.y >= num_classes)
        x = torch.stack((self.count, self.count * ~pair_is_void)).T.contiguous()
        best, arg = scatter_max(x, cluster_idx, 0, None, self.num_groups)
        count, argmax = best[:, 0].contiguous(), arg[:, 0]
        obj, y = self.obj[argmax], self.y[argmax]
        is_major_void = (y < 0) | (y >= num_classes)
        if bool((~is_major_void).all()):
            return obj, count, y
        total = scatter_sum(self.count, cluster_idx, 0, None, self.num_groups)
        major_50_plus = (count / total) > 0.5
        if bool(major_50_plus[is_major_void].all()):
            return obj, count, y
        arg_nv = arg[:, 1]
        count[is_major_void] = best[:, 1][is_major_void]
        obj[is_major_void] = self.obj[arg_nv][is_major_void]
        y[is_major_void] = self.y[arg_nv][is_major_void]
        return obj, count, y

    def merge(self, idx):
        """Clusters merged into the parents ``idx`` (dense in [0, max]) - instance.py:238-266."""
        idx = tensor_idx(idx, self.device)
        if idx.shape != torch.Size([self.num_groups]):
            raise AssertionError(
                f"Expected indices of shape {torch.Size([self.num_groups])}, but received "
                f"shape {idx.shape} instead")
        if not (int(idx.min()) == 0 and idx.unique().numel() == int(idx.max()) + 1):
            raise AssertionError("Expected contiguous indices in [0, max]")
        return InstanceData(idx[self.indices], self.obj, self.count, self.y, dense=True)

    def iou_and_size(self):
        """Per pair: IoU of cluster and object, cluster size, object size
        (instance.py:268-298)."""
        a_idx = self.indices
        b_idx = _consecutive(self.obj)[0]
        a_size = scatter_sum(self.count, a_idx, 0, None, self.num_groups)[a_idx]
        b_size = scatter_sum(self.count, b_idx)[b_idx]
        if self.pair_cropped_count is not None:
            b_size = b_size + self.pair_cropped_count
        iou = self.count / (a_size + b_size - self.count)
        return iou, a_size, b_size

    def estimate_centroid(self, cluster_pos, mode="iou"):
        """``(obj_pos, obj_idx)``: weighted barycentre of the clusters overlapping each object,
        objects in increasing index order (instance.py:300-352)."""
        a_idx = self.indices
        b_idx, perm = _consecutive(self.obj)
        obj_idx = self.obj[perm]
        a_pos = cluster_pos[a_idx]
        mode = mode.lower()
        if mode == "iou":
            w = self.iou_and_size()[0]
        elif mode == "product-iou":
            _, a_size, b_size = self.iou_and_size()
            w = self.count ** 2 / (a_size * b_size)
        elif mode == "overlap":
            w = self.count
        else:
            raise NotImplementedError
        w = w.view(-1, 1)
        a_wpos = torch.cat((a_pos * w, w.to(a_pos.dtype)), dim=1)
        res = scatter_sum(a_wpos, b_idx, 0, None, obj_idx.numel())
        return res[:, :-1] / res[:, -1].view(-1, 1), obj_idx

    def instance_graph(self, edge_index, num_classes=None, smooth_affinity=True):
        """Trimmed (i < j, no loops, no duplicates) cluster graph and the affinity of each edge:
        ``(overlap(i, obj_j) / size_i + overlap(j, obj_i) / size_j) / 2`` with obj_* the major
        objects, or ``obj_i == obj_j`` (instance.py:354-460)."""
        obj_edge_index = to_trimmed(edge_index.to(self.device))
        if obj_edge_index.numel() == 0:
            return obj_edge_index, torch.zeros(0, device=self.device)
        sp_obj_idx = self.major(num_classes=num_classes)[0]
        i_obj_idx = sp_obj_idx[obj_edge_index[0]]
        j_obj_idx = sp_obj_idx[obj_edge_index[1]]
        if not smooth_affinity:
            return obj_edge_index, (i_obj_idx == j_obj_idx).float()
        # shared compact numbering of (cluster, object) pairs: those stored (A), and those an
        # edge asks about (B, C) - absent pairs have an empty overlap
        base = self.obj.max() + 1
        A = self.indices * base + self.obj
        B = obj_edge_index[0] * base + j_obj_idx
        C = obj_edge_index[1] * base + i_obj_idx
        uid = torch.unique(torch.cat((A, B, C)), sorted=True, return_inverse=True)[1]
        nA, nB = A.numel(), B.numel()
        overlaps = torch.zeros(int(uid.max()) + 1, device=self.device)
        overlaps[uid[:nA]] = self.count.float()
        overlap_i_obj_j = overlaps[uid[nA:nA + nB]]
        overlap_j_obj_i = overlaps[uid[nA + nB:]]
        sp_size = scatter_sum(self.count, self.indices, 0, None, self.num_groups)
        size_i = sp_size[obj_edge_index[0]].float()
        size_j = sp_size[obj_edge_index[1]].float()
        return obj_edge_index, (overlap_i_obj_j / size_i + overlap_j_obj_i / size_j) / 2

    def search_void(self, num_classes):
        """``(cluster_mask, pair_mask, pair_cropped_count)``: clusters with > 50 % void points,
        pairs that are void or belong to such a cluster, and per pair the part of its object
        removed with those clusters (instance.py:462-546)."""
        is_pair_b_void = (self.y < 0) | (self.y >= num_classes)
        pair_a_idx = self.indices
        a_size = scatter_sum(self.count, pair_a_idx, 0, None, self.num_groups)
        a_void = scatter_sum(self.count * is_pair_b_void, pair_a_idx, 0, None, self.num_groups)
        is_a_void = (a_void / a_size.float()) > 0.5
        b_idx = _consecutive(self.obj)[0]
        pair_cropped_count = scatter_sum(self.count * is_a_void[pair_a_idx], b_idx)[b_idx]
        return is_a_void, is_pair_b_void | is_a_void[pair_a_idx], pair_cropped_count

    def remove_void(self, num_classes):
        """``(instance_data, non_void_cluster_mask)`` without the void clusters, objects and
        pairs (instance.py:548-617)."""
        is_cluster_void, is_pair_void, pair_cropped_count = self.search_void(num_classes)
        keep = ~is_pair_void
        idx = _consecutive(self.indices[keep])[0]
        out = InstanceData(idx, self.obj[keep], self.count[keep], self.y[keep], dense=True)
        # (cluster, obj) pairs are unique and sorted by cluster: the dense constructor keeps
        # their order, so the per-pair crop counts stay aligned
        out.pair_cropped_count = pair_cropped_count[keep]
        return out, ~is_cluster_void

    def target_label_histogram(self, num_classes):
        """[num_clusters, num_classes + 1] histogram of point labels, void in the last column
        (instance.py:634-652)."""
        y = self.y.clone()
        y[(y < 0) | (y > num_classes)] = num_classes
        hist = torch.nn.functional.one_hot(y, num_classes=num_classes + 1) * self.count.view(-1, 1)
        return scatter_sum(hist, self.indices, 0, None, self.num_groups)

    def oracle(self, num_classes):
        """Best predictions the partition allows: clusters grouped by their major object,
        scored by the IoU with it -> ``(scores, y, InstanceData)`` (instance.py:690-735)."""
        obj, _, y = self.major(num_classes=num_classes)
        idx, perm = _consecutive(obj)
        oracle = self.merge(idx)
        iou = oracle.iou_and_size()[0]
        argmax = scatter_max(oracle.count, oracle.indices, 0, None, oracle.num_groups)[1]
        return iou[argmax], y[perm], oracle

    def __repr__(self):
        return (f"InstanceData(num_clusters={self.num_clusters}, num_overlaps={self.num_overlaps}, "
                f"device={self.device})")
