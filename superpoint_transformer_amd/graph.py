"""Superedge construction helpers of the preprocessing graph stage
(src/utils/graph.py, src/utils/edge.py, src/utils/sparse.py) on the HIP ops.

``subedges`` (graph.py:99-463) finds, for every edge between two segments, the level-0 point
pairs that "make up" the edge - the input of the superedge features.  The reference strings
~25 torch / torch_scatter calls over edge-wise expanded point lists; the same steps run here
with every segment-wise reduction on the CSR kernels (segment sum / min / max / mean,
``scatter_pca``) and the anchor search on ``spt_cluster_pair_anchors_f32``; the expansions,
masks and the per-group sorts are index plumbing (torch sorts on composite keys).
"""
import torch

from . import _lib
from .csr import csr_of
from .ops import segment_reduce

__all__ = ["to_trimmed", "edge_wise_points", "base_vectors_3d", "subedges"]


def to_trimmed(edge_index):
    """Undirected, duplicate-free, loop-free edges with i < j, sorted by (i, j)
    (graph.py:466-502: flip, coalesce, remove_self_loops)."""
    lo = torch.minimum(edge_index[0], edge_index[1])
    hi = torch.maximum(edge_index[0], edge_index[1])
    keep = lo != hi
    lo, hi = lo[keep], hi[keep]
    n = int(hi.max()) + 1 if hi.numel() else 1
    key = torch.unique(lo * n + hi, sorted=True)
    return torch.stack([key // n, key % n])


def base_vectors_3d(x):
    """Orthonormal bases whose first vector is ``x`` normalised (geometry.py:42-78); the two
    degenerate cases get the reference's arbitrary fill-ins."""
    a = x.clone()
    zero = a.norm(dim=1) == 0
    a[zero] = torch.tensor([1.0, 0.0, 0.0], dtype=x.dtype, device=x.device)
    a = a / a.norm(dim=1).view(-1, 1)
    b = torch.stack((a[:, 1] - a[:, 2], a[:, 2] - a[:, 0], a[:, 0] - a[:, 1]), dim=1)
    zb = b.norm(dim=1) == 0
    b[zb] = torch.tensor([2.0, 1.0, -1.0], dtype=x.dtype, device=x.device)
    b = b / b.norm(dim=1).view(-1, 1)
    c = torch.linalg.cross(a, b)
    return torch.stack((a, b, c), dim=1)


def _arange_interleave(width, start):
    """cat([arange(s, s + w) for s, w in zip(start, width)]) (tensor.py:122-140)."""
    total = int(width.sum())
    if total == 0:
        return torch.empty(0, dtype=torch.long, device=width.device)
    grp = torch.repeat_interleave(torch.arange(width.numel(), device=width.device), width)
    first = torch.cumsum(width, 0) - width
    return start[grp] + torch.arange(total, device=width.device) - first[grp]


def edge_wise_points(points, index, edge_index, num_segments=None):
    """For every edge, all points of its source segment and all points of its target segment
    (edge.py:22-77): ``((S_points, S_idx, S_uid), (T_points, T_idx, T_uid))``; edges are
    identified by their rank in (source, target) order."""
    csr = csr_of(index, num_segments)
    pointers, order = csr.rowptr.long(), csr.perm.long()
    size = pointers[1:] - pointers[:-1]
    n = max(int(edge_index.max()) + 1 if edge_index.numel() else 1, 1)
    uid = torch.unique(edge_index[0] * n + edge_index[1], sorted=True, return_inverse=True)[1]

    def expand(x_idx):
        sz = size[x_idx]
        pid = order[_arange_interleave(sz, pointers[:-1][x_idx])]
        return points[pid], pid, torch.repeat_interleave(uid, sz)

    return expand(edge_index[0]), expand(edge_index[1])


def _group_sort(value, group, descending=False):
    """Order by group first, ``value`` second (sparse.py:90-104: ``sparse_sort``): two stable
    sorts on exact keys instead of one sort on a normalised float key."""
    p1 = torch.argsort(value, descending=descending, stable=True)
    p2 = torch.argsort(group[p1], stable=True)
    return p1[p2]


def _preserving(mask, uid, num):
    """``idx_preserving_mask`` (scatter.py:241-246): never empty a group entirely."""
    kept = segment_reduce(mask.float().view(-1, 1), uid, num, "sum").view(-1)
    return mask | (kept == 0)[uid]


def subedges(points, index, edge_index, ratio=0.2, k_min=20, cycles=3, pca_on_cpu=False,
             margin=0.2, halfspace_filter=True, bbox_filter=True, target_pc_flip=True,
             source_pc_sort=False, chunk_size=None, verbose=False):
    """graph.py:99-463.  Returns ``(edge_index [2,E] trimmed, ST_pairs [2,M], ST_uid [M])``:
    pair m joins point ``ST_pairs[0,m]`` of the source segment with ``ST_pairs[1,m]`` of the
    target segment of edge ``ST_uid[m]``.  ``chunk_size`` / ``pca_on_cpu`` are accepted and
    ignored (nothing here needs chunking, the PCA is a kernel)."""
    from .neighbors import _pair_anchors
    from .segment import scatter_pca
    _lib.require_cuda(points, index, edge_index)
    points = points.detach().float().contiguous()
    index = index.long().contiguous()
    edge_index = to_trimmed(edge_index.long())
    E = edge_index.shape[1]
    dev = points.device
    if E == 0:
        z = torch.empty(0, dtype=torch.long, device=dev)
        return edge_index, torch.stack([z, z]), z
    num_segments = int(index.max()) + 1

    # closest pair of points of the two segments: origin and first axis of the edge's frame
    anchors, _ = _pair_anchors(points, index, edge_index, cycles, num_segments)
    s_anchor, t_anchor = points[anchors[0]], points[anchors[1]]
    base = base_vectors_3d(t_anchor - s_anchor)                      # [E,3,3], rows = axes

    (S_pts, S_idx, S_uid), (T_pts, T_idx, T_uid) = edge_wise_points(points, index, edge_index,
                                                                   num_segments)

    def to_anchor_base(X, uid, anchor):
        return torch.einsum("nd,nkd->nk", X - anchor[uid], base[uid])

    S_pts = to_anchor_base(S_pts, S_uid, s_anchor)
    T_pts = to_anchor_base(T_pts, T_uid, t_anchor)

    def take(mask, P, I, U):
        keep = torch.where(_preserving(mask, U, E))[0]
        return P[keep], I[keep], U[keep]

    if halfspace_filter:                                             # graph.py:297-312
        S_pts, S_idx, S_uid = take(S_pts[:, 0] <= margin, S_pts, S_idx, S_uid)
        T_pts, T_idx, T_uid = take(T_pts[:, 0] >= -margin, T_pts, T_idx, T_uid)

    if bbox_filter:                                                  # graph.py:325-350
        s_min = segment_reduce(S_pts[:, 1:].contiguous(), S_uid, E, "min")
        s_max = segment_reduce(S_pts[:, 1:].contiguous(), S_uid, E, "max")
        t_min = segment_reduce(T_pts[:, 1:].contiguous(), T_uid, E, "min")
        t_max = segment_reduce(T_pts[:, 1:].contiguous(), T_uid, E, "max")
        st_min = torch.max(s_min, t_min).clamp(max=-margin)
        st_max = torch.min(s_max, t_max).clamp(min=margin)

        def in_bbox(P, U):
            return (P[:, 1:] >= st_min[U]).all(dim=1) & (P[:, 1:] <= st_max[U]).all(dim=1)
        S_pts, S_idx, S_uid = take(in_bbox(S_pts, S_uid), S_pts, S_idx, S_uid)
        T_pts, T_idx, T_uid = take(in_bbox(T_pts, T_uid), T_pts, T_idx, T_uid)

    # closest to the anchor first, along the edge direction (graph.py:359-370)
    p = _group_sort(S_pts[:, 0], S_uid, descending=True)
    S_pts, S_idx, S_uid = S_pts[p], S_idx[p], S_uid[p]
    p = _group_sort(T_pts[:, 0], T_uid, descending=False)
    T_pts, T_idx, T_uid = T_pts[p], T_idx[p], T_uid[p]

    # top `ratio` of the points, at least k_min, the same number on both sides (graph.py:379-399)
    s_size = torch.bincount(S_uid, minlength=E)
    t_size = torch.bincount(T_uid, minlength=E)
    s_k = (s_size * ratio).long().clamp(min=k_min).min(s_size)
    t_k = (t_size * ratio).long().clamp(min=k_min).min(t_size)
    st_k = torch.min(s_k, t_k)
    sel = _arange_interleave(st_k, torch.cumsum(s_size, 0) - s_size)
    S_pts, S_idx, S_uid = S_pts[sel], S_idx[sel], S_uid[sel]
    sel = _arange_interleave(st_k, torch.cumsum(t_size, 0) - t_size)
    T_pts, T_idx, T_uid = T_pts[sel], T_idx[sel], T_uid[sel]

    # first principal component of each side's selected points (graph.py:415-428)
    s_v = scatter_pca(S_pts.contiguous(), S_uid, E)[1][:, :, -1].contiguous()
    t_v = scatter_pca(T_pts.contiguous(), T_uid, E)[1][:, :, -1].contiguous()

    if target_pc_flip and not source_pc_sort:                        # graph.py:437-444
        T_proj = (T_pts * t_v[T_uid]).sum(dim=1)
        s_mean = segment_reduce(S_pts.contiguous(), S_uid, E, "mean")
        _, amin = segment_reduce(T_proj.view(-1, 1).contiguous(), T_uid, E, "min", return_arg=True)
        t_minp = T_pts[amin.view(-1).long().clamp(max=max(T_pts.shape[0] - 1, 0))]
        st_u = t_minp - s_mean
        st_u = st_u / st_u.norm(dim=1).view(-1, 1)
        flip = (s_v * t_v).sum(dim=1) <= (s_v * st_u).sum(dim=1)
        t_v = torch.where(flip.view(-1, 1), -t_v, t_v)
    elif source_pc_sort:
        t_v = s_v

    def sort_along(P, I, U, v):                                      # sparse.py:107-137
        centroid = segment_reduce(P.contiguous(), U, E, "mean")
        proj = ((P - centroid[U]) * v[U]).sum(dim=1)
        q = _group_sort(proj, U, descending=False)
        return I[q], U[q]

    S_idx, S_uid = sort_along(S_pts, S_idx, S_uid, s_v)
    T_idx, T_uid = sort_along(T_pts, T_idx, T_uid, t_v)
    return edge_index, torch.vstack((S_idx, T_idx)), S_uid
