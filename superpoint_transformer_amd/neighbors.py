"""Radius-bounded kNN and point geometric features on the HIP kernels: the
preprocessing half of the hot path (src/utils/neighbors.py:24-123,186-242,
668-684; src/utils/geometry.py:80-174)."""
import ctypes

import torch

from . import _lib
from . import ops
from .ops import _workspace

__all__ = ["frnn_grid_points", "knn_1", "knn_1_features", "knn_1_graph", "oversample_partial_neighborhoods", "knn_2", "neighbors_dense_to_csr",
           "geometric_features", "GEOF_COLUMNS", "cluster_radius_nn_graph",
           "scatter_nearest_neighbor"]

GEOF_COLUMNS = ["linearity", "planarity", "scattering", "verticality", "normal_x",
                "normal_y", "normal_z", "length", "surface", "volume", "curvature"]


def _bbox(xyz):
    """[min (3), max (3)] of a contiguous f32 cloud, on the device."""
    buf = torch.empty(12, dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        st = _lib.lib.spt_bbox_f32(_lib.ptr(xyz), xyz.shape[0], _lib.ptr(buf),
                                   _lib.stream_ptr(xyz.device))
    _lib.check(st, "spt_bbox_f32")
    return buf[0:6]


def _grid_for(search, r, K, cell_size=None, self_search=False):
    """Host-side grid description.  The result of a search does not depend on the cell size,
    only the speed does: ~1.5 K points per non-empty cell measured fastest for the
    wave-per-query kernel, ~0.6 K for the self-search.  Read-backs: the bounding box, the size
    of the probe subsample, three cell counts (all from small kernels of this library: the
    probes were 2.5 ms of a 25 ms preprocessing call as torch expressions)."""
    lo_hi = _bbox(search).tolist()                      # the one read-back of the bounding box
    lo_h, hi_h = lo_hi[0:3], lo_hi[3:6]
    ext = [max(h - l, 1e-6) for l, h in zip(lo_h, hi_h)]
    n = search.shape[0]
    if cell_size is None:
        # occupancy of the non-empty cells at two probe sizes (s1, s1/2) on a subsample gives the
        # local dimension d of the cloud (2 = surfaces, 3 = volumetric scans, vegetation): the
        # points per cell scale as s^d
        import math
        import os
        dev = search.device
        o3 = (ctypes.c_float * 3)(*lo_h)
        # probe on a SPATIALLY COHERENT subsample (whole coarse cells of size r, ~2 M points):
        # the counts per fine cell are then exact, which a random subsample cannot give at the
        # scale of the target cell (a few dozen points).  One kernel compacts it
        # (spt_knn_subsample_f32; as torch expressions on the full cloud this was 1 ms at 15 M points)
        sub = search
        if n > 2_000_000:
            buf = torch.empty((n, 3), dtype=torch.float32, device=dev)
            cnt = torch.empty(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                st = _lib.lib.spt_knn_subsample_f32(
                    _lib.ptr(search), n, ctypes.cast(o3, ctypes.c_void_p), float(r),
                    int(65536 * 2_000_000 / n), _lib.ptr(buf), _lib.ptr(cnt), _lib.stream_ptr(dev))
            _lib.check(st, "spt_knn_subsample_f32")
            m = int(cnt.item())
            if m >= 1000:
                sub = buf[:m]
        m = sub.shape[0]
        cnt64 = torch.empty(1, dtype=torch.int64, device=dev)

        def per_cell(sz):
            """Points per non-empty cell of the subsample at cell size ``sz`` (exact count: one
            bit per cell, no sort; grids too large for a bitmap: cell ids + ``torch.unique``)."""
            d1 = [min(int(e / sz) + 1, (1 << 31) - 1) for e in ext]
            d3 = (ctypes.c_int32 * 3)(*d1)
            ncells = d1[0] * d1[1] * d1[2]
            with torch.cuda.device(dev):
                if ncells <= (1 << 28):
                    # a bitmap of <= 32 MB (the cached workspace keeps its largest size for the
                    # life of the process: finer grids take the ids + unique route on temporaries)
                    ws = _workspace(_lib.lib.spt_grid_count_cells_workspace_bytes(ncells), dev)
                    st = _lib.lib.spt_grid_count_cells_f32(
                        _lib.ptr(sub), m, float(sz), ctypes.cast(o3, ctypes.c_void_p),
                        ctypes.cast(d3, ctypes.c_void_p), _lib.ptr(cnt64), _lib.ptr(ws), ws.numel(),
                        _lib.stream_ptr(dev))
                    _lib.check(st, "spt_grid_count_cells_f32")
                    return m / max(int(cnt64.item()), 1)
                ids = torch.empty(m, dtype=torch.int64, device=dev)
                st = _lib.lib.spt_grid_cell_ids_f32(
                    _lib.ptr(sub), m, float(sz), ctypes.cast(o3, ctypes.c_void_p),
                    ctypes.cast(d3, ctypes.c_void_p), _lib.ptr(ids), _lib.stream_ptr(dev))
            _lib.check(st, "spt_grid_cell_ids_f32")
            return m / max(int(torch.unique(ids).numel()), 1)

        # first guess from a coarse probe assuming surfaces, then the local dimension and the
        # occupancy measured AT that guess (stacked surfaces look volumetric from afar)
        s0 = float(r) / 4
        s1 = min(max(s0 * (max(K, 8.0) / max(per_cell(s0), 1e-3)) ** 0.5, float(r) / 64), float(r))
        o1, o2 = per_cell(s1), per_cell(2 * s1)
        occ1 = o1
        dim = min(max(math.log2(max(o2 / max(o1, 1e-9), 1.0 + 1e-6)), 1.0), 3.0)
        # self-search (shared candidate streams): the wave scans 5 x 9 cells per 64 queries, so
        # fewer points per cell than the wave-per-query kernel likes (SPT_KNN_OCC: tuning knob)
        # (0.6: re-measured in round 5 at both preprocessing settings - 0.55 .. 0.65 within 2 %,
        # 0.75 4 % and 0.9 20 % slower, with or without the eigenfeatures in the kernel)
        occ_k = float(os.environ.get("SPT_KNN_OCC", "0.6" if self_search else "1.5"))
        target = max(occ_k * K, 8.0)
        s = s1 * (target / max(occ1, 1e-3)) ** (1.0 / dim)
        s = min(max(s, float(r) / 64), float(r))
    else:
        s = float(cell_size)
    while True:
        dims = [int(e / s) + 1 for e in ext]
        if dims[0] * dims[1] * dims[2] < (1 << 30):
            break
        s *= 1.26
    return s, lo_h, dims


def frnn_grid_points(query, search, K, r, squared=True, inclusive=False, cell_size=None):
    """Same contract as ``frnn.frnn_grid_points`` on one cloud (the reference
    always calls it with batch size 1, src/utils/neighbors.py:82-83):
    returns ``(dists [nq,K], idxs [nq,K])``, ascending, -1 padded.  ``K > 64``: the kernels hold
    64 neighbours per query; the rest comes from continuation searches ("the K' nearest strictly
    after the last one found", ``spt_grid_knn_after_f32``), 64 at a time."""
    _lib.require_cuda(query, search)
    q = query.detach().float().contiguous()
    s = q if search is query else search.detach().float().contiguous()
    nq, ns = q.shape[0], s.shape[0]
    K = int(K)
    dev = q.device
    idx = torch.empty((nq, K), dtype=torch.int64, device=dev)
    dist = torch.empty((nq, K), dtype=torch.float32, device=dev)
    if nq == 0:
        return dist, idx
    if ns == 0:
        return dist.fill_(-1), idx.fill_(-1)
    cs, origin, dims = _grid_for(s, r, min(K, 64), cell_size, self_search=search is query)
    # self-search of a large cloud: keep the grid's cell order of the points, the kernels
    # that gather neighbourhoods afterwards (geometric_features) visit points in that order
    order = torch.empty(ns, dtype=torch.int32, device=dev) \
        if (search is query and ns >= _ORDER_MIN_POINTS) else None
    ncells = dims[0] * dims[1] * dims[2]
    nb = _lib.lib.spt_grid_knn_workspace_bytes(ns, ncells)
    ws = _workspace(nb, dev)
    o3 = (ctypes.c_float * 3)(*origin)
    d3 = (ctypes.c_int32 * 3)(*dims)
    k0 = min(K, 64)
    chunk_i = idx if K <= 64 else torch.empty((nq, k0), dtype=torch.int64, device=dev)
    chunk_d = dist if K <= 64 else torch.empty((nq, k0), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_grid_knn_f32(
            _lib.ptr(q), nq, _lib.ptr(s), ns, k0, float(r), cs,
            ctypes.cast(o3, ctypes.c_void_p), ctypes.cast(d3, ctypes.c_void_p), 1,
            int(inclusive), 1 if K > 64 else int(squared), _lib.ptr(chunk_i), _lib.ptr(chunk_d),
            _lib.ptr(order), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(st, "spt_grid_knn_f32")
    if K > 64:
        idx[:, :k0], dist[:, :k0] = chunk_i, chunk_d
        done = k0
        while done < K:
            kk = min(64, K - done)
            after_i = idx[:, done - 1].contiguous()
            after_d = dist[:, done - 1].contiguous()            # squared (kept so until the end)
            ci = torch.empty((nq, kk), dtype=torch.int64, device=dev)
            cd = torch.empty((nq, kk), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                st = _lib.lib.spt_grid_knn_after_f32(
                    _lib.ptr(q), nq, _lib.ptr(s), ns, kk, float(r), cs,
                    ctypes.cast(o3, ctypes.c_void_p), ctypes.cast(d3, ctypes.c_void_p),
                    int(inclusive), 1, _lib.ptr(after_i), _lib.ptr(after_d), _lib.ptr(ci),
                    _lib.ptr(cd), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            _lib.check(st, "spt_grid_knn_after_f32")
            idx[:, done:done + kk], dist[:, done:done + kk] = ci, cd
            done += kk
        if not squared:
            dist = torch.where(idx >= 0, dist.clamp(min=0).sqrt(), dist)
    if order is not None:
        _remember_order(search, order)
    return dist, idx


def _batch_offset(xyz, batch, r_max):
    """The reference separates batch items by shifting z (neighbors.py:75-78)."""
    if batch is None:
        return xyz
    z = xyz[:, 2]
    off = torch.zeros_like(xyz)
    off[:, 2] = batch.to(xyz.dtype) * (z.max() - z.min() + r_max + 1)
    return xyz + off


def knn_1(xyz, k, r_max=1, batch=None, oversample=False, self_is_neighbor=False,
          squared=True):
    """k nearest OTHER points within r_max of every point (neighbors.py:51-123):
    searches k+1 and drops the first column (the point itself).  Returns
    (neighbors [N,k] int64 with -1 padding, distances [N,k])."""
    p = _batch_offset(xyz, batch, r_max)
    ks = k if self_is_neighbor else k + 1
    dist, idx = frnn_grid_points(p, p, ks, r_max, squared=squared)
    if not self_is_neighbor:
        idx, dist = idx[:, 1:], dist[:, 1:]
    if oversample:
        idx, dist = oversample_partial_neighborhoods(idx.contiguous(), dist.contiguous(), k)
    return idx, dist


def knn_1_features(xyz, k, r_max=1, k_min=1, raw=False, squared=True, formulation=-1,
                   cell_size=None):
    """``knn_1(xyz, k, r_max)`` and ``geometric_features(xyz, neighbors, k_min)`` in one device
    pass: the ``KNN`` -> ``PointFeatures`` pair of the reference's preprocessing chain
    (src/transforms/neighbors.py:11-95, src/transforms/point.py:160-180) without the
    ``[N, k]`` index rows and the neighbours' positions travelling back from HBM for the
    eigenfeatures - the kNN kernel sums each neighbourhood's moments while the winners' rows are
    at hand (``spt_grid_knn_geof_f32``).  Returns ``(neighbors [N,k], distances [N,k],
    features [N,11])``; neighbours / distances are exactly ``knn_1``'s, the features
    ``geometric_features``'s (same f64 moments about the point, another summation order:
    identical where the sums are exact, within an ulp of the f32 outputs otherwise).  One cloud
    (no ``batch``), the point itself excluded from its neighbours, no oversampling, k <= 63;
    other settings: call the two functions.  ``formulation`` / ``cell_size``: as in
    ``spt_grid_knn_ex_f32`` / ``frnn_grid_points`` (tests; the results do not depend on them)."""
    k = int(k)
    if k + 1 > 64:
        raise ValueError("knn_1_features: k + 1 <= 64 (call knn_1 and geometric_features for more)")
    _lib.require_cuda(xyz)
    p = xyz.detach().float().contiguous()
    n = p.shape[0]
    dev = p.device
    K = k + 1
    idx = torch.empty((n, K), dtype=torch.int64, device=dev)
    dist = torch.empty((n, K), dtype=torch.float32, device=dev)
    feats = torch.empty((n, 11), dtype=torch.float32, device=dev)
    if n == 0:
        return idx[:, 1:], dist[:, 1:], feats
    cs, origin, dims = _grid_for(p, r_max, K, cell_size, self_search=True)
    order = torch.empty(n, dtype=torch.int32, device=dev) if n >= _ORDER_MIN_POINTS else None
    ncells = dims[0] * dims[1] * dims[2]
    ws = _workspace(_lib.lib.spt_grid_knn_workspace_bytes(n, ncells), dev)
    o3 = (ctypes.c_float * 3)(*origin)
    d3 = (ctypes.c_int32 * 3)(*dims)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_grid_knn_geof_f32(
            _lib.ptr(p), n, K, float(r_max), cs, ctypes.cast(o3, ctypes.c_void_p),
            ctypes.cast(d3, ctypes.c_void_p), 0, int(squared), int(k_min), 0 if raw else 1,
            _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(feats), _lib.ptr(order), int(formulation),
            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(st, "spt_grid_knn_geof_f32")
    if order is not None:
        _remember_order(xyz, order)
    return idx[:, 1:], dist[:, 1:], feats


def oversample_partial_neighborhoods(neighbors, distances, k):
    """Fill the missing (-1) entries of every neighbourhood holding 1..k-1 neighbours with
    uniform draws among the neighbours it does hold, distances following (neighbors.py:420-488;
    neighbours sorted by increasing distance, so the missing ones are the last).  An empty
    neighbourhood stays empty.  In place, like the reference."""
    assert neighbors.dim() == distances.dim() == 2
    found = (neighbors != -1).sum(dim=1)
    row, col = torch.where(neighbors == -1)
    n_valid = found[row]
    # the reference's 0.9999 keeps rand() ~ 1 from indexing one past the valid entries
    pick = (n_valid * torch.rand(row.numel(), device=neighbors.device) * 0.9999).floor().long()
    neighbors[row, col] = neighbors[row, pick]
    distances[row, col] = distances[row, pick]
    return neighbors, distances


def knn_1_graph(xyz, k, r_max=1, batch=None, oversample=False, self_is_neighbor=False,
                trim=True):
    """``knn_1`` as a duplicate-free edge list with the (smallest) distance of each edge;
    ``trim``: undirected edges kept once, i < j, no loops (neighbors.py:126-183)."""
    neighbors, distances = knn_1(xyz, k, r_max=r_max, batch=batch, oversample=oversample,
                                 self_is_neighbor=self_is_neighbor)
    n = xyz.shape[0]
    source = torch.arange(n, device=xyz.device).repeat_interleave(k)
    target = neighbors.flatten()
    found = target != -1
    source, target, d = source[found], target[found], distances.flatten()[found]
    if trim:
        lo, hi = torch.minimum(source, target), torch.maximum(source, target)
        keep = lo != hi
        source, target, d = lo[keep], hi[keep], d[keep]
    key, inv = torch.unique(source * n + target, sorted=True, return_inverse=True)
    if key.numel() == 0:
        return torch.stack([key, key]), d
    dmin = ops.segment_reduce(d, inv, key.numel(), "min")
    return torch.stack([key // n, key % n]), dmin


def knn_2(x_search, x_query, k, r_max=1, batch_search=None, batch_query=None, squared=True):
    """k nearest search points of every query (neighbors.py:186-242)."""
    if (batch_search is None) != (batch_query is None):
        raise ValueError("pass both batch vectors or neither")
    if batch_search is not None:
        z = torch.cat((x_search[:, 2], x_query[:, 2]))
        zoff = z.max() - z.min() + r_max + 1
        x_search = x_search.clone()
        x_query = x_query.clone()
        x_search[:, 2] += batch_search.to(x_search.dtype) * zoff
        x_query[:, 2] += batch_query.to(x_query.dtype) * zoff
    dist, idx = frnn_grid_points(x_query, x_search, k, r_max, squared=squared)
    if k == 1:
        return idx[:, 0], dist[:, 0]
    return idx, dist


def neighbors_dense_to_csr(nn):
    """[N,k] with negative = missing -> (ptr [N+1], val [M], sizes [N])
    (neighbors.py:668-684): per-row counts, device scan, ordered emit (one kernel each).  The
    one host sync reads M to size ``val`` - the reference's ``nn[~mask]`` syncs the same way."""
    _lib.require_cuda(nn)
    if nn.dim() != 2:
        raise ValueError("nn must be [N, k]")
    t = nn.long().contiguous()
    n, k = t.shape
    dev = t.device
    ptr = torch.empty(n + 1, dtype=torch.long, device=dev)
    val = torch.empty(max(n * k, 1), dtype=torch.long, device=dev)
    sizes = torch.empty(n, dtype=torch.long, device=dev)
    nb = _lib.lib.spt_neighbors_dense_to_csr_workspace_bytes(n)
    ws = _workspace(nb, dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_neighbors_dense_to_csr(_lib.ptr(t), n, k, _lib.ptr(ptr), _lib.ptr(val),
                                                 _lib.ptr(sizes), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_ptr(dev))
    _lib.check(st, "spt_neighbors_dense_to_csr")
    return ptr, val[:int(ptr[-1])], sizes


_ORDER_MIN_POINTS = 200_000
_ORDER_ATTR = "_spt_cell_order"


def _remember_order(xyz, order):
    try:
        setattr(xyz, _ORDER_ATTR, (xyz._version, xyz.data_ptr(), order))
    except AttributeError:
        pass


def _recall_order(xyz):
    memo = getattr(xyz, _ORDER_ATTR, None)
    if memo is not None and memo[0] == xyz._version and memo[1] == xyz.data_ptr() \
            and memo[2].numel() == xyz.shape[0]:
        return memo[2]
    return None


def spatial_order(xyz, points_per_cell=32):
    """int32 permutation grouping the points by the cells of a uniform grid holding
    about ``points_per_cell`` points each (one host sync for the bounding box)."""
    _lib.require_cuda(xyz)
    p = xyz.detach().float().contiguous()
    n = p.shape[0]
    box = _bbox(p)
    ext = float((box[3:6] - box[0:3]).max())
    s, lo_h, dims = _grid_for(p, max(ext, 1e-3) / 64, points_per_cell / 1.5)
    ncells = dims[0] * dims[1] * dims[2]
    order = torch.empty(max(n, 1), dtype=torch.int32, device=p.device)
    nbytes = _lib.lib.spt_spatial_order_workspace_bytes(n, ncells)
    ws = _workspace(nbytes, p.device)
    o3 = (ctypes.c_float * 3)(*lo_h)
    d3 = (ctypes.c_int32 * 3)(*dims)
    with torch.cuda.device(p.device):
        st = _lib.lib.spt_spatial_order(
            _lib.ptr(p), n, float(s), ctypes.cast(o3, ctypes.c_void_p),
            ctypes.cast(d3, ctypes.c_void_p), _lib.ptr(order), _lib.ptr(ws), nbytes,
            _lib.stream_ptr(p.device))
    _lib.check(st, "spt_spatial_order")
    return order[:n]


def _geometric_features_optimal(xyz, nn, k_min, k_step, k_min_search, add_self, raw, order):
    """geometry.py:248-287: the features of the neighbourhood size - among k0 = max(k_min,
    k_min_search), the multiples of ``k_step`` and k_max - whose eigenvalues have the lowest
    eigenentropy, per point (first such size on ties).  One launch of the feature kernel per
    candidate size on the first k columns; the eigenvalues are read back from its length /
    scattering / planarity columns (l1 = length, l3 = scattering (l1 + 1e-3),
    l2 = l3 + planarity (l1 + 1e-3))."""
    k_max = nn.shape[1] + int(add_self)                     # columns including the point itself
    k0 = max(k_min, k_min_search)
    sizes = [k for k in range(k0, k_max + 1)
             if not ((k > k0) and (k % k_step != 0) and (k != k_max))]
    if not sizes:
        raise ValueError(f"no neighbourhood size to search: max(k_min, k_min_search) = {k0} "
                         f"exceeds the {k_max} neighbours given")
    best = entropy = None
    for k in sizes:
        f = geometric_features(xyz, nn[:, :k - int(add_self)], k_min=k_min,
                               add_self_as_neighbor=add_self, raw=True, order=order)
        l1 = f[:, 7]
        l3 = f[:, 2] * (l1 + 1e-3)
        l2 = l3 + f[:, 1] * (l1 + 1e-3)
        ev = torch.stack((l3, l2, l1), dim=1) ** 2
        e = ev / (ev.sum(dim=1, keepdim=True) + 1e-3)
        ent = (-e * torch.log(e + 1e-3)).sum(dim=1)
        if best is None:
            best, entropy = f, ent
            continue
        better = ent < entropy
        best = torch.where(better.view(-1, 1), f, best)
        entropy = torch.where(better, ent, entropy)
    if not raw:                                             # geometry.py:121, 124
        best[:, 3] *= 2
        flip = best[:, 6] < 0
        best[flip, 4:7] *= -1
    return best


def geometric_features(xyz, nn, k_min=1, add_self_as_neighbor=True, raw=False, order=None,
                       k_step=-1, k_min_search=25):
    """[N,11] features in pgeof's column order (``GEOF_COLUMNS``) from dense
    neighbours ``nn`` [N,k] (-1 = missing).  ``raw=False`` includes the tail of
    ``geometric_features`` (verticality * 2, normals flipped to z >= 0,
    geometry.py:121,124).  ``order``: visiting order of the points.  Default: the grid cell
    order a preceding ``knn_1`` left on this tensor (neighbourhoods gathered by one wave then
    overlap: 11.8 vs 17.4 ms at 15 M shuffled points), else as stored; ``False`` forces "as
    stored"; a permutation (e.g. ``spatial_order(xyz)``) is used as given.  The result does
    not depend on it.  ``k_step >= 0``: per point, the neighbourhood size of lowest
    eigenentropy (``nn`` sorted by increasing distance; see ``_geometric_features_optimal``)."""
    if k_step is not None and k_step >= 0:
        return _geometric_features_optimal(xyz, nn, int(k_min), int(k_step), int(k_min_search),
                                           bool(add_self_as_neighbor), raw, order)
    _lib.require_cuda(xyz, nn)
    p = xyz.detach().float().contiguous()
    if nn.dtype != torch.int64:
        nn = nn.long()
    # a column slice of a wider row-major table (knn_1's result) is read in place
    if nn.dim() != 2 or (nn.shape[1] > 0 and nn.stride(1) != 1) or nn.stride(0) < nn.shape[1] \
            or nn.shape[0] <= 1:
        nn = nn.contiguous()
    n, k = nn.shape
    ld = nn.stride(0) if n > 1 and k > 0 else k
    if order is None and n == xyz.shape[0]:
        order = _recall_order(xyz)          # left by a preceding knn_1 on this very tensor
    if order is False:
        order = None                        # visit the points as stored
    if order is not None:
        order = order.to(torch.int32).contiguous()
    feats = torch.empty((n, 11), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        st = _lib.lib.spt_point_geof_dense_ld_f32(
            _lib.ptr(p), n, _lib.ptr(nn), k, int(ld), int(add_self_as_neighbor), int(k_min),
            0 if raw else 1, _lib.ptr(order), _lib.ptr(feats), _lib.stream_ptr(p.device))
    _lib.check(st, "spt_point_geof_dense_ld_f32")
    return feats


def geometric_features_csr(xyz, nn_val, nn_ptr, k_min=1, add_self=False, raw=True):
    """pgeof.compute_features' layout: CSR neighbourhoods (the caller already
    put the point itself in its list, geometry.py:95-96,142-153).  One feature
    row per neighbourhood: ``nn_ptr`` may describe fewer groups than ``xyz`` has
    rows (the sampled points of each segment, transforms/graph.py:234-242)."""
    _lib.require_cuda(xyz, nn_val, nn_ptr)
    p = xyz.detach().float().contiguous()
    n = nn_ptr.numel() - 1
    feats = torch.empty((n, 11), dtype=torch.float32, device=p.device)
    val64 = nn_val.long().contiguous()       # named: a converted copy must outlive the launch
    ptr64 = nn_ptr.long().contiguous()
    with torch.cuda.device(p.device):
        st = _lib.lib.spt_point_geof_csr_f32(
            _lib.ptr(p), n, _lib.ptr(val64), _lib.ptr(ptr64), int(add_self), int(k_min),
            0 if raw else 1, _lib.ptr(feats), _lib.stream_ptr(p.device))
    _lib.check(st, "spt_point_geof_csr_f32")
    return feats


def scatter_nearest_neighbor(points, index, edge_index, cycles=3, chunk_size=None,
                             num_clusters=None):
    """For each pair of clusters of ``edge_index`` the (approximately) two closest
    points between them (src/utils/scatter.py:128-238).  Returns
    ``(candidate [2E,3], candidate_idx [2,E])`` like the reference; ``chunk_size`` is
    accepted and ignored (nothing edge-wise is materialised here)."""
    anchors, _ = _pair_anchors(points, index, edge_index, cycles, num_clusters)
    p = points.detach().float()
    return torch.vstack((p[anchors[0]], p[anchors[1]])), anchors


def _pair_anchors(points, index, edge_index, cycles, num_clusters=None, edge_stride=None,
                  num_edges=None):
    from .csr import csr_of
    from .ops import segment_reduce
    _lib.require_cuda(points, index, edge_index)
    p = points.detach().float().contiguous()
    csr = csr_of(index, num_clusters)
    centroid = segment_reduce(p, index, csr.num_seg, "mean").contiguous()
    if edge_stride is None:
        edge_index = edge_index.long().contiguous()
        edge_stride = num_edges = edge_index.shape[1]
    dev = p.device
    anchors = torch.empty((2, num_edges), dtype=torch.int64, device=dev)
    d_nn = torch.empty(num_edges, dtype=torch.float32, device=dev)
    mean_size = csr.n / max(csr.num_seg, 1)
    lanes = 8 if mean_size <= 12 else 16 if mean_size <= 48 else 64
    with torch.cuda.device(dev):
        st = _lib.lib.spt_cluster_pair_anchors_f32(
            _lib.ptr(p), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), _lib.ptr(centroid),
            _lib.ptr(edge_index), num_edges, edge_stride, int(cycles), lanes, _lib.ptr(anchors),
            _lib.ptr(d_nn), _lib.stream_ptr(dev))
    _lib.check(st, "spt_cluster_pair_anchors_f32")
    return anchors, d_nn


def cluster_radius_nn_graph(x_points, idx, k_max=100, gap=0, batch=None, trim=True, cycles=3,
                            chunk_size=None, verbose=False, squared=True, num_clusters=None,
                            return_intermediate=False):
    """Radius neighbours of clusters: two clusters are neighbours if two of their
    points are ``gap`` or less apart (src/utils/neighbors.py:491-665; same steps:
    bounding-box centres -> knn_1 within max diameter + gap -> radius-sum filter ->
    to_trimmed / coalesce -> anchor points -> ``d_nn <= gap``).  Returns
    ``(edge_index [2,E], distances [E])``.

    ``squared``: whether the centre distances compared with the radius sum are the
    squared ones FRNN returns (neighbors.py:591-593 compares them as they come)."""
    from .csr import csr_of
    from .ops import segment_reduce
    _lib.require_cuda(x_points, idx)
    if k_max + 1 > 256:
        raise NotImplementedError("k_max <= 255 (four chained 64-neighbour searches; "
                                  "the reference's configs use 30)")
    x = x_points.detach().float().contiguous()
    dev = x.device
    csr = csr_of(idx, num_clusters)
    S = csr.num_seg
    lo = segment_reduce(x, idx, S, "min")
    hi = segment_reduce(x, idx, S, "max")
    diam = (hi - lo).max(dim=1).values
    center = (hi + lo) / 2
    r_search = float(diam.max() + gap)                         # host sync, as in the reference
    nb, dist = knn_1(center, k_max, r_max=r_search, batch=batch, squared=squared)
    r_seg = (diam / 2).contiguous()
    m = S * k_max
    edges = torch.empty((2, max(m, 1)), dtype=torch.int64, device=dev)
    edge_dist = torch.empty(max(m, 1), dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    nbytes = _lib.lib.spt_cluster_graph_edges_workspace_bytes(S, k_max)
    ws = _workspace(nbytes, dev)
    nb = nb.contiguous()
    dist = dist.contiguous()
    with torch.cuda.device(dev):
        st = _lib.lib.spt_cluster_graph_edges(
            _lib.ptr(nb), _lib.ptr(dist), _lib.ptr(r_seg), S, int(k_max), float(gap), int(trim),
            _lib.ptr(edges), _lib.ptr(edge_dist), _lib.ptr(count), _lib.ptr(ws), nbytes,
            _lib.stream_ptr(dev))
    _lib.check(st, "spt_cluster_graph_edges")
    E = int(count)
    anchors, d_nn = _pair_anchors(x, idx, edges, cycles, S, edge_stride=edges.shape[1],
                                  num_edges=E)
    keep = d_nn <= gap
    edge_index = edges[:, :E][:, keep]
    if return_intermediate:
        return edge_index, d_nn[keep], dict(trimmed=edges[:, :E], anchors=anchors, d_nn=d_nn,
                                            center_dist=edge_dist[:E])
    return edge_index, d_nn[keep]
