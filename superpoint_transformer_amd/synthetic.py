"""Seeded synthetic NAGs of S3DIS / DALES shape (bench + large-scale tests).

Recipe of SURVEY.md section 8(d), with the ratios measured on the reference's
demo room (notebooks/demo_nag_v3.h5): |P0|/|P1| = 34.87, |P1|/|P2| = 2.38;
level-0 -> level-1 segment sizes ~ clip(round(LogNormal(3.04, 1.0)), 1, 300);
upper levels 1 + Geometric; mean directed+self-loop degree 16.4 at level 1 and
29.9 at level 2.  Level-0 rows are SHUFFLED so that ``super_index``
is unsorted like the reference's (12 705 runs for 1 192 segments in the demo
room), and edge targets are non-local (median |s-t| ~ 0.23 N in the demo).
Default superpoint graph: uniformly random endpoints (``graph="random"``: a stress
case); ``graph="local"`` builds it by kNN on the segment centroids.

Everything is generated with torch on the requested device from one seed, so
every rank / run sees the same scene for the same (seed, sizes).
"""
import math

import torch

SCENES = {
    # name: (N0, N1, N2, E1, E2, num_clouds)
    "R": (41_568, 1_192, 501, 19_508, 14_965, 1),          # demo room shape
    "T": (1_200_000, 35_000, 14_500, 575_000, 435_000, 4),  # S3DIS train batch
    "S": (15_000_000, 428_571, 178_571, 7_030_000, 5_340_000, 1),  # full scene
    "D": (12_000_000, 342_857, 142_857, 5_620_000, 4_270_000, 1),  # DALES tile
}


def _segment_sizes(gen, n_child, n_parent, kind, device):
    """Positive integer sizes summing to n_child over n_parent segments."""
    if kind == "lognormal":
        z = torch.randn(n_parent, generator=gen, device=device)
        s = torch.exp(3.04 + 1.0 * z).round().clamp_(1, 300)
    else:
        mean = n_child / n_parent
        u = torch.rand(n_parent, generator=gen, device=device).clamp_(1e-9, 1 - 1e-9)
        p = 1.0 / max(mean, 1.0 + 1e-6)
        s = 1 + torch.floor(torch.log(u) / math.log(1 - p + 1e-12))
    s = s.long()
    # rescale proportionally (floor) so that the total never exceeds n_child,
    # every segment keeps >= 1 row, then hand out the remainder at random
    extra = s - 1
    tot = int(extra.sum())
    want = n_child - n_parent
    assert want >= 0, "need at least one child per parent"
    if tot > 0:
        extra = (extra.double() * (want / tot)).floor().long()
    s = 1 + extra
    diff = n_child - int(s.sum())
    if diff > 0:
        add = torch.randint(0, n_parent, (diff,), generator=gen, device=device)
        s = s + torch.bincount(add, minlength=n_parent)
    assert int(s.sum()) == n_child and int(s.min()) >= 1
    return s


def _super_index(gen, n_child, n_parent, kind, device, shuffle):
    sizes = _segment_sizes(gen, n_child, n_parent, kind, device)
    idx = torch.repeat_interleave(torch.arange(n_parent, device=device), sizes)
    if shuffle:
        idx = idx[torch.randperm(n_child, generator=gen, device=device)]
    return idx


def _edges(gen, n, e_target, device, cloud=None):
    """[2,E] directed edges incl. both directions and self loops, in the
    reference's final layout [i<j | j>i | loops] (transforms/graph.py:1268,
    :1442-1446): row 0 (source = softmax group) is unsorted.  Endpoints are drawn
    UNIFORMLY (Poisson-like degrees around E / n, no spatial locality): the stress case.
    ``cloud`` [n] (sorted cloud id of every node, batches of several clouds): both endpoints of
    an edge lie in ONE cloud and the stored (i < j) list is the concatenation of the clouds'
    lists, as NAGBatch.from_nag_list builds it (src/data/nag.py:878-898) - each third of the
    final layout is then sorted by cloud."""
    m = max((e_target - n) // 2, 0)
    a = torch.randint(0, n, (m,), generator=gen, device=device)
    if cloud is None:
        b = torch.randint(0, n, (m,), generator=gen, device=device)
    else:
        nb = int(cloud.max()) + 1
        size = torch.bincount(cloud, minlength=nb)
        start = torch.cumsum(size, 0) - size
        ca = cloud[a]
        u = torch.rand(m, generator=gen, device=device)
        b = start[ca] + (u * size[ca]).long().clamp_(max=size[ca] - 1)
        order = torch.argsort(ca, stable=True)
        a, b = a[order], b[order]
    keep = a != b
    a, b = a[keep], b[keep]
    lo, hi = torch.minimum(a, b), torch.maximum(a, b)
    loops = torch.arange(n, device=device)
    s = torch.cat([lo, hi, loops])
    t = torch.cat([hi, lo, loops])
    ei = torch.stack([s, t])
    ei._spt_mirror_pairs = int(lo.numel())      # what OnTheFlyHorizontalEdgeFeatures knows: csr.EdgeCSR
    return ei


def _knn_indices(pos, k):
    """[n, k] indices of the k nearest other points (-1 padded): the HIP grid kNN on the GPU,
    scipy's k-d tree for the CPU-baseline samples."""
    if pos.is_cuda:
        from .neighbors import knn_1
        r = 4.0 * (pos.max(0).values - pos.min(0).values).max().item() * (k / max(pos.shape[0], 1)) ** (1 / 3)
        nb, _ = knn_1(pos.contiguous(), k, max(r, 1e-3))
        return nb
    from scipy.spatial import cKDTree
    import numpy as np
    kk = min(k + 1, pos.shape[0])
    _, idx = cKDTree(pos.numpy()).query(pos.numpy(), k=kk)
    idx = np.asarray(idx).reshape(pos.shape[0], kk)[:, 1:]
    out = torch.full((pos.shape[0], k), -1, dtype=torch.long)
    out[:, :idx.shape[1]] = torch.from_numpy(idx.astype(np.int64))
    return out


def _edges_local(gen, pos, e_target, device, cloud=None):
    """The superpoint graph of SURVEY 8(d): undirected edges between spatially nearest segment
    centroids (every node proposes its k nearest, mutual proposals merge, a random subset hits
    the edge budget), then the reference's final layout [i<j | j>i | loops] like ``_edges``.
    ``cloud``: neighbours are searched inside a node's own cloud only (the clouds of a batch are
    moved 10 km apart for the search) and the (i < j) list is ordered by cloud."""
    n = pos.shape[0]
    m = max((e_target - n) // 2, 0)
    k = max(1, min(45, int(math.ceil(1.6 * m / max(n, 1)))))
    spos = pos if cloud is None else pos + torch.stack(
        [cloud.to(pos.dtype) * 1.0e4, torch.zeros_like(pos[:, 0]), torch.zeros_like(pos[:, 0])], 1)
    nb = _knn_indices(spos, k).to(device)
    a = torch.arange(n, device=device).repeat_interleave(k)
    b = nb.reshape(-1)
    keep = (b >= 0) & (a != b)
    if cloud is not None:
        keep &= cloud[a] == cloud[b.clamp(min=0)]
    a, b = a[keep], b[keep]
    key = torch.unique(torch.minimum(a, b) * n + torch.maximum(a, b))
    if key.numel() > m:
        key = key[torch.randperm(key.numel(), generator=gen, device=device)[:m]].sort().values
    lo, hi = key // n, key % n          # ascending lo: by cloud (the nodes of a cloud are contiguous)
    loops = torch.arange(n, device=device)
    ei = torch.stack([torch.cat([lo, hi, loops]), torch.cat([hi, lo, loops])])
    ei._spt_mirror_pairs = int(lo.numel())
    return ei


def _morton_code(pos):
    """30-bit Morton code of ``pos`` (10 bits per axis)."""
    lo, hi = pos.min(0).values, pos.max(0).values
    q = ((pos - lo) / (hi - lo).clamp_min(1e-9) * 1023.0).long().clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def _morton_order(pos):
    """argsort of the 30-bit Morton code of ``pos``."""
    return torch.argsort(_morton_code(pos), stable=True)


class SyntheticNAG:
    """Plain container: per-level dicts of tensors named like the reference's
    Data attributes (pos, x, super_index, edge_index, edge_attr, node_size,
    batch)."""

    def __init__(self, levels, num_clouds):
        self.levels = levels
        self.num_clouds = num_clouds

    def __getitem__(self, i):
        return self.levels[i]

    @property
    def num_points(self):
        return [lv["pos"].shape[0] for lv in self.levels]


def make_nag(scene="S", seed=1234, device="cpu", sizes=None, point_dim=8,
             edge_dim=18, scale=1.0, segment_dim=0, graph="random", order="storage"):
    """Build a 3-level synthetic NAG. ``scale`` < 1 shrinks every size
    proportionally (CPU-baseline samples).  ``segment_dim=0``: levels >= 1 carry
    no handcrafted node features, like the S3DIS config (``segment_hf: []``).

    ``graph``: "random" (default) - edge endpoints drawn uniformly: no locality at all, the
    worst case for everything that gathers rows by edge target (a stress case; degrees come out
    Poisson-like around the mean, not the clipped normal of the demo room); "local" - the
    kNN-on-centroids generator of SURVEY 8(d) (``_edges_local``).  ``order``: "storage"
    (default) keeps the nodes of levels 1 and 2 in their shuffled order - spatial neighbours
    are then far apart in memory, like the demo room (median |s - t| ~ 0.23 N); "morton" stores
    them along a Morton curve (what a spatially sorted dataset would hand over); "grouped" = the
    layout ``transforms.MortonOrder`` produces at load time (level 2 along the curve, level 1 by
    parent then curve, level-0 points grouped by superpoint)."""
    n0, n1, n2, e1, e2, b = sizes if sizes is not None else SCENES[scene]
    if scale != 1.0:
        n0, n1, n2 = (max(int(v * scale), 8) for v in (n0, n1, n2))
        e1, e2 = max(int(e1 * scale), n1), max(int(e2 * scale), n2)
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    si0 = _super_index(gen, n0, n1, "lognormal", device, shuffle=True)
    si1 = _super_index(gen, n1, n2, "geometric", device, shuffle=True)
    # cloud (batch) ids: contiguous blocks of level-2 nodes, pushed down
    b2 = (torch.arange(n2, device=device) * b // n2).long()
    if b > 1:
        # a batch is a concatenation of clouds (NAGBatch.from_nag_list, nag.py:878-898):
        # the nodes of one cloud are contiguous at EVERY level, shuffled inside the block
        o1 = torch.argsort(b2[si1], stable=True)
        inv1 = torch.empty_like(o1)
        inv1[o1] = torch.arange(n1, device=device)
        si1, si0 = si1[o1], inv1[si0]
        si0 = si0[torch.argsort(b2[si1][si0], stable=True)]
    b1 = b2[si1]
    b0 = b1[si0]

    def rnd(*shape, s=1.0):
        return torch.randn(*shape, generator=gen, device=device) * s

    # positions: parents are random centres, children jitter around them
    pos2 = torch.rand(n2, 3, generator=gen, device=device) * 40.0
    pos1 = pos2[si1] + rnd(n1, 3, s=1.5)
    pos0 = pos1[si0] + rnd(n0, 3, s=0.3)
    if order == "morton":
        # relabel levels 2 and 1 along a Morton curve (clouds stay contiguous: the cloud id is the
        # primary key); children keep pointing at the same parents under the new numbering
        def relabel(pos_l, b_l):
            o = _morton_order(pos_l)
            o = o[torch.argsort(b_l[o], stable=True)]
            inv = torch.empty_like(o)
            inv[o] = torch.arange(o.numel(), device=device)
            return o, inv
        o2, inv2 = relabel(pos2, b2)
        pos2, b2, si1 = pos2[o2], b2[o2], inv2[si1]
        o1, inv1_ = relabel(pos1, b1)
        pos1, b1, si1, si0 = pos1[o1], b1[o1], si1[o1], inv1_[si0]
    elif order == "grouped":
        # the layout transforms.MortonOrder produces at load time: level 2 along a Morton curve
        # (cloud first), level 1 sorted by parent then by its own Morton code, the points of level 0
        # grouped by their superpoint (the pool's CSR view becomes the identity)
        def by_key(key):
            o = torch.argsort(key, stable=True)
            inv = torch.empty_like(o)
            inv[o] = torch.arange(o.numel(), device=device)
            return o, inv
        o2, inv2 = by_key(_morton_code(pos2) + (b2 << 30))
        pos2, b2, si1 = pos2[o2], b2[o2], inv2[si1]
        o1, inv1_ = by_key(_morton_code(pos1) + (si1 << 30))
        pos1, b1, si1, si0 = pos1[o1], b1[o1], si1[o1], inv1_[si0]
        o0 = torch.argsort(si0, stable=True)
        pos0, b0, si0 = pos0[o0], b0[o0], si0[o0]
    elif order != "storage":
        raise ValueError("order must be 'storage', 'morton' or 'grouped'")
    if b >= 1:
        # what NAGBatch.from_nag_list knows on the host (Batch.ptr): the clouds' node ranges (a batch
        # of ONE cloud has them too - data.py from_nag_list; without them every step would read the
        # id range of each batch vector back from the device before the folded pre-norm kernels)
        for bt in (b0, b1, b2):
            bt._spt_host_ptr = [0] + torch.cumsum(torch.bincount(bt, minlength=b), 0).tolist()
    ns1 = torch.bincount(si0, minlength=n1)
    ns2 = torch.zeros(n2, dtype=torch.long, device=device).index_add_(0, si1, ns1)
    c1, c2 = (b1, b2) if b > 1 else (None, None)
    if graph == "local":
        ei1 = _edges_local(gen, pos1, e1, device, c1)
        ei2 = _edges_local(gen, pos2, e2, device, c2)
    elif graph == "random":
        ei1 = _edges(gen, n1, e1, device, c1)
        ei2 = _edges(gen, n2, e2, device, c2)
    else:
        raise ValueError("graph must be 'random' or 'local'")

    if b > 1:
        # host knowledge of the batch layout, like the node ranges above: the runs of constant cloud
        # id along every edge index (sorted inside each third of [i<j | j>i | loops]) - what the
        # edge MLP's run table is built from without reading the device back (ops.graph_runs_via)
        for ei_, b_ in ((ei1, b1), (ei2, b2)):
            eb = b_[ei_[0]]
            starts = [0] + (torch.nonzero(eb[1:] != eb[:-1]).flatten() + 1).tolist()
            ends = starts[1:] + [int(eb.numel())]
            ids = eb[torch.tensor(starts, device=device)].tolist()
            ei_._spt_host_runs = ((b, int(eb.numel())), list(zip(starts, ends, ids)))

    def sub_of(si, n_parent):
        """The level's cluster CSR as a stored NAG carries it next to ``super_index``
        (``nag[i+1].sub``, src/data/cluster.py:19-77): children of every cluster ascending."""
        from .data import Cluster
        ptr = torch.zeros(n_parent + 1, dtype=torch.long, device=device)
        ptr[1:] = torch.cumsum(torch.bincount(si, minlength=n_parent), 0)
        return Cluster(ptr, torch.argsort(si, stable=True), ascending=True)

    levels = [
        dict(pos=pos0, x=torch.rand(n0, point_dim, generator=gen, device=device),
             super_index=si0, batch=b0),
        dict(pos=pos1, x=rnd(n1, segment_dim) if segment_dim else None, super_index=si1, batch=b1, node_size=ns1,
             edge_index=ei1, edge_attr=rnd(ei1.shape[1], edge_dim, s=0.3), sub=sub_of(si0, n1)),
        dict(pos=pos2, x=rnd(n2, segment_dim) if segment_dim else None, super_index=None, batch=b2, node_size=ns2,
             edge_index=ei2, edge_attr=rnd(ei2.shape[1], edge_dim, s=0.3), sub=sub_of(si1, n2)),
    ]
    return SyntheticNAG(levels, b)


def make_voxel_cloud(n, voxel=0.03, seed=1234, device="cpu", patch=3.0, extent=(50.0, 50.0, 5.0)):
    """Voxelised-surface point cloud for the preprocessing leg (kNN + geometric
    features): axis-aligned planar patches of ``patch`` metres sampled on a
    ``voxel`` lattice (what GridSampling3D leaves of walls / floors / roofs),
    placed at random in ``extent``, de-duplicated per voxel, rows shuffled.
    Returns [<= n, 3] f32 (a few percent fewer than ``n`` where patches cross)."""
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    side = max(int(patch / voxel), 2)
    per = side * side
    npatch = (n + per - 1) // per
    centers = torch.rand(npatch, 3, generator=gen, device=device) * \
        torch.tensor(extent, device=device)
    centers = (centers / voxel).round() * voxel          # patches live on the voxel lattice
    normal = torch.randint(0, 3, (npatch,), generator=gen, device=device)
    ij = torch.arange(per, device=device)
    i = (ij // side).float() * voxel
    j = (ij % side).float() * voxel
    # in-plane axes for each normal direction
    u_axis = torch.tensor([1, 0, 0], device=device)[normal]   # axis index of u
    v_axis = torch.tensor([2, 2, 1], device=device)[normal]   # axis index of v
    pos = centers.view(npatch, 1, 3).repeat(1, per, 1)
    pos.scatter_add_(2, u_axis.view(-1, 1, 1).expand(npatch, per, 1), i.view(1, per, 1).expand(npatch, per, 1))
    pos.scatter_add_(2, v_axis.view(-1, 1, 1).expand(npatch, per, 1), j.view(1, per, 1).expand(npatch, per, 1))
    pos = pos.view(-1, 3)[:n]
    # one point per voxel, like GridSampling3D's output (patches may overlap)
    q = (pos / voxel).round().long()
    q = q - q.min(dim=0).values
    key = q[:, 0] + (q[:, 1] << 21) + (q[:, 2] << 42)
    order = key.argsort()
    ks = key[order]
    first = torch.ones_like(ks, dtype=torch.bool)
    first[1:] = ks[1:] != ks[:-1]
    pos = pos[order[first]]
    return pos[torch.randperm(pos.shape[0], generator=gen, device=device)].contiguous()


def make_raw_nag(scene="R", seed=1234, device="cpu", sizes=None, point_dim=8):
    """A NAG as it sits on disk BEFORE the per-batch on-device transforms
    (configs/datamodule/semantic/default.yaml:206-290): trimmed (i < j) horizontal edges
    with the 7 stored edge attributes, segment attributes (normal, log_length,
    log_surface, log_volume, log_size), the cluster CSR of every level, no node_size.
    Returns a ``superpoint_transformer_amd.data.NAG``."""
    from .data import NAG, Data, Cluster
    n0, n1, n2, e1, e2, b = sizes if sizes is not None else SCENES[scene]
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    si0 = _super_index(gen, n0, n1, "lognormal", device, shuffle=True)
    si1 = _super_index(gen, n1, n2, "geometric", device, shuffle=True)

    def rnd(*shape, s=1.0):
        return torch.randn(*shape, generator=gen, device=device) * s

    pos2 = torch.rand(n2, 3, generator=gen, device=device) * 40.0
    pos1 = pos2[si1] + rnd(n1, 3, s=1.5)
    pos0 = pos1[si0] + rnd(n0, 3, s=0.3)

    def trimmed(n, e_target):
        m = max((e_target - n) // 2, 1)
        a = torch.randint(0, n, (m,), generator=gen, device=device)
        c = torch.randint(0, n, (m,), generator=gen, device=device)
        lo, hi = torch.minimum(a, c), torch.maximum(a, c)
        key = torch.unique(lo[lo != hi] * n + hi[lo != hi])
        return torch.stack([key // n, key % n])

    def segment_level(n, pos, si, sub_index, sub_n, e_target):
        ei = trimmed(n, e_target)
        nrm = torch.nn.functional.normalize(rnd(n, 3), dim=1)
        nrm = nrm * torch.where(nrm[:, 2:3] < 0, -1.0, 1.0)
        d = Data(pos=pos, normal=nrm, log_length=torch.rand(n, 1, generator=gen, device=device),
                 log_surface=torch.rand(n, 1, generator=gen, device=device),
                 log_volume=torch.rand(n, 1, generator=gen, device=device),
                 log_size=torch.rand(n, 1, generator=gen, device=device),
                 sub=Cluster(sub_index, torch.arange(sub_n, device=device), dense=True, ascending=True),
                 edge_index=ei, edge_attr=rnd(ei.shape[1], 7, s=0.3))
        if si is not None:
            d.super_index = si
        return d

    return NAG([
        Data(pos=pos0, x=torch.rand(n0, point_dim, generator=gen, device=device), super_index=si0),
        segment_level(n1, pos1, si1, si0, n0, e1),
        segment_level(n2, pos2, None, si1, n1, e2)])
