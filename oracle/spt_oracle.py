"""CPU oracle for the SPT hot path -- TEST INFRASTRUCTURE ONLY.

Plain torch-CPU / numpy restatement of the reference's algorithms for the
path of SURVEY.md section 8.  Only ``tests/``, ``__graft_entry__.smoke()``,
``tests/golden/make_golden.py`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``superpoint_transformer_amd``) never does.

Every function cites the reference file:line it follows (paths relative to
drprojects/superpoint_transformer v3.0.0).  Functions whose algorithm lives
in an UN-VENDORED third-party dependency (torch_scatter, torch_geometric
2.3.0, FRNN, pgeof -- install.sh:96-137; none installable here) restate the
published semantics and are marked "[third-party restated]".

Pinning status:
  * pinned against the reference's OWN Python code: ``tests/golden/
    make_golden.py`` imports the reference's modules verbatim (attention.py,
    norm.py, pool.py, unpool.py, transformer.py, utils/scatter.py,
    utils/nn.py, utils/geometry.py, utils/neighbors.py:knn_brute_force /
    neighbors_dense_to_csr) from /root/reference and stores their outputs as
    fixtures which ``tests/test_oracle_golden.py`` checks this file against;
  * PARITY UNPINNED at the third-party boundary: the torch_scatter / PyG /
    FRNN / pgeof binaries cannot be run here, and the reference's tests hold
    no golden vector for them.

All functions accept float32 or float64 tensors; tests evaluate the oracle in
float64 to get a tolerance reference for the float32 kernels.
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------
# torch_scatter                                       [third-party restated]
# --------------------------------------------------------------------------


def _dim_size(index, dim_size):
    if dim_size is not None:
        return int(dim_size)
    return int(index.max()) + 1 if index.numel() > 0 else 0


def _expand_index(index, src):
    idx = index.view((-1,) + (1,) * (src.dim() - 1))
    return idx.expand_as(src)


def scatter_sum(src, index, dim=0, out=None, dim_size=None):
    """torch_scatter.scatter_sum along dim 0 with a 1-D index (call sites:
    src/nn/attention.py:315, src/data/nag.py:97,108, src/utils/scatter.py:30)."""
    assert dim == 0 and out is None and index.dim() == 1
    n = _dim_size(index, dim_size)
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    if src.numel():
        res.index_add_(0, index, src)
    return res


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    """torch_scatter.scatter_mean: sum / clamp(count, 1); integer src floors
    (call sites: src/utils/scatter.py:59, src/nn/norm.py:125)."""
    n = _dim_size(index, dim_size)
    s = scatter_sum(src, index, dim, out, n)
    cnt = torch.bincount(index, minlength=n).clamp(min=1)
    cnt = cnt.view((-1,) + (1,) * (src.dim() - 1))
    if src.is_floating_point():
        return s / cnt.to(src.dtype)
    return torch.div(s, cnt, rounding_mode="floor")


def _scatter_minmax(src, index, dim_size, is_max):
    """torch_scatter.scatter_{min,max}: empty groups -> 0 and arg = src.size(0);
    ties -> FIRST occurrence (torch_scatter's CPU kernel walks rows in order
    with a strict comparison)."""
    n = _dim_size(index, dim_size)
    N = src.shape[0]
    flat = src.reshape(N, -1)
    C = flat.shape[1]
    out = torch.zeros((n, C), dtype=src.dtype)
    arg = torch.full((n, C), N, dtype=torch.int64)
    if N:
        if src.is_floating_point():
            lim = float("-inf") if is_max else float("inf")
        else:                                    # integer sources (src/data/instance.py:192)
            lim = torch.iinfo(src.dtype).min if is_max else torch.iinfo(src.dtype).max
        red = torch.full((n, C), lim, dtype=src.dtype)
        red.scatter_reduce_(0, index.view(-1, 1).expand(N, C), flat,
                            "amax" if is_max else "amin", include_self=True)
        hit = flat == red[index]
        rows = torch.arange(N).view(-1, 1).expand(N, C)
        cand = torch.where(hit, rows, torch.full_like(rows, N))
        arg.scatter_reduce_(0, index.view(-1, 1).expand(N, C), cand, "amin",
                            include_self=True)
        nonempty = torch.bincount(index, minlength=n) > 0
        out[nonempty] = red[nonempty]
    shape = (n,) + tuple(src.shape[1:])
    return out.reshape(shape), arg.reshape(shape)


class _ScatterMinMax(torch.autograd.Function):
    """Differentiable wrapper with torch_scatter's gradient rule: the whole
    output gradient goes to the single arg element (no tie splitting)."""

    @staticmethod
    def forward(ctx, src, index, dim_size, is_max):
        out, arg = _scatter_minmax(src.detach(), index, dim_size, is_max)
        ctx.n = src.shape[0]
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, gout, _garg):
        (arg,) = ctx.saved_tensors
        return scatter_max_grad(gout, arg, ctx.n), None, None, None


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0 and out is None
    return _ScatterMinMax.apply(src, index, dim_size, True)


def scatter_min(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0 and out is None
    return _ScatterMinMax.apply(src, index, dim_size, False)


def scatter_std(src, index, dim=0, out=None, dim_size=None, unbiased=True):
    """torch_scatter.scatter_std (src/transforms/graph.py:285,1036)."""
    n = _dim_size(index, dim_size)
    cnt = torch.bincount(index, minlength=n).to(src.dtype)
    cnt = cnt.view((-1,) + (1,) * (src.dim() - 1))
    mean = scatter_sum(src, index, 0, None, n) / cnt.clamp(min=1)
    var = scatter_sum((src - mean[index]) ** 2, index, 0, None, n)
    denom = (cnt - 1 if unbiased else cnt).clamp(min=1)
    return (var / (denom + 1e-6)).sqrt()


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter (src/nn/norm.py:118-126)."""
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == "min":
        return scatter_min(src, index, dim, out, dim_size)[0]
    if reduce == "max":
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)


def scatter_max_grad(gout, arg, n):
    """Gradient rule of torch_scatter min/max: all of gout goes to the arg row."""
    C = gout.reshape(gout.shape[0], -1).shape[1]
    g = torch.zeros((n + 1, C), dtype=gout.dtype)
    g.scatter_add_(0, arg.reshape(-1, C), gout.reshape(-1, C))
    return g[:n].reshape((n,) + tuple(gout.shape[1:]))


# --------------------------------------------------------------------------
# CSR view (what spt_csr_build must produce)
# --------------------------------------------------------------------------


def csr_view(index, num_seg):
    """Stable argsort + row pointers of an unsorted segment index."""
    perm = torch.sort(index, stable=True).indices
    counts = torch.bincount(index, minlength=num_seg)
    rowptr = torch.zeros(num_seg + 1, dtype=torch.int64)
    rowptr[1:] = counts.cumsum(0)
    return perm.to(torch.int32), rowptr.to(torch.int32)


# --------------------------------------------------------------------------
# torch_geometric 2.3.0                                [third-party restated]
# --------------------------------------------------------------------------


def pyg_softmax(src, index, ptr=None, num_nodes=None, dim=0):
    """torch_geometric.utils.softmax (src/nn/attention.py:307): max-subtract,
    exp, segment sum + 1e-16, divide."""
    assert dim == 0 and ptr is None
    n = _dim_size(index, num_nodes)
    src_max = scatter(src.detach(), index, 0, None, n, "max")
    out = (src - src_max[index]).exp()
    out_sum = scatter_sum(out, index, 0, None, n) + 1e-16
    return out / out_sum[index]


def graph_norm(x, batch, weight, bias, mean_scale, eps=1e-5, batch_size=None):
    """torch_geometric.nn.norm.GraphNorm.forward (used through
    src/nn/mlp.py:85-94 and src/nn/transformer.py:258-265)."""
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_size is None:
        batch_size = int(batch.max()) + 1 if batch.numel() else 1
    mean = scatter_mean(x, batch, 0, None, batch_size)
    out = x - mean[batch] * mean_scale
    var = scatter_mean(out * out, batch, 0, None, batch_size)
    std = (var + eps).sqrt()[batch]
    return weight * out / std + bias


# --------------------------------------------------------------------------
# src/utils/scatter.py, src/nn/norm.py, src/nn/unpool.py, src/nn/pool.py
# --------------------------------------------------------------------------


def scatter_mean_weighted(x, idx, w, dim_size=None):
    """src/utils/scatter.py:17-38."""
    x = x.view(-1, 1) if x.dim() == 1 else x
    w = w.view(-1, 1).to(x.dtype)
    wx = torch.cat((w, x * w), dim=1)
    seg = scatter_sum(wx, idx, 0, None, dim_size)
    w_seg = seg[:, 0].clone()
    w_seg[w_seg == 0] = 1
    return seg[:, 1:] / w_seg.view(-1, 1)


def unit_sphere_norm(pos, idx, w=None, num_super=None, log_diameter=False):
    """src/nn/norm.py:67-138 (UnitSphereNorm.forward)."""
    if idx is None:  # norm.py:86-110
        mn = pos.min(dim=0).values
        mx = pos.max(dim=0).values
        diameter = (mx - mn).max()
        if w is None:
            center = pos.mean(dim=0)
        else:
            ws = w.to(pos.dtype).sum()
            ws = 1 if ws == 0 else ws
            center = (pos * w.view(-1, 1).to(pos.dtype)).sum(dim=0) / ws
        out = (pos - center.view(1, -1)) / (diameter + 1e-2)
        diameter = diameter.view(1, 1)
    else:  # norm.py:112-138
        mn = scatter(pos, idx, 0, None, num_super, "min")
        mx = scatter(pos, idx, 0, None, num_super, "max")
        diam_seg = (mx - mn).max(dim=1).values
        if w is None:
            center_seg = scatter(pos, idx, 0, None, num_super, "mean")
        else:
            center_seg = scatter_mean_weighted(pos, idx, w, num_super)
        out = (pos - center_seg[idx]) / (diam_seg[idx].view(-1, 1) + 1e-2)
        diameter = diam_seg.view(-1, 1)
    if log_diameter:
        diameter = torch.log(diameter + 1)
    return out, diameter


def pool(x_child, index, num_pool, mode="max"):
    """src/nn/pool.py:61-82 -> PyG Aggregation.reduce -> scatter."""
    return scatter(x_child, index, 0, None, num_pool, mode)


def index_unpool(x, idx):
    """src/nn/unpool.py:12-13."""
    return x.index_select(0, idx)


def get_sub_size(super_indices, node_size_low=None):
    """src/data/nag.py:59-110: chained integer scatter_sum up the hierarchy.
    ``super_indices[i]`` maps level low+i to low+i+1.  Returns the list of
    sizes for levels low+1 .. low+len(super_indices)."""
    out = []
    si0 = super_indices[0]
    n1 = int(si0.max()) + 1
    if node_size_low is not None:
        sizes = scatter_sum(node_size_low, si0, 0, None, n1)
    else:
        sizes = torch.bincount(si0, minlength=n1)
    out.append(sizes)
    for si in super_indices[1:]:
        sizes = scatter_sum(sizes, si, 0, None, int(si.max()) + 1)
        out.append(sizes)
    return out


# --------------------------------------------------------------------------
# src/utils/nn.py, src/nn/attention.py
# --------------------------------------------------------------------------


def qk_scale_dg(s, dim, num_heads):
    """src/utils/nn.py:83-88 (qk_scale=None -> 'd.g')."""
    D = (dim // num_heads) ** -0.5
    # int64 ** -0.5 yields the DEFAULT dtype (float32) in the reference, even
    # when the block itself runs in float64: keep that rounding.
    G = (s.bincount() ** -0.5)[s].view(-1, 1, 1)
    return D * G


def self_attention(x, edge_index, edge_attr, p, num_heads, qk_dim):
    """src/nn/attention.py:167-325 for the SPT configuration family
    (k_rpe/q_rpe/v_rpe Linear on edge_attr, no delta-RPE, no sharing, no
    dropout, qk_scale=None).  ``p`` maps parameter names to tensors:
    qkv.weight/bias, k_rpe.*, q_rpe.*, v_rpe.* (each optional), out_proj.*
    (optional)."""
    N, E, H, D = x.shape[0], edge_index.shape[1], num_heads, qk_dim
    DH = D * H
    qkv = x @ p["qkv.weight"].t()
    if p.get("qkv.bias") is not None:
        qkv = qkv + p["qkv.bias"]
    dim = qkv.shape[1] - 2 * DH
    q = qkv[:, :DH].view(N, H, D)
    k = qkv[:, DH:2 * DH].view(N, H, D)
    v = qkv[:, 2 * DH:].view(N, H, -1)
    s, t = edge_index[0], edge_index[1]
    q, k, v = q[s], k[t], v[t]
    q = q * qk_scale_dg(s, dim, H).to(q.dtype)        # attention.py:214

    def lin(name):
        y = edge_attr @ p[name + ".weight"].t()
        if p.get(name + ".bias") is not None:
            y = y + p[name + ".bias"]
        return y

    if p.get("k_rpe.weight") is not None and edge_attr is not None:
        k = k + lin("k_rpe").view(E, H, -1)            # attention.py:225-232
    if p.get("q_rpe.weight") is not None and edge_attr is not None:
        q = q + lin("q_rpe").view(E, H, -1)            # attention.py:235-245
    if p.get("v_rpe.weight") is not None and edge_attr is not None:
        v = v + lin("v_rpe").view(E, H, -1)            # attention.py:294-301
    compat = torch.einsum("ehd,ehd->eh", q, k)         # attention.py:304
    attn = pyg_softmax(compat, s, num_nodes=N)         # attention.py:307
    out = (v * attn.unsqueeze(-1)).reshape(E, dim)     # attention.py:314
    out = scatter_sum(out, s, 0, None, N)              # attention.py:315
    if p.get("out_proj.weight") is not None:           # attention.py:318-319
        out = out @ p["out_proj.weight"].t() + p["out_proj.bias"]
    return out


def qk_scale(s, dim, num_heads, spec=None):
    """src/utils/nn.py:75-127: the scale of the queries of group ``s`` for every spelling of
    ``qk_scale`` (None and 'd.g' = D * G, 'd+g', 'd', 'g', or a number used as is)."""
    if spec is not None and not isinstance(spec, str):
        return spec
    D = (dim // num_heads) ** -0.5
    G = (s.bincount() ** -0.5)[s].view(-1, 1, 1)
    key = "dg" if spec is None else spec.lower().replace(" ", "")
    if key in ("d+g", "g+d"):
        return D + G
    if key in ("dg", "gd", "d*g", "g*d", "d.g", "g.d"):
        return D * G
    if key == "d":
        return D
    if key == "g":
        return G
    raise ValueError(spec)


def attentive_pool(x_child, query, index, edge_attr, p, num_heads, qk_dim, num_pool,
                   spec=None, heads_share_rpe=False):
    """src/nn/pool.py:160-245: ``query`` [Np, H*D] is what ``_get_query`` returned; ``p`` maps
    kv.*, k_rpe.*, q_rpe.*, in_proj.*, out_proj.* (each optional but kv) to tensors."""
    H, D = num_heads, qk_dim

    def lin(name, x):
        y = x @ p[name + ".weight"].t()
        if p.get(name + ".bias") is not None:
            y = y + p[name + ".bias"]
        return y

    if p.get("in_proj.weight") is not None:                   # pool.py:177-178
        x_child = lin("in_proj", x_child)
    nc = x_child.shape[0]
    kv = lin("kv", x_child)
    dim = kv.shape[1] - D * H
    q = query[index].view(nc, H, D)                            # pool.py:187-189
    k = kv[:, :D * H].view(nc, H, D)
    v = kv[:, D * H:].view(nc, H, -1)
    sc = qk_scale(index, dim, H, spec)
    q = q * (sc.to(q.dtype) if torch.is_tensor(sc) else sc)   # pool.py:192

    def rpe(name):
        r = lin(name, edge_attr)
        return (r.repeat(1, H) if heads_share_rpe else r).view(nc, H, -1)

    if p.get("k_rpe.weight") is not None:                      # pool.py:203-219
        k = k + rpe("k_rpe")
    if p.get("q_rpe.weight") is not None:
        q = q + rpe("q_rpe")
    compat = torch.einsum("nhd,nhd->nh", q, k)                 # pool.py:222
    attn = pyg_softmax(compat, index, num_nodes=num_pool)      # pool.py:225
    x = scatter_sum((v * attn.unsqueeze(-1)).reshape(nc, dim), index, 0, None, num_pool)
    if p.get("out_proj.weight") is not None:                   # pool.py:236-237
        x = lin("out_proj", x)
    return x


# --------------------------------------------------------------------------
# kNN: src/utils/neighbors.py
# --------------------------------------------------------------------------


def knn_brute_force(x_search, x_query, k, r_max=1.0):
    """src/utils/neighbors.py:245-295 without batches: full distance matrix,
    sort, first k, > r_max -> -1.  Ties keep torch.sort's order."""
    d = (x_search.unsqueeze(0) - x_query.unsqueeze(1)).norm(dim=2)
    d, nb = d.sort(dim=1, stable=True)
    d, nb = d[:, :k].clone(), nb[:, :k].clone()
    mask = d > r_max
    d[mask] = -1
    nb[mask] = -1
    return nb, d


def frnn_grid_points(query, search, K, r, strict=True):
    """Contract of the un-vendored FRNN CUDA extension as the reference uses
    it (src/utils/neighbors.py:24-48,90) [third-party restated]: for each
    query the K nearest search points with squared distance < r^2, ascending
    (ties: ascending index), idx padded with -1 and dists with -1.
    Distances are SQUARED (pytorch3d convention FRNN inherits).  Computed in
    float32 exactly like the kernel: d2 = dx*dx + dy*dy + dz*dz, sequential,
    no fma."""
    q = query.to(torch.float32)
    s = search.to(torch.float32)
    nq = q.shape[0]
    idx = torch.full((nq, K), -1, dtype=torch.int64)
    dist = torch.full((nq, K), -1.0, dtype=torch.float32)
    r2 = np.float32(r) * np.float32(r)
    sn = s.numpy()
    chunk = max(1, (1 << 24) // max(1, s.shape[0]))
    for a in range(0, nq, chunk):
        qq = q[a:a + chunk].numpy()
        dx = qq[:, None, 0] - sn[None, :, 0]
        dy = qq[:, None, 1] - sn[None, :, 1]
        dz = qq[:, None, 2] - sn[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz          # float32, same order as kernel
        ok = d2 < r2 if strict else d2 <= r2
        d2m = np.where(ok, d2, np.float32(np.inf))
        order = np.argsort(d2m, axis=1, kind="stable")[:, :K]
        dsel = np.take_along_axis(d2m, order, axis=1)
        good = np.isfinite(dsel)
        kk = order.shape[1]
        idx[a:a + chunk, :kk] = torch.from_numpy(np.where(good, order, -1))
        dist[a:a + chunk, :kk] = torch.from_numpy(
            np.where(good, dsel, np.float32(-1)).astype(np.float32))
    return dist, idx


def knn_1(xyz, k, r_max=1.0, self_is_neighbor=False):
    """src/utils/neighbors.py:51-123 without batch/oversample: search k+1 and
    drop column 0 (the point itself at distance 0)."""
    k_search = k if self_is_neighbor else k + 1
    dist, idx = frnn_grid_points(xyz, xyz, k_search, r_max)
    if self_is_neighbor:
        return idx, dist
    return idx[:, 1:], dist[:, 1:]


def neighbors_dense_to_csr(nn):
    """src/utils/neighbors.py:668-684."""
    k = nn.shape[1]
    mask = nn < 0
    sizes = k - mask.sum(dim=1)
    ptr = torch.zeros(nn.shape[0] + 1, dtype=torch.int64)
    ptr[1:] = sizes.cumsum(0)
    return ptr, nn[~mask], sizes


# --------------------------------------------------------------------------
# geometric features: src/utils/scatter.py:41-125, src/utils/geometry.py
# --------------------------------------------------------------------------


def scatter_pca(x, idx, num_groups=None):
    """src/utils/scatter.py:41-125 (algorithm='eigh'): population covariance
    of each group, eigh(UPLO='U'), ascending eigenvalues, NaN -> (1,1,1)/I,
    eigenvalues clamped at 0."""
    n = _dim_size(idx, num_groups)
    mean = scatter_mean(x, idx, 0, None, n)
    xc = x - mean[idx]
    ij = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    ut = torch.stack([xc[:, i] * xc[:, j] for i, j in ij], dim=1)
    sizes = torch.bincount(idx, minlength=n)
    ut = scatter_sum(ut, idx, 0, None, n) / sizes.view(-1, 1).to(x.dtype)
    cov = torch.zeros((n, 3, 3), dtype=x.dtype)
    for c, (i, j) in enumerate(ij):
        cov[:, i, j] = ut[:, c]
        cov[:, j, i] = ut[:, c]
    bad = torch.isnan(cov).flatten(1).any(1)
    cov[bad] = torch.eye(3, dtype=x.dtype)
    val, vec = torch.linalg.eigh(cov, UPLO="U")
    val[bad] = 1.0
    vec[bad] = torch.eye(3, dtype=x.dtype)
    return val.clamp(min=0), vec


def geometric_features(xyz, nn, k_min=1, add_self_as_neighbor=True):
    """src/utils/geometry.py:80-126 + :236-338 with k_step=-1 (the dataset
    configs' setting, configs/datamodule/semantic/default.yaml:121-127):
    returns f32[N,11] in pgeof's column order (geometry.py:165-174)
    [lin, plan, scat, vert, nx, ny, nz, length, surface, volume, curvature]
    AFTER the post-processing of geometry.py:121,124 (verticality*2, normal
    flipped to z>=0)."""
    N = nn.shape[0]      # one feature row per neighbourhood (== xyz rows when add_self)
    if add_self_as_neighbor:                                   # geometry.py:95-96
        nn = torch.cat((torch.arange(N).view(-1, 1), nn), dim=1)
    ptr, val, sizes = neighbors_dense_to_csr(nn)               # geometry.py:347
    idx = torch.repeat_interleave(torch.arange(N), ptr[1:] - ptr[:-1])
    eigenval, eigenvec = scatter_pca(xyz[val], idx, N)         # geometry.py:354
    normal = eigenvec[:, :, 0].clone()                         # geometry.py:290
    l1 = eigenval[:, 2].sqrt()
    l2 = eigenval[:, 1].sqrt()
    l3 = eigenval[:, 0].sqrt()
    lin = (l1 - l2) / (l1 + 1e-3)                              # geometry.py:300-306
    plan = (l2 - l3) / (l1 + 1e-3)
    scat = l3 / (l1 + 1e-3)
    length = l1
    surface = (l1 * l2 + 1e-6).sqrt()
    volume = (l1 * l2 * l3 + 1e-9).pow(1 / 3)
    curv = l3 / (l1 + l2 + l3 + 1e-3)
    unary = (eigenvec.abs() * eigenval.unsqueeze(1)).sum(dim=2)  # geometry.py:308
    vert = unary[:, 2] / (unary.norm(dim=1) + 1e-8)
    small = sizes < k_min                                      # geometry.py:312-321
    f = torch.stack([lin, plan, scat, vert, normal[:, 0], normal[:, 1],
                     normal[:, 2], length, surface, volume, curv], dim=1)
    f[small] = 0
    f[:, 3] *= 2                                               # geometry.py:121
    flip = f[:, 6] < 0                                         # geometry.py:124
    f[flip, 4:7] *= -1
    return f


def geometric_features_optimal(xyz, nn, k_min=1, k_step=1, k_min_search=1,
                               add_self_as_neighbor=True, return_margin=False):
    """src/utils/geometry.py:248-287 (k_step >= 0): ``geometric_features`` of the first k
    neighbours for k = k0, the multiples of k_step and k_max, keeping per point the size whose
    eigenvalues have the lowest eigenentropy (strict comparison: the first such size on
    ties).  ``return_margin``: also the gap between the lowest and second lowest entropy of
    each point over its DISTINCT neighbourhoods (a partial neighbourhood repeats itself for
    every k past its size - exact ties, resolved by the strict comparison): how safely an
    f32 implementation picks the same size."""
    N = nn.shape[0]
    if add_self_as_neighbor:
        nn = torch.cat((torch.arange(N).view(-1, 1), nn), dim=1)
    k_max = nn.shape[1]
    k0 = max(k_min, k_min_search)
    best = entropy = None
    ents, seen = [], None
    for k in range(k0, k_max + 1):
        if (k > k0) and (k % k_step != 0) and (k != k_max):     # geometry.py:257-258
            continue
        sub = nn[:, :k]
        ptr, val, sizes = neighbors_dense_to_csr(sub)
        idx = torch.repeat_interleave(torch.arange(N), ptr[1:] - ptr[:-1])
        ev, _ = scatter_pca(xyz[val], idx, N)
        e = ev / (ev.sum(dim=1).view(-1, 1) + 1e-3)              # geometry.py:268-270
        ent = (-e * torch.log(e + 1e-3)).sum(dim=1)
        f = geometric_features(xyz, sub, k_min, add_self_as_neighbor=False)
        repeat = torch.zeros(N, dtype=torch.bool) if seen is None else sizes == seen
        ents.append(torch.where(repeat, torch.full_like(ent, float("inf")), ent))
        seen = sizes
        if best is None:
            best, entropy = f, ent
            continue
        better = ent < entropy                                    # geometry.py:282-286
        best[better] = f[better]
        entropy[better] = ent[better]
    if not return_margin:
        return best
    top = torch.stack(ents, dim=1).sort(dim=1).values
    margin = top[:, 1] - top[:, 0] if top.shape[1] > 1 else torch.full((N,), float("inf"))
    return best, margin


# --------------------------------------------------------------------------
# on-the-fly horizontal edge features + self loops
# --------------------------------------------------------------------------


def horizontal_edge_features(se, edge_attr7, pos, normal, log_length, log_surface,
                             log_volume, log_size, add_self_loops=True):
    """src/transforms/graph.py:1135-1277 (all default keys, in f_list order)
    followed by NAGAddSelfLoops :1419-1452 (zero-feature loops appended)."""
    ea = edge_attr7.to(pos.dtype)
    s, t = se[0], se[1]
    mean_off = ea[:, :3]
    direction = mean_off / mean_off.norm(dim=1).view(-1, 1)           # graph.py:1203
    direction = torch.where(direction.isnan(), torch.zeros_like(direction), direction)
    direction = direction.clip(-1, 1)
    cols_a, cols_b = [mean_off], [-mean_off]                           # graph.py:1210-1214

    def both(f):
        cols_a.append(f)
        cols_b.append(f)

    def anti(f):
        cols_a.append(f)
        cols_b.append(-f)

    both(ea[:, 3:6])                                                   # std_off
    both(ea[:, 6:7])                                                   # mean_dist
    both((direction * normal[s]).sum(1).abs().view(-1, 1))             # angle_source
    both((direction * normal[t]).sum(1).abs().view(-1, 1))             # angle_target
    both((normal[s] * normal[t]).sum(1).abs().view(-1, 1))             # normal_angle
    for a in (log_length, log_surface, log_volume, log_size):          # graph.py:1231-1245
        a = a.view(-1)
        anti((a[s] - a[t]).view(-1, 1))
    cdir = pos[t] - pos[s]                                             # graph.py:1250-1257
    cdist = cdir.norm(dim=1).view(-1, 1)
    cdir = cdir / cdist
    cdist = cdist.sqrt()
    cdir = torch.where(cdir.isnan(), torch.zeros_like(cdir), cdir).clip(-1, 1)
    anti(cdir)
    both(cdist)
    attr = torch.cat((torch.cat(cols_a, 1), torch.cat(cols_b, 1)), 0)
    ei = torch.cat((se, se.flip(0)), dim=1)                            # graph.py:1268
    if add_self_loops:                                                 # graph.py:1442-1446
        n = pos.shape[0]
        loops = torch.arange(n)
        ei = torch.cat((ei, torch.stack((loops, loops))), dim=1)
        attr = torch.cat((attr, torch.zeros((n, attr.shape[1]), dtype=attr.dtype)), 0)
    return ei, attr


# --------------------------------------------------------------------------
# segment-level preprocessing
# --------------------------------------------------------------------------


def sparse_sample_counts(idx, n_max=32, n_min=1, mask=None):
    """The deterministic part of sparse_sample (src/utils/sparse.py:168-205,
    237-243): how many elements each segment contributes and the resulting
    pointers.  (Which elements are drawn is random: randperm + stable sort,
    sparse.py:217-231.)"""
    assert 0 <= n_min <= n_max
    size = idx.bincount()
    num_segments = int(idx.max()) + 1
    if n_max > 0:
        n_samples = (n_max * torch.tanh(size / n_max)).floor().long()
    else:
        n_samples = size.sqrt().round().long()
    n_samples = n_samples.clamp(min=n_min).clamp(max=size)
    if mask is not None:
        size = idx[mask].bincount(minlength=num_segments)
        n_samples = n_samples.clamp(max=size)
    ptr = torch.cat((torch.zeros(1, dtype=torch.long), n_samples)).cumsum(0)
    return n_samples, ptr


def check_sparse_sample(idx, samples, ptr, n_max, n_min, mask=None):
    """Contract of sparse_sample's output, as a list of violated clauses (empty =
    ok): pointers as above, every sample a member of its segment, no duplicate,
    none masked out."""
    bad = []
    n_samples, ptr_ref = sparse_sample_counts(idx, n_max, n_min, mask)
    if not torch.equal(ptr, ptr_ref):
        bad.append("pointers")
    if samples.numel() != int(ptr_ref[-1]):
        bad.append("total")
        return bad
    seg = torch.repeat_interleave(torch.arange(n_samples.numel()), n_samples)
    if not torch.equal(idx[samples], seg):
        bad.append("membership")
    if torch.unique(samples).numel() != samples.numel():
        bad.append("duplicates")
    if mask is not None and not bool(mask[samples].all()):
        bad.append("mask")
    return bad


def scatter_mean_orientation(orientation, idx, num_groups=None):
    """src/utils/scatter.py:249-300."""
    n = _dim_size(idx, num_groups)
    eps = 1e-4
    x = orientation.clone()
    x = x / (x.norm(dim=1).view(-1, 1) + eps)
    x = x.clamp(min=-1, max=1)
    phi = x[:, 2].arcsin()
    phi_mean = scatter_mean(phi, idx, 0, None, n)
    is_horizontal = (phi_mean < math.pi / 4)[idx]
    _, argmin = scatter_min(phi, idx, 0, None, n)
    is_opposing = (x * x[argmin[idx]]).sum(dim=1) < 0
    flip = is_horizontal & is_opposing
    x = torch.where(flip.view(-1, 1), -x, x)
    x_mean = scatter_mean(x, idx, 0, None, n)
    x_mean = x_mean / (x_mean.norm(dim=1).view(-1, 1) + eps)
    x_mean = x_mean.clamp(min=-1, max=1)
    neg = x_mean[:, -1] < 0
    return torch.where(neg.view(-1, 1), -x_mean, x_mean)


def segment_features(pos, super_index, num_segments, samples, ptr, sub_size=None,
                     point_attrs=None):
    """_compute_cluster_features (src/transforms/graph.py:193-321) for given
    samples: geometric features of the sampled points of each segment (k_min = 5,
    the default of geometric_features, no self), log_ variants, log_size, mean_ /
    std_ of the point attributes (mean_normal = scatter_mean_orientation)."""
    width = ptr[1:] - ptr[:-1]
    kmax = int(width.max()) if width.numel() else 0
    nn = torch.full((num_segments, max(kmax, 1)), -1, dtype=torch.long)
    rows = torch.repeat_interleave(torch.arange(num_segments), width)
    cols = torch.arange(samples.numel()) - ptr[:-1][rows]
    nn[rows, cols] = samples                                    # csr_to_dense, graph.py:239-242
    f = geometric_features(pos, nn, k_min=5, add_self_as_neighbor=False)
    out = dict(linearity=f[:, 0:1], planarity=f[:, 1:2], scattering=f[:, 2:3],
               verticality=f[:, 3:4], normal=f[:, 4:7], curvature=f[:, 10:11],
               log_length=torch.log(f[:, 7:8] + 1), log_surface=torch.log(f[:, 8:9] + 1),
               log_volume=torch.log(f[:, 9:10] + 1))
    if sub_size is None:
        sub_size = torch.bincount(super_index, minlength=num_segments)
    out["log_size"] = (torch.log(sub_size + 1).view(-1, 1) - np.log(2)) / 10
    for key, a in (point_attrs or {}).items():
        if key == "normal":
            out[f"mean_{key}"] = scatter_mean_orientation(a, super_index, num_segments)
        else:
            out[f"mean_{key}"] = scatter_mean(a, super_index, 0, None, num_segments)
        out[f"std_{key}"] = scatter_std(a, super_index, 0, None, num_segments)
    return out


# --------------------------------------------------------------------------
# cluster radius graph: src/utils/neighbors.py:491-665
# --------------------------------------------------------------------------


def consecutive_cluster(src):
    """torch_geometric.nn.pool.consecutive.consecutive_cluster  [third-party restated]:
    dense relabelling in sorted order + for every new label the position of one
    element carrying it (the LAST one: scatter_ of an arange on CPU)."""
    unique, inv = torch.unique(src, sorted=True, return_inverse=True)
    perm = torch.empty(unique.numel(), dtype=torch.long)
    perm[inv] = torch.arange(inv.numel())
    return inv, perm


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce="sum"):
    """torch_geometric.utils.coalesce  [third-party restated]: edges sorted by
    (row, col), duplicates merged with `reduce`."""
    n = int(num_nodes) if num_nodes is not None else (
        int(edge_index.max()) + 1 if edge_index.numel() else 0)
    key = edge_index[0] * max(n, 1) + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    ei = torch.stack([uniq // max(n, 1), uniq % max(n, 1)])
    if edge_attr is None:
        return ei
    red = "sum" if reduce in ("add", "sum") else reduce
    return ei, scatter(edge_attr, inv, 0, None, uniq.numel(), red)


def remove_self_loops(edge_index, edge_attr=None):
    """torch_geometric.utils.remove_self_loops  [third-party restated]."""
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def to_trimmed(edge_index, edge_attr=None, reduce="mean"):
    """src/utils/graph.py:466-502."""
    edge_index = edge_index.clone()
    flip = edge_index[0] > edge_index[1]
    edge_index[:, flip] = edge_index[:, flip].flip(0)
    if edge_attr is None:
        return remove_self_loops(coalesce(edge_index))[0]
    edge_index, edge_attr = coalesce(edge_index, edge_attr, reduce=reduce)
    return remove_self_loops(edge_index, edge_attr)


def scatter_nearest_neighbor(points, index, edge_index, cycles=3):
    """src/utils/scatter.py:128-238, edge by edge instead of through the
    edge_wise_points expansion: candidates start at the cluster centroids; one cycle =
    closest point of t to s's candidate, then closest point of s to t's candidate;
    ties -> first point of the cluster in ascending index order (the order of
    indices_to_pointers' sort, src/utils/edge.py:56,69)."""
    n = int(index.max()) + 1
    centroid = scatter_mean(points, index, 0, None, n)
    order = torch.argsort(index, stable=True)
    ptr = torch.cat((torch.zeros(1, dtype=torch.long), torch.bincount(index, minlength=n))).cumsum(0)
    E = edge_index.shape[1]
    cand = torch.empty((2, E), dtype=torch.long)
    for e in range(E):
        s, t = int(edge_index[0, e]), int(edge_index[1, e])
        ps, pt = order[ptr[s]:ptr[s + 1]], order[ptr[t]:ptr[t + 1]]
        sc, tc = centroid[s], centroid[t]
        for _ in range(cycles):
            ti = pt[(points[pt] - sc).norm(dim=1).argmin()]     # argmin: first minimum
            tc = points[ti]
            si = ps[(points[ps] - tc).norm(dim=1).argmin()]
            sc = points[si]
        cand[0, e], cand[1, e] = si, ti
    return cand


def cluster_radius_nn_graph(x_points, idx, k_max=100, gap=0, batch=None, trim=True, cycles=3,
                            squared=True):
    """src/utils/neighbors.py:491-665.  `squared`: convention of the FRNN distances the
    reference compares against the radius sum at :591-593 (upstream FRNN returns
    squared distances; unverifiable here - see DESIGN.md)."""
    n = int(idx.max()) + 1
    lo = scatter(x_points, idx, 0, None, n, "min")
    hi = scatter(x_points, idx, 0, None, n, "max")
    diam = (hi - lo).max(dim=1).values
    center = (hi + lo) / 2
    r_search = float(diam.max() + gap)
    if batch is not None:                                       # neighbors.py:75-78
        off = torch.zeros_like(center)
        off[:, 2] = batch * (center[:, 2].max() - center[:, 2].min() + r_search + 1)
        center = center + off
    d, nb = frnn_grid_points(center, center, k_max + 1, r_search)
    nb, d = nb[:, 1:], d[:, 1:]
    if not squared:
        d = torch.where(d >= 0, d.clamp(min=0).sqrt(), d)
    source = torch.arange(n).repeat_interleave(k_max)
    edge_index = torch.vstack((source, nb.flatten()))
    d = d.flatten()
    r_seg = diam / 2
    keep = d <= r_seg[edge_index].sum(dim=0) + 1.732 * gap
    edge_index, d = edge_index[:, keep], d[keep]
    keep = edge_index[1] != -1
    edge_index, d = edge_index[:, keep], d[keep]
    if trim:
        edge_index, d = to_trimmed(edge_index, d, reduce="min")
    else:
        edge_index, d = coalesce(edge_index, d, reduce="min")
    anchors = scatter_nearest_neighbor(x_points, idx, edge_index, cycles)
    d_nn = (x_points[anchors[0]] - x_points[anchors[1]]).norm(dim=1)
    keep = d_nn <= gap
    return edge_index[:, keep], d_nn[keep], dict(trimmed=edge_index, anchors=anchors, d_nn=d_nn)


# --------------------------------------------------------------------------
# NAG selection / re-indexing
# levels: list of dicts of tensors; 'sub' -> (pointers, points)
# --------------------------------------------------------------------------


def cluster_from_index(index, values):
    """Cluster(index, values, dense=True): CSRData.__init__ / indices_to_pointers
    (src/data/csr.py:85-88, src/utils/sparse.py:23-42), stable order."""
    n = int(index.max()) + 1 if index.numel() else 0
    order = torch.argsort(index, stable=True)
    ptr = torch.cat((torch.zeros(1, dtype=torch.long), torch.bincount(index, minlength=n))).cumsum(0)
    return ptr, values[order]


def cluster_select(pointers, points, idx, update_sub=True):
    """CSRData.select (src/data/csr.py:328-408) + Cluster.select
    (src/data/cluster.py:79-140)."""
    sizes = (pointers[1:] - pointers[:-1])[idx]
    new_ptr = torch.cat((torch.zeros(1, dtype=torch.long), sizes)).cumsum(0)
    start = pointers[:-1][idx]
    val_idx = torch.arange(int(new_ptr[-1])) + (start - new_ptr[:-1]).repeat_interleave(sizes)
    new_points = points[val_idx]
    if not update_sub:
        return (new_ptr, new_points), (None, None)
    inv, perm = consecutive_cluster(new_points)                  # cluster.py:130
    idx_sub = new_points[perm]
    sub_super = torch.empty(inv.numel(), dtype=torch.long)       # to_super_index, cluster.py:67-77
    sub_super[inv] = torch.arange(idx.numel()).repeat_interleave(sizes)
    return (new_ptr, inv), (idx_sub, sub_super)


def data_select(d, idx, update_sub=True, update_super=True):
    """Data.select (src/data/data.py:286-470) on a dict level."""
    n = next(d[k] for k in ("pos", "x", "super_index") if k in d).shape[0]
    out = {}
    idx_edge = None
    if "edge_index" in d:                                         # data.py:360-373
        reindex = torch.full((n,), -1, dtype=torch.long)
        reindex[idx] = torch.arange(idx.numel())
        ei = reindex[d["edge_index"]]
        idx_edge = torch.where((ei != -1).all(dim=0))[0]
        out["edge_index"] = ei[:, idx_edge]
    out_sub = (None, None)
    if "sub" in d:
        out["sub"], out_sub = cluster_select(d["sub"][0], d["sub"][1], idx, update_sub)
    out_super = (None, None)
    if "super_index" in d:
        out["super_index"] = d["super_index"][idx]
        if update_super:                                          # data.py:399-416
            inv, perm = consecutive_cluster(out["super_index"])
            idx_super = out["super_index"][perm]
            out["super_index"] = inv
            out_super = (idx_super, cluster_from_index(inv, torch.arange(idx.numel())))
    ne = d["edge_index"].shape[1] if "edge_index" in d else -1
    for k, v in d.items():
        if k in ("edge_index", "sub", "super_index"):
            continue
        if k.startswith("v_edge_"):
            out[k] = v[idx]
        elif k.startswith("edge_") and v.shape[0] == ne:
            out[k] = v[idx_edge]
        else:
            out[k] = v[idx]
    return out, out_sub, out_super


def nag_select(levels, i_level, idx):
    """NAG.select (src/data/nag.py:306-399)."""
    L = len(levels)
    out = [None] * L
    out[i_level], out_sub, out_super = data_select(levels[i_level], idx)
    for i in range(i_level - 1, -1, -1):
        idx_sub, sub_super = out_sub
        out[i], out_sub, _ = data_select(levels[i], idx_sub, True, False)
        out[i]["super_index"] = sub_super
    for i in range(i_level + 1, L):
        idx_super, super_sub = out_super
        out[i], _, out_super = data_select(levels[i], idx_super, False, True)
        out[i]["sub"] = super_sub
    return out


def vertical_edge_features(child_pos, child_normal, child_logs, parent_pos, parent_normal,
                           parent_logs, super_index):
    """src/transforms/graph.py:1335-1416 with all default keys: per child node
    [centroid_dir (3), sqrt(centroid_dist), |<n_child, n_parent>|, parent - child of
    log_length / log_surface / log_volume / log_size]; a 0/0 direction becomes 0, directions are
    clipped to [-1, 1].  ``*_logs`` = (log_length, log_surface, log_volume, log_size)."""
    idx = super_index
    d = parent_pos[idx] - child_pos
    dist = d.norm(dim=1)
    d = d / dist.view(-1, 1)
    d = torch.where(d.isnan(), torch.zeros_like(d), d).clip(-1, 1)
    cols = [d, dist.sqrt().view(-1, 1),
            (child_normal * parent_normal[idx]).sum(dim=1).abs().view(-1, 1)]
    for pc, cc in zip(parent_logs, child_logs):
        cols.append((pc.view(-1)[idx] - cc.view(-1)).view(-1, 1))
    return torch.cat(cols, dim=1)


# --------------------------------------------------------------------------
# subedges: src/utils/graph.py:99-463
# --------------------------------------------------------------------------


def base_vectors_3d(x):
    """src/utils/geometry.py:42-78."""
    a = x.clone()
    a[a.norm(dim=1) == 0] = torch.tensor([1.0, 0.0, 0.0], dtype=x.dtype)
    a = a / a.norm(dim=1).view(-1, 1)
    b = torch.stack((a[:, 1] - a[:, 2], a[:, 2] - a[:, 0], a[:, 0] - a[:, 1]), dim=1)
    b[b.norm(dim=1) == 0] = torch.tensor([2.0, 1.0, -1.0], dtype=x.dtype)
    b = b / b.norm(dim=1).view(-1, 1)
    return torch.stack((a, b, torch.linalg.cross(a, b)), dim=1)


def subedges(points, index, edge_index, ratio=0.2, k_min=20, cycles=3, margin=0.2,
             halfspace_filter=True, bbox_filter=True, target_pc_flip=True):
    """src/utils/graph.py:99-463 restated EDGE BY EDGE (the reference works on the edge-wise
    expansion of all points; per edge the steps are: anchors -> local frame -> half-space filter
    -> bounding-box filter (each never emptying a side) -> closest-to-anchor first -> top
    max(ratio * size, k_min) on both sides, equal count -> first principal component of each
    side -> target direction flipped against the source's -> both sides ordered along their
    component and paired rank by rank).  Returns (edge_index, ST_pairs, ST_uid).

    Not pinned by construction: the SIGN of the eigenvectors torch.linalg.eigh returns is
    arbitrary, and graph.py:442 decides the target flip from it - so the pairing order of an
    edge may legitimately differ between LAPACK builds; the selected point SETS do not."""
    edge_index = to_trimmed(edge_index)
    E = edge_index.shape[1]
    anchors = scatter_nearest_neighbor(points, index, edge_index, cycles)
    base = base_vectors_3d(points[anchors[1]] - points[anchors[0]])
    order = torch.argsort(index, stable=True)
    n = int(index.max()) + 1
    ptr = torch.cat((torch.zeros(1, dtype=torch.long), torch.bincount(index, minlength=n))).cumsum(0)
    S_out, T_out, U_out = [], [], []
    for e in range(E):
        s, t = int(edge_index[0, e]), int(edge_index[1, e])
        sid, tid = order[ptr[s]:ptr[s + 1]], order[ptr[t]:ptr[t + 1]]
        S = (points[sid] - points[anchors[0, e]]) @ base[e].t()
        T = (points[tid] - points[anchors[1, e]]) @ base[e].t()

        def keep(mask, P, I):
            if not bool(mask.any()):
                return P, I                                   # idx_preserving_mask
            return P[mask], I[mask]
        if halfspace_filter:
            S, sid = keep(S[:, 0] <= margin, S, sid)
            T, tid = keep(T[:, 0] >= -margin, T, tid)
        if bbox_filter:
            lo = torch.max(S[:, 1:].min(0).values, T[:, 1:].min(0).values).clamp(max=-margin)
            hi = torch.min(S[:, 1:].max(0).values, T[:, 1:].max(0).values).clamp(min=margin)
            S, sid = keep(((S[:, 1:] >= lo) & (S[:, 1:] <= hi)).all(1), S, sid)
            T, tid = keep(((T[:, 1:] >= lo) & (T[:, 1:] <= hi)).all(1), T, tid)
        ps = torch.argsort(S[:, 0], descending=True, stable=True)
        pt = torch.argsort(T[:, 0], descending=False, stable=True)
        S, sid, T, tid = S[ps], sid[ps], T[pt], tid[pt]
        ks = min(max(int(S.shape[0] * ratio), k_min), S.shape[0])
        kt = min(max(int(T.shape[0] * ratio), k_min), T.shape[0])
        k = min(ks, kt)
        S, sid, T, tid = S[:k], sid[:k], T[:k], tid[:k]
        zeros = torch.zeros(k, dtype=torch.long)
        s_v = scatter_pca(S, zeros, 1)[1][0, :, -1]
        t_v = scatter_pca(T, zeros, 1)[1][0, :, -1]
        if target_pc_flip:
            t_min = T[(T @ t_v).argmin()]
            st_u = t_min - S.mean(0)
            st_u = st_u / st_u.norm()
            if float(s_v @ t_v) <= float(s_v @ st_u):
                t_v = -t_v
        qs = torch.argsort((S - S.mean(0)) @ s_v, stable=True)
        qt = torch.argsort((T - T.mean(0)) @ t_v, stable=True)
        S_out.append(sid[qs])
        T_out.append(tid[qt])
        U_out.append(torch.full((k,), e, dtype=torch.long))
    return edge_index, torch.vstack((torch.cat(S_out), torch.cat(T_out))), torch.cat(U_out)


# --------------------------------------------------------------------------
# Cluster-object overlaps of the panoptic path (src/data/instance.py) - plain
# loops over numpy arrays, sized for test cases of a few hundred clusters.
# An "instance" here is the tuple (pointers, obj, count, y) of int64 arrays.
# --------------------------------------------------------------------------


def instance_from_dense(cluster, obj, count, y):
    """InstanceData(..., dense=True) (instance.py:70-100): duplicate (cluster, obj) pairs
    merged with their counts summed, pairs grouped by cluster and ordered by obj inside."""
    cluster, obj, count, y = (np.asarray(t, dtype=np.int64) for t in (cluster, obj, count, y))
    merged = {}
    for c, o, n, l in zip(cluster, obj, count, y):
        key = (int(c), int(o))
        prev = merged.get(key, (0, int(l)))
        merged[key] = (prev[0] + int(n), int(l))
    keys = sorted(merged)
    num = max(k[0] for k in keys) + 1
    sizes = np.zeros(num, dtype=np.int64)
    for c, _ in keys:
        sizes[c] += 1
    assert (sizes > 0).all(), "Indices must be dense"          # sparse.py:28
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return (ptr, np.array([k[1] for k in keys], dtype=np.int64),
            np.array([merged[k][0] for k in keys], dtype=np.int64),
            np.array([merged[k][1] for k in keys], dtype=np.int64))


def instance_indices(ptr):
    return np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))


def instance_major(inst, num_classes=None):
    """instance.py:160-236: per cluster the overlap with the largest count (first on ties);
    when at least one void-dominated cluster holds <= 50 % void points, EVERY void-dominated
    cluster takes its largest non-void overlap instead."""
    ptr, obj, count, y = inst
    nc = num_classes if num_classes else int(y.max()) + 1
    n = len(ptr) - 1
    void = (y < 0) | (y >= nc)
    best, best_nv = np.zeros(n, np.int64), np.zeros(n, np.int64)
    tot = np.zeros(n, np.int64)
    for c in range(n):
        lo, hi = ptr[c], ptr[c + 1]
        best[c] = lo + int(np.argmax(count[lo:hi]))
        best_nv[c] = lo + int(np.argmax(count[lo:hi] * ~void[lo:hi]))
        tot[c] = count[lo:hi].sum()
    o, k, l = obj[best].copy(), count[best].copy(), y[best].copy()
    mv = (l < 0) | (l >= nc)
    if not mv.any():
        return o, k, l
    if ((k / tot) > 0.5)[mv].all():
        return o, k, l
    o[mv], l[mv] = obj[best_nv][mv], y[best_nv][mv]
    k[mv] = (count * ~void)[best_nv][mv]
    return o, k, l


def instance_iou_and_size(inst, pair_cropped_count=None):
    """instance.py:268-298 (f32 division like the reference's int / int)."""
    ptr, obj, count, y = inst
    idx = instance_indices(ptr)
    a = np.array([count[idx == i].sum() for i in idx], dtype=np.int64)
    b = np.array([count[obj == o].sum() for o in obj], dtype=np.int64)
    if pair_cropped_count is not None:
        b = b + pair_cropped_count
    iou = count.astype(np.float32) / (a + b - count).astype(np.float32)
    return iou, a, b


def instance_estimate_centroid(inst, cluster_pos, mode="iou"):
    """instance.py:300-352: objects in increasing index order."""
    ptr, obj, count, y = inst
    idx = instance_indices(ptr)
    iou, a, b = instance_iou_and_size(inst)
    if mode == "iou":
        w = iou.astype(np.float64)
    elif mode == "product-iou":
        w = count.astype(np.float64) ** 2 / (a * b)
    elif mode == "overlap":
        w = count.astype(np.float64)
    else:
        raise NotImplementedError
    ids = np.unique(obj)
    pos = np.zeros((len(ids), cluster_pos.shape[1]))
    for j, o in enumerate(ids):
        m = obj == o
        pos[j] = (cluster_pos[idx[m]].astype(np.float64) * w[m, None]).sum(0) / w[m].sum()
    return pos, ids


def instance_graph(inst, edge_index, num_classes=None, smooth_affinity=True):
    """instance.py:354-460: trimmed graph (i < j, sorted, unique, no loops) and per edge
    ``(|i & obj_j| / |i| + |j & obj_i| / |j|) / 2`` or ``obj_i == obj_j``."""
    ptr, obj, count, y = inst
    pairs = sorted({(min(int(s), int(t)), max(int(s), int(t)))
                    for s, t in zip(edge_index[0], edge_index[1]) if s != t})
    e = np.array(pairs, dtype=np.int64).reshape(-1, 2).T
    if e.size == 0:
        return e, np.zeros(0, np.float32)
    major = instance_major(inst, num_classes)[0]
    if not smooth_affinity:
        return e, (major[e[0]] == major[e[1]]).astype(np.float32)
    idx = instance_indices(ptr)
    overlap = {(int(c), int(o)): int(n) for c, o, n in zip(idx, obj, count)}
    size = np.array([count[ptr[c]:ptr[c + 1]].sum() for c in range(len(ptr) - 1)])
    aff = np.zeros(e.shape[1], np.float32)
    for k, (i, j) in enumerate(pairs):
        oij = np.float32(overlap.get((i, int(major[j])), 0))
        oji = np.float32(overlap.get((j, int(major[i])), 0))
        aff[k] = (oij / np.float32(size[i]) + oji / np.float32(size[j])) / np.float32(2)
    return e, aff


def instance_search_void(inst, num_classes):
    """instance.py:462-546."""
    ptr, obj, count, y = inst
    idx = instance_indices(ptr)
    void = (y < 0) | (y >= num_classes)
    n = len(ptr) - 1
    cl_void = np.zeros(n, bool)
    for c in range(n):
        lo, hi = ptr[c], ptr[c + 1]
        cl_void[c] = count[lo:hi][void[lo:hi]].sum() / np.float32(count[lo:hi].sum()) > 0.5
    crop = np.array([count[(obj == o) & cl_void[idx]].sum() for o in obj], dtype=np.int64)
    return cl_void, void | cl_void[idx], crop


# --------------------------------------------------------------------------
# Segment sampling weights (src/transforms/sampling.py:771-798 and 895-921)
# --------------------------------------------------------------------------


def segment_sampling_weights(node_size, y_hist, by_size=False, by_class=False):
    """Uniform + (by_size) cube root of the number of points + (by_class) rarity of the rarest
    class held, each normalised to sum 1 before it is added; f32 like the reference."""
    n = len(node_size)
    w = np.ones(n, np.float32)
    if by_size:
        sw = np.asarray(node_size, np.float32) ** np.float32(0.333)
        w = w + sw / sw.sum()
    if by_class and y_hist is not None:
        h = np.asarray(y_hist)
        sc = np.float32(1) / (np.sqrt(h.sum(0).astype(np.float32)) + np.float32(1))
        sc = sc / sc.sum()
        cw = ((h > 0) * sc[None]).max(1)
        w = w + cw / cw.sum()
    return w / w.sum()
