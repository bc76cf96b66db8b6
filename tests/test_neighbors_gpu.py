"""GPU parity: grid kNN and point geometric features.

kNN indices are BIT-EXACT against the oracle's exhaustive search (same f32
distance expression, ties by ascending index), squared distances bit-exact;
the golden fixture is the reference's own knn_brute_force output.  Geometric
features: |err| <= 1e-4 vs the float64 oracle / the fixture produced by the
reference's _geometric_features_torch, away from degenerate spectra."""
import numpy as np
import pytest
import torch

from conftest import demo_nag, load_golden, t64, tl
from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu


def _clouds():
    g = torch.Generator().manual_seed(77)
    planar = torch.rand(6000, 3, generator=g) * torch.tensor([8.0, 8.0, 0.02])
    lines = torch.rand(2000, 3, generator=g) * torch.tensor([10.0, 0.02, 0.02]) + 3
    blob = torch.randn(3000, 3, generator=g) * 0.5 + torch.tensor([4.0, 4.0, 3.0])
    far = torch.tensor([[50.0, 50.0, 50.0], [50.0, 50.0, 50.001], [-30.0, 0.0, 0.0]])
    mixed = torch.cat([planar, lines, blob, far])
    # voxel-like lattice: massive exact distance ties
    ax = torch.arange(16).float() * 0.25
    lattice = torch.stack(torch.meshgrid(ax, ax, ax[:6], indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[torch.randperm(lattice.shape[0], generator=g)]
    return {"mixed": mixed, "lattice": lattice, "tiny": torch.rand(5, 3, generator=g)}


@pytest.mark.parametrize("name,K,r", [("mixed", 46, 2.0), ("mixed", 11, 0.3), ("mixed", 64, 50.0),
                                      ("lattice", 26, 0.6), ("lattice", 46, 10.0),
                                      ("tiny", 8, 1.0), ("tiny", 1, 0.01)])
@pytest.mark.parametrize("cell", [None, 0.11, 3.7])
def test_grid_knn_is_bit_exact(name, K, r, cell, dev):
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()[name]
    dist, idx = NB.frnn_grid_points(xyz.to(dev), xyz.to(dev), K, r, cell_size=cell)
    rd, ri = O.frnn_grid_points(xyz, xyz, K, r)
    assert torch.equal(idx.cpu(), ri)
    assert torch.equal(dist.cpu(), rd)          # same f32 expression: exact


@pytest.mark.parametrize("name,K,r", [("mixed", 46, 2.0), ("mixed", 11, 0.3), ("mixed", 64, 50.0),
                                      ("lattice", 26, 0.6), ("lattice", 46, 10.0), ("lattice", 7, 0.25),
                                      ("tiny", 8, 1.0), ("tiny", 1, 0.01)])
@pytest.mark.parametrize("cell", [None, 0.11, 0.6, 3.7])
@pytest.mark.parametrize("inclusive", [False, True])
def test_self_search_cell_path_is_bit_exact(name, K, r, cell, inclusive, dev):
    """Self-search (query IS search: the preprocessing call) runs the shared-candidate-stream
    kernel + the wave-per-query kernel for its leftovers; both must reproduce the oracle, at
    any cell size (cells far too fine / far too coarse push queries to the leftover path)."""
    from superpoint_transformer_amd import _lib
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()[name]
    p = xyz.to(dev)
    rd, ri = O.frnn_grid_points(xyz, xyz, K, r, strict=not inclusive)
    for path in (1, 0):
        prev = _lib.lib.spt_knn_use_cell_path(path)
        try:
            dist, idx = NB.frnn_grid_points(p, p, K, r, cell_size=cell, inclusive=inclusive)
        finally:
            _lib.lib.spt_knn_use_cell_path(prev)
        assert torch.equal(idx.cpu(), ri), f"path {path}"
        assert torch.equal(dist.cpu(), rd), f"path {path}"


@pytest.mark.parametrize("name,K,r", [("mixed", 101, 2.0), ("mixed", 65, 0.4), ("mixed", 130, 60.0),
                                      ("lattice", 200, 1.1), ("tiny", 70, 2.0)])
@pytest.mark.parametrize("squared", [True, False])
def test_more_than_64_neighbours_by_continuation(name, K, r, squared, dev):
    """K > 64: the first 64 come from the usual search, the rest from continuation searches
    ("strictly after the last one found" in (d2, index) order) - lattice ties straddle the
    cut, lists shorter than 64 must stay -1 padded, two-set search included."""
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()[name]
    p = xyz.to(dev)
    rd, ri = O.frnn_grid_points(xyz, xyz, K, r)
    if not squared:
        rd = torch.where(ri >= 0, rd.clamp(min=0).sqrt(), rd)
    dist, idx = NB.frnn_grid_points(p, p, K, r, squared=squared)
    assert torch.equal(idx.cpu(), ri)
    if squared:
        assert torch.equal(dist.cpu(), rd)
    else:
        torch.testing.assert_close(dist.cpu(), rd, rtol=1e-6, atol=0)      # device sqrt: 1 ulp
    q = xyz[::7] + 0.01
    rd, ri = O.frnn_grid_points(q, xyz, K, r)
    dist, idx = NB.frnn_grid_points(q.to(dev), p, K, r)
    assert torch.equal(idx.cpu(), ri)
    assert torch.equal(dist.cpu(), rd)


def test_self_search_paths_agree_on_a_voxel_cloud(dev):
    """2 M-point voxelised surfaces at the S3DIS settings: the two self-search kernels
    agree bit for bit (indices and squared distances) on every row."""
    from superpoint_transformer_amd import _lib
    from superpoint_transformer_amd import neighbors as NB
    from superpoint_transformer_amd.synthetic import make_voxel_cloud
    pos = make_voxel_cloud(2_000_000, voxel=0.03, seed=11, device=dev)
    out = []
    for path in (1, 0):
        prev = _lib.lib.spt_knn_use_cell_path(path)
        try:
            out.append(NB.frnn_grid_points(pos, pos, 46, 2.0))
        finally:
            _lib.lib.spt_knn_use_cell_path(prev)
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][0], out[1][0])


def test_grid_knn_two_sets_and_euclidean_and_inclusive(dev):
    from superpoint_transformer_amd import neighbors as NB
    g = torch.Generator().manual_seed(5)
    s = torch.rand(4000, 3, generator=g) * 5
    q = torch.rand(700, 3, generator=g) * 7 - 1         # some queries outside the search box
    dist, idx = NB.frnn_grid_points(q.to(dev), s.to(dev), 20, 0.8, squared=False)
    rd, ri = O.frnn_grid_points(q, s, 20, 0.8)
    assert torch.equal(idx.cpu(), ri)
    ok = ri >= 0
    torch.testing.assert_close(dist.cpu()[ok], rd[ok].sqrt(), rtol=1e-6, atol=0)  # device sqrtf: 1 ulp
    assert (dist.cpu()[~ok] == -1).all()
    # inclusive radius on a lattice where many points sit exactly at r
    ax = torch.arange(8).float()
    lat = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    d1, i1 = NB.frnn_grid_points(lat.to(dev), lat.to(dev), 7, 1.0, inclusive=True)
    rd1, ri1 = O.frnn_grid_points(lat, lat, 7, 1.0, strict=False)
    assert torch.equal(i1.cpu(), ri1) and torch.equal(d1.cpu(), rd1)
    d0, i0 = NB.frnn_grid_points(lat.to(dev), lat.to(dev), 7, 1.0, inclusive=False)
    assert (i0[:, 1:] == -1).all()                      # strict: only the point itself


def test_grid_probe_kernels_match_their_torch_expressions(dev):
    """The host-side grid description probes the cloud with two small kernels: the points of
    hashed coarse cells (a spatially coherent subsample) and the number of non-empty cells at a
    candidate cell size.  Both against the torch expressions they replaced."""
    import ctypes
    from superpoint_transformer_amd import _lib
    from superpoint_transformer_amd.ops import _workspace
    g = torch.Generator().manual_seed(5)
    xyz = (torch.rand(300_000, 3, generator=g) * torch.tensor([40.0, 25.0, 3.0]) - 7.0).to(dev)
    lo = xyz.min(0).values.tolist()
    o3 = (ctypes.c_float * 3)(*lo)
    r, thresh = 2.0, 9000
    buf = torch.empty_like(xyz)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    st = _lib.lib.spt_knn_subsample_f32(_lib.ptr(xyz), xyz.shape[0], ctypes.cast(o3, ctypes.c_void_p), r,
                                        thresh, _lib.ptr(buf), _lib.ptr(cnt), _lib.stream_ptr(dev))
    _lib.check(st, "spt_knn_subsample_f32")
    lo_t = torch.tensor(lo, device=dev)
    cc = ((xyz - lo_t) / r).floor().long()
    key = (cc[:, 0] * 73856093) ^ (cc[:, 1] * 19349663) ^ (cc[:, 2] * 83492791)
    keep = (key & 0xFFFF) < thresh
    m = int(cnt.item())
    assert m == int(keep.sum()) and 0 < m < xyz.shape[0]
    got = buf[:m].cpu()
    ref = xyz[keep].cpu()
    srt = lambda t: t[torch.from_numpy(np.lexsort((t[:, 2].numpy(), t[:, 1].numpy(), t[:, 0].numpy())))]
    assert torch.equal(srt(got), srt(ref))                    # the same points, in some order
    for sz in (0.5, 0.07):
        ext = (xyz.max(0).values - xyz.min(0).values).tolist()
        d1 = [int(e / sz) + 1 for e in ext]
        d3 = (ctypes.c_int32 * 3)(*d1)
        ids = torch.empty(xyz.shape[0], dtype=torch.int64, device=dev)
        st = _lib.lib.spt_grid_cell_ids_f32(_lib.ptr(xyz), xyz.shape[0], sz, ctypes.cast(o3, ctypes.c_void_p),
                                            ctypes.cast(d3, ctypes.c_void_p), _lib.ptr(ids), _lib.stream_ptr(dev))
        _lib.check(st, "spt_grid_cell_ids_f32")
        inv = (torch.ones((), device=dev) / torch.tensor(sz, device=dev))      # 1.0f / sz in f32, as the kernel
        c = ((xyz - lo_t) * inv).floor().long()
        for q in range(3):
            c[:, q].clamp_(0, d1[q] - 1)
        assert torch.equal(ids, (c[:, 2] * d1[1] + c[:, 1]) * d1[0] + c[:, 0])
        count = torch.empty(1, dtype=torch.int64, device=dev)
        ncells = d1[0] * d1[1] * d1[2]
        ws = _workspace(_lib.lib.spt_grid_count_cells_workspace_bytes(ncells), dev)
        st = _lib.lib.spt_grid_count_cells_f32(_lib.ptr(xyz), xyz.shape[0], sz, ctypes.cast(o3, ctypes.c_void_p),
                                               ctypes.cast(d3, ctypes.c_void_p), _lib.ptr(count), _lib.ptr(ws),
                                               ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "spt_grid_count_cells_f32")
        assert int(count.item()) == int(torch.unique(ids).numel())


def test_knn_1_matches_reference_brute_force_fixture(dev):
    """Golden vector = output of the reference's own knn_brute_force."""
    from superpoint_transformer_amd import neighbors as NB
    g = load_golden("knn_brute_force.npz")
    xyz = torch.from_numpy(g["xyz"])
    k, r = int(g["k"]), float(g["r_max"])
    nb, d = NB.knn_1(xyz.to(dev), k, r)
    assert torch.equal(nb.cpu(), tl(g["neighbors"]))
    ref_d = torch.from_numpy(g["distances"])
    ok = nb.cpu() >= 0
    torch.testing.assert_close(d.cpu()[ok].sqrt(), ref_d[ok], rtol=1e-5, atol=1e-6)


def test_knn_1_on_demo_room_properties(dev):
    """Real S3DIS room at the reference's settings (k=45, r=2 m): properties that
    hold at full size - sorted rows, self excluded, symmetric-consistent radius,
    and an exhaustive oracle check on a random sample of queries."""
    from superpoint_transformer_amd import neighbors as NB
    pos = torch.from_numpy(demo_nag()[0]["pos"]).float()
    nb, d = NB.knn_1(pos.to(dev), 45, 2.0)
    nb, d = nb.cpu(), d.cpu()
    ok = nb >= 0
    assert (d[ok] < 4.0).all() and (d[~ok] == -1).all()
    dd = torch.where(ok, d, torch.full_like(d, float("inf")))
    assert (dd[:, 1:] >= dd[:, :-1]).all()
    assert (nb != torch.arange(pos.shape[0]).view(-1, 1)).all()
    sample = torch.randperm(pos.shape[0], generator=torch.Generator().manual_seed(0))[:300]
    rd, ri = O.frnn_grid_points(pos[sample], pos, 46, 2.0)
    assert torch.equal(nb[sample], ri[:, 1:])
    assert torch.equal(d[sample], rd[:, 1:])


def _spectrum_ok(xyz, nn, gap=1e-3):
    n = xyz.shape[0]
    full = torch.cat((torch.arange(n).view(-1, 1), nn), dim=1)
    ptr, val, _ = O.neighbors_dense_to_csr(full)
    gidx = torch.repeat_interleave(torch.arange(n), ptr[1:] - ptr[:-1])
    ev, _ = O.scatter_pca(xyz.double()[val], gidx, n)
    l = ev.flip(1)
    return ((l[:, 0] - l[:, 1]) > gap * l[:, 0]) & ((l[:, 1] - l[:, 2]) > gap * l[:, 0])


def test_geometric_features_match_reference_fixture(dev):
    from superpoint_transformer_amd import neighbors as NB
    g = load_golden("geometric_features.npz")
    xyz = torch.from_numpy(g["xyz"]).float()
    nn = tl(g["nn"])
    f = NB.geometric_features(xyz.to(dev), nn.to(dev), k_min=int(g["k_min"])).cpu().double()
    ref = t64(g["feats"])
    good = _spectrum_ok(xyz, nn)
    scal = [0, 1, 2, 7, 8, 10]
    assert (f[:, scal] - ref[:, scal]).abs().max().item() <= 1e-4
    # volume = cbrt(l1 l2 l3 + 1e-9) has slope 3e5 at 0: on rank-deficient
    # neighbourhoods (<= 3 points are always coplanar, l3 = 0 up to rounding) an
    # eigenvalue rounding of 1e-16 moves it by 3e-4 in ANY implementation
    generic = ((nn >= 0).sum(1) + 1) >= 4
    assert (f[generic][:, 9] - ref[generic][:, 9]).abs().max().item() <= 1e-4
    assert (f[~generic][:, 9] - ref[~generic][:, 9]).abs().max().item() <= 1e-3
    # verticality and the normal depend on eigenVECTORS: unique only away from
    # repeated eigenvalues
    assert (f[good][:, 3:7] - ref[good][:, 3:7]).abs().max().item() <= 1e-4
    assert good.float().mean() > 0.9


@pytest.mark.parametrize("k_min", [1, 5])
def test_geometric_features_vs_oracle_with_partial_neighbourhoods(k_min, dev):
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()["mixed"]
    _, idx = O.frnn_grid_points(xyz, xyz, 21, 0.35)
    nn = idx[:, 1:]
    f = NB.geometric_features(xyz.to(dev), nn.to(dev), k_min=k_min).cpu().double()
    ref = O.geometric_features(xyz.double(), nn, k_min=k_min)
    scal = [0, 1, 2, 7, 8, 9, 10]
    assert (f[:, scal] - ref[:, scal]).abs().max().item() <= 1e-4
    good = _spectrum_ok(xyz, nn) & ((nn >= 0).sum(1) + 1 >= max(k_min, 4))
    assert (f[good][:, 3:7] - ref[good][:, 3:7]).abs().max().item() <= 1e-4
    small = ((nn >= 0).sum(1) + 1) < k_min
    assert (f[small] == 0).all()
    # CSR entry point (pgeof layout: the point itself is in its own list)
    n = xyz.shape[0]
    full = torch.cat((torch.arange(n).view(-1, 1), nn), dim=1)
    ptr, val, _ = O.neighbors_dense_to_csr(full)
    fc = NB.geometric_features_csr(xyz.to(dev), val.to(dev), ptr.to(dev), k_min=k_min,
                                   add_self=False, raw=False).cpu().double()
    assert torch.equal(fc, f)


def _same_features(f, f0, xyz, nn, label):
    """Features out of the kNN kernel vs geometric_features on its stored rows: the same f64
    moments about the point in another summation order.  Columns built from eigenVALUES within
    1e-6; rows whose spectrum is well separated: every column within 1e-5; and the bulk of the
    rows equal bit for bit (printed)."""
    f, f0 = f.cpu(), f0.cpu()
    scal = [0, 1, 2, 7, 8, 9, 10]
    err = (f[:, scal] - f0[:, scal]).abs().max().item() if f.numel() else 0.0
    same = (f == f0).all(1).float().mean().item() if f.numel() else 1.0
    print(f"{label}: rows equal bit for bit {100 * same:.3f} %, max |diff| of the eigenvalue columns {err:.2e}")
    assert err <= 1e-6, (label, err)
    good = _spectrum_ok(xyz, nn)
    if good.any():
        assert (f[good] - f0[good]).abs().max().item() <= 1e-5, label
    assert same >= 0.9, (label, same)


@pytest.mark.parametrize("name,k,r", [("mixed", 45, 2.0), ("mixed", 20, 0.35), ("mixed", 10, 0.3),
                                      ("mixed", 63, 50.0), ("lattice", 25, 0.6),
                                      ("lattice", 45, 10.0), ("lattice", 6, 0.25), ("tiny", 4, 1.0)])
@pytest.mark.parametrize("cell", [None, 0.11, 0.6, 3.7])
@pytest.mark.parametrize("k_min", [1, 5])
def test_knn_1_features_is_knn_1_then_geometric_features(name, k, r, cell, k_min, dev):
    """One call for the reference's KNN -> PointFeatures pair: neighbours and distances are
    knn_1's bit for bit, the features geometric_features' - from the cell kernel (moments summed
    next to the sort), from its leftovers (cells far too fine / too coarse push queries there)
    and from the one-wave-per-query formulation, partial and empty neighbourhoods included."""
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()[name]
    p = xyz.to(dev)
    nb0, d0 = NB.knn_1(p, k, r)
    f0 = NB.geometric_features(p, nb0.contiguous(), k_min=k_min, order=False)
    # knn_1's table is a column slice of the [N, k + 1] search result: read in place (row pitch)
    assert not nb0.is_contiguous() or nb0.shape[0] <= 1
    assert torch.equal(NB.geometric_features(p, nb0, k_min=k_min, order=False), f0)
    small = ((nb0 >= 0).sum(1) + 1) < k_min
    for formulation in (-1, 0):
        nb, d, f = NB.knn_1_features(p, k, r, k_min=k_min, formulation=formulation, cell_size=cell)
        assert torch.equal(nb, nb0) and torch.equal(d, d0), formulation
        assert bool((f[small] == 0).all())
        _same_features(f, f0, xyz, nb0.cpu(), f"{name} k={k} r={r} cell={cell} formulation={formulation}")
    # raw = without the tail of geometric_features (verticality * 2, normals flipped to z >= 0)
    _, _, fr = NB.knn_1_features(p, k, r, k_min=k_min, raw=True, cell_size=cell)
    fr0 = NB.geometric_features(p, nb0.contiguous(), k_min=k_min, raw=True, order=False)
    _same_features(fr, fr0, xyz, nb0.cpu(), f"{name} raw")


def test_knn_1_features_against_the_f64_oracle(dev):
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()["mixed"]
    nb, d, f = NB.knn_1_features(xyz.to(dev), 20, 0.35, k_min=1)
    rd, ri = O.frnn_grid_points(xyz, xyz, 21, 0.35)
    assert torch.equal(nb.cpu(), ri[:, 1:]) and torch.equal(d.cpu(), rd[:, 1:])
    ref = O.geometric_features(xyz.double(), ri[:, 1:], k_min=1)
    f = f.cpu().double()
    scal = [0, 1, 2, 7, 8, 9, 10]
    assert (f[:, scal] - ref[:, scal]).abs().max().item() <= 1e-4
    good = _spectrum_ok(xyz, ri[:, 1:]) & ((ri[:, 1:] >= 0).sum(1) + 1 >= 4)
    assert (f[good][:, 3:7] - ref[good][:, 3:7]).abs().max().item() <= 1e-4


def test_knn_1_features_on_demo_room(dev):
    """Real S3DIS room at the reference's settings (k = 45, r = 2 m), > 200 k points: the cell
    order is remembered for later gather kernels as knn_1 does."""
    from superpoint_transformer_amd import neighbors as NB
    pos = torch.from_numpy(demo_nag()[0]["pos"]).float().to(dev)
    nb0, d0 = NB.knn_1(pos, 45, 2.0)
    f0 = NB.geometric_features(pos, nb0, k_min=5)
    nb, d, f = NB.knn_1_features(pos, 45, 2.0, k_min=5)
    assert torch.equal(nb, nb0) and torch.equal(d, d0)
    _same_features(f, f0, pos.cpu(), nb0.cpu(), "demo room")
    with pytest.raises(ValueError):
        NB.knn_1_features(pos, 64, 2.0)


def test_preprocess_pipeline_on_demo_room_regresses_to_stored_features(dev):
    """The demo room stores the REFERENCE's own PointFeatures output (k=45,
    r=2 m; fp16 on disk).  Recomputing kNN + features here must land on them:
    a loose end-to-end pin of a9-a14 on real data."""
    from superpoint_transformer_amd import neighbors as NB
    lv0 = demo_nag()[0]
    pos = torch.from_numpy(lv0["pos"]).float().to(dev)
    nb, _ = NB.knn_1(pos, 45, 2.0)
    f = NB.geometric_features(pos, nb, k_min=1).cpu()
    for col, key in ((0, "linearity"), (1, "planarity"), (2, "scattering"), (3, "verticality")):
        ref = torch.from_numpy(lv0[key]).float().view(-1)
        err = (f[:, col] - ref).abs()
        # the stored room was voxelised/sub-sampled after feature computation, so
        # neighbourhoods differ slightly: compare distributions, not points
        assert err.median().item() < 0.05, (key, err.median().item())


def test_visiting_order_does_not_change_the_features(dev):
    """geometric_features can visit the points in a spatial order (cache locality of the
    gathers); the features are bit-identical whatever the visiting order."""
    from superpoint_transformer_amd import neighbors as NB
    from superpoint_transformer_amd.synthetic import make_voxel_cloud
    pos = make_voxel_cloud(260_000, voxel=0.03, seed=5, device=dev, extent=(12.0, 12.0, 4.0))
    nb, _ = NB.knn_1(pos, 20, 0.5)
    order = NB.spatial_order(pos)
    assert order.dtype == torch.int32
    assert torch.equal(torch.sort(order.long()).values, torch.arange(pos.shape[0], device=dev))
    assert NB._recall_order(pos) is not None                    # left behind by knn_1 (>= 200 k points)
    a = NB.geometric_features(pos, nb, k_min=1, order=False)    # as stored
    b = NB.geometric_features(pos, nb, k_min=1, order=order)    # explicit permutation
    c = NB.geometric_features(pos, nb, k_min=1)                 # the kNN grid's cell order
    assert torch.equal(a, b) and torch.equal(a, c)
    pos2 = pos.clone()
    assert NB._recall_order(pos2) is None                       # the memo belongs to that tensor


def test_neighbors_dense_to_csr_kernel_matches_reference_rule(dev):
    """a10 (src/utils/neighbors.py:668-684): the product function (count / scan / emit
    kernels) against the oracle restatement and the reference's own fixture."""
    from superpoint_transformer_amd import neighbors as NB
    g = torch.Generator().manual_seed(3)
    nn = torch.randint(0, 5000, (3000, 17), generator=g)
    nn[torch.rand(nn.shape, generator=g) < 0.35] = -1          # holes anywhere in a row
    nn[7] = -1                                                 # an empty row
    nn[11] = torch.arange(17)                                  # a full row
    ptr, val, sizes = NB.neighbors_dense_to_csr(nn.to(dev))
    rp, rv, rs = O.neighbors_dense_to_csr(nn)
    assert torch.equal(ptr.cpu(), rp) and torch.equal(val.cpu(), rv) and torch.equal(sizes.cpu(), rs)
    gold = load_golden("geometric_features.npz")               # written by the reference itself
    full = torch.cat((torch.arange(gold["nn"].shape[0]).view(-1, 1), tl(gold["nn"])), dim=1)
    ptr, val, sizes = NB.neighbors_dense_to_csr(full.to(dev))
    assert torch.equal(ptr.cpu(), tl(gold["nn_ptr"]))
    assert torch.equal(val.cpu(), tl(gold["nn_val"]))
    assert torch.equal(sizes.cpu(), tl(gold["sizes"]))
    e_ptr, e_val, e_sizes = NB.neighbors_dense_to_csr(torch.empty((0, 5), dtype=torch.long, device=dev))
    assert e_ptr.tolist() == [0] and e_val.numel() == 0 and e_sizes.numel() == 0


def test_scatter_pca_entry_matches_reference_fixture(dev):
    """a12 (src/utils/scatter.py:41-125): eigenvalues against the reference's own
    scatter_pca output (ascending, clamped at 0), eigenvectors against the f64 oracle up to
    the sign of each column (eigh's sign is arbitrary), away from repeated eigenvalues; an
    empty group yields (1,1,1) / identity."""
    from superpoint_transformer_amd import segment as SG
    gold = load_golden("geometric_features.npz")
    xyz = torch.from_numpy(gold["xyz"]).float()
    ptr, val = tl(gold["nn_ptr"]), tl(gold["nn_val"])
    n = ptr.numel() - 1
    gidx = torch.repeat_interleave(torch.arange(n), ptr[1:] - ptr[:-1])
    pts = xyz[val]
    # unsorted group index + two trailing empty groups
    perm = torch.randperm(pts.shape[0], generator=torch.Generator().manual_seed(1))
    ev, evec = SG.scatter_pca(pts[perm].to(dev), gidx[perm].to(dev), n + 2)
    ev, evec = ev.cpu().double(), evec.cpu().double()
    ref_ev = t64(gold["eigenval"])
    assert ((ev[:n] - ref_ev).abs() / ref_ev.abs().max(dim=1, keepdim=True).values.clamp(min=1e-6)).max() < 1e-5
    assert (ev[:n, 1:] >= ev[:n, :-1]).all() and (ev >= 0).all()
    assert torch.equal(ev[n:], torch.ones(2, 3, dtype=torch.float64))
    assert torch.equal(evec[n:], torch.eye(3, dtype=torch.float64).expand(2, 3, 3))
    rv, rvec = O.scatter_pca(pts.double(), gidx, n)
    gap_ok = ((rv[:, 1] - rv[:, 0]) > 1e-3 * rv[:, 2]) & ((rv[:, 2] - rv[:, 1]) > 1e-3 * rv[:, 2])
    dots = (evec[:n] * rvec).sum(dim=1).abs()                  # |<column_i, ref column_i>|
    assert (dots[gap_ok] > 1 - 1e-4).all()
    # orthonormal columns everywhere
    eye = torch.eye(3, dtype=torch.float64)
    assert ((evec.transpose(1, 2) @ evec) - eye).abs().max() < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_optimal_neighbourhood_features_match_the_reference(tag, dev):
    """k_step >= 0 (geometry.py:248-287): per point the neighbourhood size of lowest
    eigenentropy, against the reference's own ``geometric_features_torch`` (f64 fixture).
    Points whose two best DISTINCT sizes are closer than 1e-5 in entropy may legitimately pick
    the other one in f32 and are left out (a handful at most)."""
    from superpoint_transformer_amd import neighbors as NB
    g = load_golden("geometric_features_optimal.npz")
    xyz, nn = torch.from_numpy(g["xyz"]).float(), tl(g["nn"])
    k_min, k_step, k_search = (int(v) for v in g[f"{tag}_cfg"])
    f = NB.geometric_features(xyz.to(dev), nn.to(dev), k_min=k_min, k_step=k_step,
                              k_min_search=k_search).cpu().double()
    ref = t64(g[f"{tag}_feats"])
    _, margin = O.geometric_features_optimal(xyz.double(), nn, k_min, k_step, k_search,
                                             return_margin=True)
    safe = margin > 1e-5
    assert safe.float().mean() > 0.98
    scal = [0, 1, 2, 7, 8, 10]
    assert (f[safe][:, scal] - ref[safe][:, scal]).abs().max().item() <= 1e-4
    generic = safe & (((nn >= 0).sum(1) + 1) >= 4)
    assert (f[generic][:, 9] - ref[generic][:, 9]).abs().max().item() <= 1e-4
    vec_ok = safe & (ref[:, 0] > 1e-3) & (ref[:, 1] > 1e-3)          # distinct eigenvalues
    assert (f[vec_ok][:, 3:7] - ref[vec_ok][:, 3:7]).abs().max().item() <= 2e-4
    assert vec_ok.float().mean() > 0.7
    # the pgeof-shaped entry (CSR lists holding the point itself, raw columns)
    if tag == "a":
        from superpoint_transformer_amd.shims import pgeof_shim
        full = torch.cat((torch.arange(xyz.shape[0]).view(-1, 1), nn), dim=1)
        ptr, val, _ = O.neighbors_dense_to_csr(full)
        raw = NB.geometric_features(xyz.to(dev), nn.to(dev), k_min=k_min, k_step=k_step,
                                    k_min_search=k_search, raw=True).cpu().numpy()
        got = pgeof_shim.compute_features_optimal(
            xyz.numpy(), val.numpy().astype("uint32"), ptr.numpy().astype("uint32"),
            k_min, k_step, k_search)
        assert (got == raw).all()
    with pytest.raises(ValueError):
        NB.geometric_features(xyz.to(dev), nn.to(dev), k_min=1, k_step=2, k_min_search=40)


def test_oversampled_partial_neighbourhoods(dev):
    """neighbors.py:420-488: every missing entry of a partial neighbourhood becomes a copy of
    one of its found neighbours (with its distance); found entries and empty rows untouched."""
    from superpoint_transformer_amd import neighbors as NB
    xyz = _clouds()["mixed"].to(dev)
    k, r = 20, 0.35
    nb0, d0 = NB.knn_1(xyz, k, r)
    nb, d = NB.knn_1(xyz, k, r, oversample=True)
    found = (nb0 >= 0).sum(1)
    partial = (found > 0) & (found < k)
    assert int(partial.sum()) > 100 and int((found == 0).sum()) >= 0
    keep = nb0 >= 0
    assert torch.equal(nb[keep], nb0[keep]) and torch.equal(d[keep], d0[keep])
    assert torch.equal(nb[found == 0], nb0[found == 0])               # stays empty
    assert bool((nb[found > 0] >= 0).all())                           # nothing missing any more
    rows = torch.where(partial)[0]
    filled = ~keep[rows]
    # each filled entry (neighbour, distance) is one of the row's found pairs
    same = (nb[rows].unsqueeze(2) == nb0[rows].unsqueeze(1)) & \
           (d[rows].unsqueeze(2) == d0[rows].unsqueeze(1)) & keep[rows].unsqueeze(1)
    assert bool(same.any(dim=2)[filled].all())
    # and the draws are spread: some row with >= 4 found neighbours and >= 8 holes uses >= 2 of them
    rich = rows[(found[rows] >= 4) & (found[rows] <= k - 8)]
    if rich.numel():
        distinct = torch.tensor([nb[i][~keep[i]].unique().numel() for i in rich[:50].tolist()])
        assert int(distinct.max()) >= 2
