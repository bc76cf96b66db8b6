"""Parity at BASELINE.json's full sizes (scene S: 15 M points, 428 571 / 178 571
superpoints, 7.03 M edges) through properties that do not need an O(N) CPU pass
of the oracle - plus spot checks of random rows against the oracle."""
import numpy as np
import pytest
import torch

from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu


def test_attention_two_formulations_agree_at_scene_scale(dev):
    """The matrix-pipe kernels and the generic lane-per-output kernels are two
    independent implementations of attention.py:202-315: at N = 428 571,
    E = 7.03 M they must agree (f32 round-off), and random nodes must match the
    float64 oracle evaluated on their edges only."""
    from superpoint_transformer_amd import _lib, ops
    n, e, H, D = 428_571, 7_030_000, 16, 4
    g = torch.Generator(device=dev).manual_seed(0)
    s = torch.randint(0, n, (e,), device=dev, generator=g)
    t = torch.randint(0, n, (e,), device=dev, generator=g)
    ei = torch.stack([s, t])
    qkv = torch.randn(n, 192, device=dev, generator=g)
    ea = torch.randn(e, 32, device=dev, generator=g) * 0.3
    W = [(torch.randn(64, 32, device=dev, generator=g) * 0.1,
          torch.randn(64, device=dev, generator=g) * 0.1) for _ in range(3)]
    gw = torch.randn(n, 64, device=dev, generator=g)
    res = {}
    prev = _lib.lib.spt_attn_use_mfma(2)
    try:
        # 2 = split-bf16 matrix pipe with the packed backward (default), 12 = the same with the
        # per-node backward tiling, 1 = f32 matrix pipe, 0 = VALU
        for mode in (2, 12, 1, 0):
            packed_prev = _lib.lib.spt_attn_bwd_packed(0 if mode == 12 else 1)
            mode_set = 2 if mode == 12 else mode
            _lib.lib.spt_attn_use_mfma(mode_set)
            q = qkv.clone().requires_grad_()
            a = ea.clone().requires_grad_()
            ws = [(w.clone().requires_grad_(), b.clone().requires_grad_()) for w, b in W]
            out = ops.edge_attention(q, ei, a, *ws, num_heads=H, qk_dim=D, scale_a=0.5)
            out.backward(gw)
            res[mode] = (out.detach(), q.grad, a.grad, [w.grad for w, _ in ws], [b.grad for _, b in ws])
            _lib.lib.spt_attn_bwd_packed(packed_prev)
    finally:
        _lib.lib.spt_attn_use_mfma(prev)
    v = res[0]
    for mode in (2, 12, 1):
        m = res[mode]
        torch.testing.assert_close(m[0], v[0], rtol=1e-4, atol=1e-5)
        # d edge_attr = D W: the split-bf16 error is relative to sum |d||w| (~6e-6 of it), not
        # to the (possibly cancelled) result: one entry in 225 M sits at 1.8e-5 absolute
        torch.testing.assert_close(m[2], v[2], rtol=1e-3, atol=1e-5 if mode == 1 else 4e-5)
        # dqkv: k/v columns are f32-atomic sums in every formulation
        torch.testing.assert_close(m[1], v[1], rtol=1e-3, atol=2e-4)
        for a_, b_ in zip(m[3] + m[4], v[3] + v[4]):                           # 7 M-term sums
            assert ((a_ - b_).abs().max() / b_.abs().max().clamp(min=1e-3)).item() < 2e-4
    m = res[2]
    # oracle spot check: 40 random nodes, their outgoing edges only
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:40].to(dev)
    mask = torch.isin(s, pick)
    sub_e = mask.nonzero().view(-1)
    ss, tt = s[sub_e].cpu(), t[sub_e].cpu()
    p = {"qkv.weight": torch.eye(192, dtype=torch.float64), "qkv.bias": None}
    for nm, (w, b) in zip(("k_rpe", "q_rpe", "v_rpe"), W):
        p[nm + ".weight"], p[nm + ".bias"] = w.cpu().double(), b.cpu().double()
    # restrict to the picked nodes: keep global ids (the oracle only needs rows s and t)
    old = O.qk_scale_dg
    deg = torch.bincount(s, minlength=n).cpu()
    O.qk_scale_dg = lambda s_, d_, h_: (0.5 * deg[s_].double() ** -0.5).view(-1, 1, 1)
    try:
        ref = O.self_attention(qkv.cpu().double(), torch.stack([ss, tt]), ea[sub_e].cpu().double(),
                               p, H, D)
    finally:
        O.qk_scale_dg = old
    got = m[0][pick].cpu().double()
    assert ((got - ref[pick.cpu()]).abs() - 1e-4 * ref[pick.cpu()].abs()).max().item() <= 1e-5


def test_knn_at_scene_scale_properties_and_oracle_spot_check(dev):
    """15 M voxelised points, k = 45, r = 2 m (the S3DIS setting)."""
    from superpoint_transformer_amd import neighbors as NB
    from superpoint_transformer_amd.synthetic import make_voxel_cloud
    n, k, r = 15_000_000, 45, 2.0
    pos = make_voxel_cloud(n, voxel=0.03, seed=11, device=dev)
    n = pos.shape[0]
    assert n > 14_000_000
    nb, d = NB.knn_1(pos, k, r)
    ok = nb >= 0
    assert bool((d[ok] < r * r).all()) and bool((d[~ok] == -1).all())
    dd = torch.where(ok, d, torch.full_like(d, float("inf")))
    assert bool((dd[:, 1:] >= dd[:, :-1]).all())                              # ascending rows
    assert bool((nb != torch.arange(n, device=dev).view(-1, 1)).all())        # self excluded
    assert bool((nb < n).all())
    # the stored distance is the f32 distance to the stored index
    rows = torch.randint(0, n, (100_000,), device=dev)
    j = nb[rows]
    diff = pos[rows].unsqueeze(1) - pos[j.clamp(min=0)]
    d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert bool((d2[j >= 0] == d[rows][j >= 0]).all())
    # exhaustive oracle on a few queries against ALL 15 M points
    q = torch.randint(0, n, (24,), generator=torch.Generator().manual_seed(3))
    rd, ri = O.frnn_grid_points(pos[q.to(dev)].cpu(), pos.cpu(), k + 1, r)
    assert torch.equal(nb[q.to(dev)].cpu(), ri[:, 1:])
    assert torch.equal(d[q.to(dev)].cpu(), rd[:, 1:])


@pytest.mark.parametrize("setting", ["dales", "s3dis"])
def test_knn_and_geof_at_full_size_against_the_cpu_twin(setting, dev):
    """The WHOLE neighbour table, bit for bit, against an independent implementation of the
    contract: the OpenMP CPU twin (oracle/cpu/spt_cpu.cpp - heap-based ring search on its own
    grid, pinned on the exhaustive oracle in tests/test_cpu_twin.py), at the DALES settings
    (12 M voxels of 10 cm, k = 25, r = 10 m: configs/datamodule/semantic/dales.yaml) and the
    S3DIS ones (3 cm, k = 45, r = 2 m) - sizes the exhaustive Python oracle cannot reach.
    Geometric features of the same tables: columns that do not depend on an eigenvector basis
    within 1e-4, the twin computing in f64."""
    from bench import PRE_CFG, PRE_GEOM
    from oracle import cpu_twin as T
    from superpoint_transformer_amd import neighbors as NB
    from superpoint_transformer_amd.synthetic import make_voxel_cloud
    scene, n = ("D", 12_000_000) if setting == "dales" else ("S", 6_000_000)
    voxel, k, r = PRE_CFG[scene]
    pos = make_voxel_cloud(n, voxel=voxel, seed=21, device=dev, **PRE_GEOM.get(scene, {}))
    nb, d = NB.knn_1(pos, k, r)
    feats = NB.geometric_features(pos, nb, k_min=1)
    torch.cuda.synchronize()
    ri, rd = T.knn_1(pos.cpu(), k, r)
    assert torch.equal(nb.cpu(), ri)
    assert torch.equal(d.cpu(), rd)
    rf = T.point_geof(pos.cpu(), ri, k_min=1)
    cols = [0, 1, 2, 7, 8, 9, 10]
    err = (feats.cpu()[:, cols] - rf[:, cols]).abs().max().item()
    assert err < 1e-4, err
    # the same tables and features out of ONE call (the kNN kernel sums the moments itself)
    nb1, d1, f1 = NB.knn_1_features(pos, k, r, k_min=1)
    assert torch.equal(nb1, nb) and torch.equal(d1, d)
    del nb1, d1
    same = (f1 == feats).all(1).float().mean().item()
    diff = (f1[:, cols] - feats[:, cols]).abs().max().item()
    print(f"{setting}: fused features equal bit for bit on {100 * same:.4f} % of the rows, "
          f"eigenvalue columns max |diff| {diff:.2e}")
    assert diff <= 1e-6 and same >= 0.99, (same, diff)


def test_graph_norm_and_usn_at_scene_scale(dev):
    from superpoint_transformer_amd import ops
    n, c, ns = 15_000_000, 128, 428_571
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(n, c, device=dev, generator=g) * 3 + 1
    ones, zeros = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    y = ops.graph_norm(x, None, ones, zeros, ones, num_graphs=1)
    mu = y.double().mean(0)
    var = y.double().var(0, unbiased=False)
    assert mu.abs().max().item() < 1e-5 and (var - 1).abs().max().item() < 1e-4  # normalised
    idx = torch.randint(0, ns, (n,), device=dev, generator=g)
    pos = torch.randn(n, 3, device=dev, generator=g) * 10
    pn, diam = ops.unit_sphere_norm(pos, idx, None, ns)
    assert bool((pn.abs() <= 1.0 + 1e-6).all())                               # inside the unit box
    rows = torch.randint(0, ns, (50,), generator=torch.Generator().manual_seed(4))
    for s_ in rows.tolist():                                                   # oracle on whole segments
        mem = (idx == s_).nonzero().view(-1)
        if mem.numel() == 0:
            continue
        ro, rd = O.unit_sphere_norm(pos[mem].cpu().double(), torch.zeros(mem.numel(), dtype=torch.long))
        assert diam[s_].item() == rd.float().item()
        assert (pn[mem].cpu().double() - ro).abs().max().item() < 1e-5


def _instance_properties(dev, n_seg, n_obj, n_edges, seed=0, spot=300):
    """Size-independent properties of the panoptic targets (src/data/instance.py) plus a spot
    check of random segments / edges against the loop oracle on THEIR overlaps only."""
    from superpoint_transformer_amd.instance import InstanceData
    nc = 13
    g = torch.Generator().manual_seed(seed)
    # every segment overlaps 1..3 objects; object labels pure, one object in eight void
    k = torch.randint(1, 4, (n_seg,), generator=g)
    cl = torch.arange(n_seg).repeat_interleave(k)
    ob = torch.randint(0, n_obj, (cl.numel(),), generator=g)
    cnt = torch.randint(1, 5000, (cl.numel(),), generator=g)
    obj_y = torch.randint(0, nc, (n_obj,), generator=g)
    obj_y[torch.arange(n_obj) % 8 == 0] = nc
    y = obj_y[ob]
    d = InstanceData(cl.to(dev), ob.to(dev), cnt.to(dev), y.to(dev), dense=True)
    idx = d.indices
    assert d.num_clusters == n_seg and bool((d.sizes >= 1).all())
    # pairs sorted by (segment, object), unique, counts conserved
    key = idx * n_obj + d.obj
    assert bool((key[1:] > key[:-1]).all())
    assert int(d.count.sum()) == int(cnt.sum())
    total = torch.zeros(n_seg, dtype=torch.long, device=dev).index_add_(0, idx, d.count)

    obj_m, cnt_m, y_m = d.major(nc)
    assert bool((cnt_m <= total).all()) and bool((cnt_m >= 0).all())
    # the major object is one of the segment's objects and its count is the stored one
    hit = (d.obj == obj_m[idx]) & (d.count == cnt_m[idx])
    has = torch.zeros(n_seg, dtype=torch.bool, device=dev)
    has[idx[hit]] = True
    void_only = (cnt_m == 0)                       # segments made of void objects only
    assert bool((has | void_only).all())
    assert bool(((y_m >= 0) & (y_m < nc) | void_only | (cnt_m * 2 > total)).all())

    iou, a_size, b_size = d.iou_and_size()
    assert bool((a_size == total[idx]).all())
    assert bool((iou > 0).all()) and bool((iou <= 1).all())
    assert bool((b_size >= d.count).all())

    e = torch.randint(0, n_seg, (2, n_edges), generator=g).to(dev)
    ei, aff = d.instance_graph(e, nc)
    assert bool((ei[0] < ei[1]).all())
    ek = ei[0] * n_seg + ei[1]
    assert bool((ek[1:] > ek[:-1]).all())
    assert bool((aff >= 0).all()) and bool((aff <= 1).all())
    _, hard = d.instance_graph(e, nc, smooth_affinity=False)
    same = obj_m[ei[0]] == obj_m[ei[1]]
    assert torch.equal(hard, same.float())
    # different target objects and no shared object at all -> affinity 0
    assert bool((aff[hard == 1] > 0).all() | void_only.any())

    cm, pm, crop = d.search_void(nc)
    void_pair = (d.y < 0) | (d.y >= nc)
    void_cnt = torch.zeros(n_seg, dtype=torch.long, device=dev).index_add_(0, idx, d.count * void_pair)
    assert torch.equal(cm, void_cnt * 2 > total)
    assert torch.equal(pm, void_pair | cm[idx])
    r, keep = d.remove_void(nc)
    assert r.num_clusters == int(keep.sum()) and torch.equal(keep, ~cm)
    assert int(r.count.sum()) == int(d.count[~pm].sum())

    # spot check against the loop oracle on a random subset of segments (their pairs only)
    pick = torch.randperm(n_seg, generator=g)[:spot].sort().values
    sub = d.select(pick.to(dev))
    ref = tuple(t.cpu().numpy() for t in (sub.pointers, sub.obj, sub.count, sub.y))
    for a, b in zip(sub.major(nc), O.instance_major(ref, nc)):
        assert np.array_equal(a.cpu().numpy(), b)
    assert np.array_equal(obj_m[pick.to(dev)].cpu().numpy(), O.instance_major(ref, nc)[0])
    es = torch.randint(0, spot, (2, 4 * spot), generator=g)
    e_sub, aff_sub = sub.instance_graph(es.to(dev), nc)
    e_ref, aff_ref = O.instance_graph(ref, es.numpy(), nc)
    assert np.array_equal(e_sub.cpu().numpy(), e_ref)
    np.testing.assert_allclose(aff_sub.cpu().numpy(), aff_ref, rtol=1e-6)


def test_panoptic_targets_at_scene_scale(dev):
    """428 571 segments (scene S), ~860 k overlaps with 60 000 objects, 3 M candidate edges."""
    _instance_properties(dev, 428_571, 60_000, 3_000_000)
