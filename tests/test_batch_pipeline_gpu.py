"""GPU: the per-batch on-device transform chain of the training config
(configs/datamodule/semantic/default.yaml:206-290) on the NAG mirror, feeding the
SPT model: NodeSize -> SampleSubNodes -> SampleSegments -> OnTheFlyHorizontalEdgeFeatures
(+ self loops) -> SampleEdges -> forward / backward.  Each transform is checked against
the oracle given the same random draw; the chain by hierarchy invariants."""
import pytest
import torch

from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu


def as_levels(nag):
    out = []
    for d in nag:
        lv = {}
        for k, v in d:
            lv[k] = (v.pointers.cpu(), v.points.cpu()) if k == "sub" else v.cpu()
        out.append(lv)
    return out


def check_hierarchy(nag):
    n = nag.num_points
    for i in range(nag.num_levels):
        d = nag[i]
        if i + 1 < nag.num_levels:
            assert d.super_index.shape[0] == n[i]
            assert int(d.super_index.max()) + 1 == n[i + 1]
            assert torch.equal(nag[i + 1].sub.to_super_index(), d.super_index)
        if d.has_edges:
            assert int(d.edge_index.max()) < n[i] and int(d.edge_index.min()) >= 0
            assert d.edge_attr.shape[0] == d.edge_index.shape[1]
        for k, v in d:
            if k in ("pos", "normal", "node_size", "log_size"):
                assert v.shape[0] == n[i], (i, k)


def test_sample_sub_nodes_equals_select_of_its_draw(dev):
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd.transforms import NodeSize, SampleSubNodes
    nag = NodeSize(0)(make_raw_nag("R", device=dev))
    ref_levels = as_levels(nag)
    idx = nag.get_sampling(high=1, low=0, n_max=32, n_min=8, seed=5)
    out = SampleSubNodes(high=1, low=0, n_max=32, n_min=8, seed=5)(nag)
    ref = O.nag_select(ref_levels, 0, idx.cpu())
    for d, r in zip(out, ref):
        for k, v in r.items():
            if k == "sub":
                assert torch.equal(d.sub.pointers.cpu(), v[0])
            else:
                assert torch.equal(d[k].cpu(), v), k
    check_hierarchy(out)
    sizes = torch.bincount(out[0].super_index, minlength=out.num_points[1])
    full = nag[1].node_size
    expect = (32 * torch.tanh(full / 32)).floor().long().clamp(min=8).clamp(max=full)
    assert torch.equal(sizes, expect)          # node_size keeps the pre-sampling count


def test_full_chain_feeds_the_model(dev):
    from superpoint_transformer_amd.hotpath import SPTSegmenter, spt64_config
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd import transforms as T
    torch.manual_seed(0)
    nag = make_raw_nag("R", device=dev)
    n_before = nag.num_points
    chain = [T.NodeSize(0), T.SampleSubNodes(1, 0, n_max=32, n_min=8, seed=1),
             T.SampleSegments(ratio=0.2), T.OnTheFlyHorizontalEdgeFeatures(),
             T.SampleEdges(levels=(1, 2), n_min=4, n_max=16, seed=2)]
    for t in chain:
        nag = t(nag)
    check_hierarchy(nag)
    n = nag.num_points
    assert n[1] == n_before[1] - int(n_before[1] * 0.2) or n[1] <= n_before[1]
    assert nag[1].edge_attr.shape[1] == 18
    deg = torch.bincount(nag[1].edge_index[0], minlength=n[1])
    assert int(deg.max()) <= 16
    model = SPTSegmenter(**spt64_config(nag[0].x.shape[1], 18)).to(dev)
    logits = model(nag)
    assert logits[0].shape[0] == n[1] and torch.isfinite(logits[0]).all()
    sum(l.square().mean() for l in logits).backward()
    g = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(g) > 50 and all(torch.isfinite(x).all() for x in g)


def test_from_nag_list_round_trips_through_select(dev):
    from superpoint_transformer_amd.data import NAG
    from superpoint_transformer_amd.synthetic import make_raw_nag
    a = make_raw_nag("R", seed=1, device=dev, sizes=(3000, 120, 30, 600, 120, 1))
    b = make_raw_nag("R", seed=2, device=dev, sizes=(2000, 90, 20, 400, 90, 1))
    both = NAG.from_nag_list([a, b])
    check_hierarchy(both)
    assert both.num_points == [5000, 210, 50]
    # keeping the top-level nodes of item 1 gives item b back
    top = torch.where(both[2].batch == 1)[0]
    back = both.select(2, top)
    assert back.num_points == b.num_points
    for i in range(3):
        assert torch.equal(back[i].pos, b[i].pos)
        if i < 2:
            assert torch.equal(back[i].super_index, b[i].super_index)
        if i > 0:
            assert torch.equal(back[i].edge_index, b[i].edge_index)
            assert torch.equal(back[i].edge_attr, b[i].edge_attr)
            assert torch.equal(back[i].sub.pointers, b[i].sub.pointers)


def test_batch_preparation_at_train_batch_scale(dev):
    """Scene T (1.2 M points, the S3DIS train-batch shape): wall time of the chain."""
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd import transforms as T
    nag0 = make_raw_nag("T", device=dev)
    chain = [T.NodeSize(0), T.SampleSubNodes(1, 0, n_max=32, n_min=8, seed=1),
             T.SampleSegments(ratio=0.2), T.OnTheFlyHorizontalEdgeFeatures(),
             T.SampleEdges(levels=(1, 2), n_min=4, n_max=16, seed=2)]
    for rep in range(2):
        nag = nag0.clone()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for t in chain:
            nag = t(nag)
        ev[1].record()
        torch.cuda.synchronize()
    print(f"on-device batch preparation, scene T: {ev[0].elapsed_time(ev[1]):.1f} ms "
          f"{nag0.num_points} -> {nag.num_points}")
    check_hierarchy(nag)


def test_radius_subgraphs_match_the_brute_force_neighbourhoods(dev):
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd import transforms as T
    nag = make_raw_nag("R", device=dev)
    pos1 = nag[1].pos.cpu()
    seeds = torch.tensor([5, 700, 311])
    for cyl in (False, True):
        t = T.SampleRadiusSubgraphs(r=6.0, k_max=10000, i_level=1, k=3, disjoint=False,
                                    cylindrical=cyl, idx_seed=seeds)
        out = t(nag)
        w = torch.tensor([[1.0, 1.0, 0.0 if cyl else 1.0]])
        nb, _ = O.knn_brute_force(pos1 * w, pos1[seeds] * w, pos1.shape[0], r_max=6.0)
        expect = nb[nb >= 0].unique()                       # sampling.py:1228
        assert out.num_points[1] == expect.numel() and expect.numel() > 20
        assert torch.equal(out[1].pos.cpu(), pos1[expect])
        check_hierarchy(out)
    # k_max caps every ball at its nearest nodes
    t = T.SampleRadiusSubgraphs(r=6.0, k_max=15, i_level=1, k=1, idx_seed=seeds[:1])
    out = t(nag)
    nb, _ = O.knn_brute_force(pos1, pos1[seeds[:1]], 15, r_max=6.0)
    assert torch.equal(out[1].pos.cpu(), pos1[nb[nb >= 0].unique()])


def test_disjoint_radius_subgraphs_become_batch_items(dev):
    from superpoint_transformer_amd.hotpath import SPTSegmenter, spt64_config
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd import transforms as T
    torch.manual_seed(3)
    nag = T.NodeSize(0)(make_raw_nag("R", device=dev))
    out = T.SampleRadiusSubgraphs(r=7.0, k_max=10000, i_level=1, k=4, disjoint=True)(nag)
    check_hierarchy(out)
    for i in range(3):
        b = out[i].batch
        assert int(b.max()) == 3 and bool((b[1:] >= b[:-1]).all())    # 4 contiguous items
    assert torch.equal(out[1].batch[out[0].super_index], out[0].batch)
    out = T.OnTheFlyHorizontalEdgeFeatures()(out)
    out.num_clouds = 4
    model = SPTSegmenter(**spt64_config(out[0].x.shape[1], 18)).to(dev)
    logits = model(out)
    assert logits[0].shape[0] == out.num_points[1] and torch.isfinite(logits[0]).all()


def test_restrict_size_caps_nodes_and_edges_of_the_upper_levels(dev):
    """NAGRestrictSize (sampling.py:1346-1423) with the training pipeline's ``level='1+'``: the
    surplus level-1 nodes leave through ``NAG.select`` (the same draw = the same result), the
    surplus edges are dropped with their attributes; level 0 is not capped itself."""
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd.transforms import NAGRestrictSize, NodeSize
    nag = NodeSize(0)(make_raw_nag("R", device=dev))
    n = nag.num_points
    cap_n, cap_e = n[1] // 2, 3000
    torch.manual_seed(11)
    idx = torch.multinomial(torch.ones(n[1], device=dev), cap_n, replacement=False)
    expect = nag.select(1, idx)
    torch.manual_seed(11)
    out = NAGRestrictSize(level="1+", num_nodes=cap_n, num_edges=cap_e)(nag)
    check_hierarchy(out)
    assert out.num_points[1] == cap_n and out.num_points[0] == expect.num_points[0] < n[0]
    assert torch.equal(out[0].pos, expect[0].pos) and torch.equal(out[1].pos, expect[1].pos)
    assert out.num_points[2] <= cap_n                      # level 2 was under the cap or cut to it
    for i in (1, 2):
        assert out[i].num_edges <= cap_e
        assert out[i].edge_attr.shape[0] == out[i].num_edges
    same = NAGRestrictSize(level="1+", num_nodes=10 ** 9, num_edges=10 ** 9)(out)
    assert same.num_points == out.num_points and same[1].num_edges == out[1].num_edges


def test_morton_order_is_a_renumbering_the_model_cannot_see(dev):
    """``transforms.MortonOrder`` (round 5): levels re-ordered along a Morton curve, children
    re-grouped under their parents - the hierarchy stays consistent, the children of every
    superpoint become contiguous, spatial neighbours move close in memory, and the model's logits,
    put back in the original node order, are those of the un-ordered NAG (same sums, another
    order: f32 round-off)."""
    from superpoint_transformer_amd.hotpath import SPTSegmenter, spt64_config
    from superpoint_transformer_amd.synthetic import make_raw_nag
    from superpoint_transformer_amd import transforms as T
    torch.manual_seed(0)
    nag = make_raw_nag("R", device=dev)
    for t in (T.NodeSize(0), T.OnTheFlyHorizontalEdgeFeatures()):
        nag = t(nag)
    mo = T.MortonOrder()(nag)
    check_hierarchy(mo)
    assert mo.num_points == nag.num_points
    for i in range(nag.num_levels):
        origin = mo[i].morton_origin
        assert torch.equal(torch.sort(origin).values, torch.arange(nag.num_points[i], device=dev))
        assert torch.equal(mo[i].pos, nag[i].pos[origin])                  # a pure renumbering
        if i + 1 < nag.num_levels:                                         # parents follow their children
            assert torch.equal(mo[i + 1].morton_origin[mo[i].super_index], nag[i].super_index[origin])
            assert bool((mo[i].super_index[1:] >= mo[i].super_index[:-1]).all())   # children contiguous
    # edges join the same pairs of nodes
    e_new = mo[1].morton_origin[mo[1].edge_index]
    key = lambda e: torch.sort(e[0] * nag.num_points[1] + e[1]).values
    assert torch.equal(key(e_new), key(nag[1].edge_index))
    # locality: spatial neighbours move close in memory - the median index distance between a
    # level-1 node and its nearest other node shrinks (the synthetic graph's EDGES are drawn at
    # random, so their spans say nothing)
    def nn_span(pos):
        d = torch.cdist(pos, pos)
        d.fill_diagonal_(float("inf"))
        return float((d.argmin(1) - torch.arange(pos.shape[0], device=dev)).abs().float().median())
    assert nn_span(mo[1].pos) < 0.5 * nn_span(nag[1].pos)
    model = SPTSegmenter(**spt64_config(nag[0].x.shape[1], 18)).to(dev).eval()
    with torch.no_grad():
        ref = model(nag)
        got = model(mo)
    for i, (a, b) in enumerate(zip(ref, got)):
        back = T.MortonOrder.restore(mo, b, i + 1)
        assert torch.allclose(back, a, rtol=1e-4, atol=1e-5 * float(a.abs().max())), i
