"""GPU parity of the cluster-object overlap structure of the panoptic path:
``InstanceData`` (dense constructor, select, batching, major, merge, iou_and_size,
estimate_centroid, instance_graph, search_void / remove_void, label histogram, oracle) and
``OnTheFlyInstanceGraph`` against the fixture produced by the reference's own
src/data/instance.py + src/transforms/instance.py, and against the loop oracle on a larger
seeded case.  Integer / index outputs: BIT-EXACT.  Float outputs (IoU, affinities: a handful
of f32 operations; centroids: an f32 segment sum in another order): rtol 1e-6 / 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "instance_data.npz"))
NC = int(G["num_classes"])


def T(key, dev):
    return torch.from_numpy(G[key]).to(dev)


def same(a, key):
    b = torch.from_numpy(G[key])
    assert a.shape == b.shape and torch.equal(a.cpu().to(b.dtype), b), key


def close(a, key, tol):
    torch.testing.assert_close(a.cpu(), torch.from_numpy(G[key]), rtol=tol, atol=tol, msg=key)


def build(dev):
    from superpoint_transformer_amd.instance import InstanceData
    return InstanceData(T("in_cluster", dev), T("in_obj", dev), T("in_count", dev), T("in_y", dev),
                        dense=True)


def same_inst(d, prefix):
    for a, k in ((d.pointers, "pointers"), (d.obj, "obj"), (d.count, "count"), (d.y, "y")):
        if prefix + k in G:
            same(a, prefix + k)


def test_dense_constructor_select_merge_batch(dev):
    from superpoint_transformer_amd.instance import InstanceData
    d = build(dev)
    same_inst(d, "")
    same_inst(d.select(T("select_idx", dev)), "select_")
    same_inst(d.merge(T("merge_idx", dev)), "merge_")
    d2 = InstanceData(T("b2_pointers", dev), T("b2_obj", dev), T("b2_count", dev), T("b2_y", dev))
    same_inst(InstanceData.from_list([d, d2]), "batch_")
    ident = d.select(torch.arange(d.num_clusters, device=dev))
    same_inst(ident, "")
    with pytest.raises(AssertionError):                       # a parent without children
        d.merge(torch.full((d.num_clusters,), 3, device=dev))


def test_major_and_oracle(dev):
    d = build(dev)
    for tag, nc in (("nc", NC), ("all", None)):
        o, c, y = d.major(nc)
        same(o, f"major_{tag}_obj"), same(c, f"major_{tag}_count"), same(y, f"major_{tag}_y")
    sc, oy, od = d.oracle(NC)
    close(sc, "oracle_scores", 1e-6), same(oy, "oracle_y")
    same(od.pointers, "oracle_pointers"), same(od.obj, "oracle_obj"), same(od.count, "oracle_count")


def test_iou_centroids_and_void_handling(dev):
    d = build(dev)
    iou, a, b = d.iou_and_size()
    close(iou, "iou", 1e-6), same(a, "a_size"), same(b, "b_size")
    for mode in ("iou", "product-iou", "overlap"):
        p, oi = d.estimate_centroid(T("cluster_pos", dev), mode)
        close(p, f"centroid_{mode}_pos", 1e-5), same(oi, f"centroid_{mode}_idx")
    with pytest.raises(NotImplementedError):
        d.estimate_centroid(T("cluster_pos", dev), "ratio-product")
    cm, pm, crop = d.search_void(NC)
    same(cm, "void_cluster_mask"), same(pm, "void_pair_mask"), same(crop, "void_cropped")
    r, keep = d.remove_void(NC)
    same_inst(r, "rv_")
    same(r.pair_cropped_count, "rv_cropped"), same(keep, "rv_keep")
    iou, a, b = r.iou_and_size()
    close(iou, "rv_iou", 1e-6), same(a, "rv_a_size"), same(b, "rv_b_size")
    same(d.target_label_histogram(NC), "label_hist")


def test_instance_graph(dev):
    d = build(dev)
    for tag, smooth in (("smooth", True), ("hard", False)):
        for ctag, nc in (("nc", NC), ("all", None)):
            e, aff = d.instance_graph(T("edge_index", dev).clone(), nc, smooth)
            same(e, f"graph_{tag}_{ctag}_edge_index")
            close(aff, f"graph_{tag}_{ctag}_affinity", 1e-6)
    e, aff = d.instance_graph(torch.empty(2, 0, dtype=torch.long, device=dev))
    assert e.shape == (2, 0) and aff.numel() == 0


def test_on_the_fly_instance_graph_matches_the_reference(dev):
    from superpoint_transformer_amd.data import NAG, Data
    from superpoint_transformer_amd.transforms import OnTheFlyInstanceGraph
    lvl0 = Data(pos=torch.randn(10, 3, device=dev))
    lvl1 = Data(pos=T("cluster_pos", dev), edge_index=T("edge_index", dev).clone(), obj=build(dev))
    nag = OnTheFlyInstanceGraph(level=1, num_classes=NC, adjacency_mode="available")(
        NAG([lvl0, lvl1]))
    same(nag[1].obj_edge_index, "otf_iou_edge_index")
    close(nag[1].obj_edge_affinity, "otf_iou_affinity", 1e-6)
    close(nag[1].obj_pos, "otf_iou_obj_pos", 1e-5)
    # without annotations: the trimmed graph only
    bare = OnTheFlyInstanceGraph(level=1, adjacency_mode="available")(
        NAG([lvl0, Data(pos=T("cluster_pos", dev), edge_index=T("edge_index", dev).clone())]))
    same(bare[1].obj_edge_index, "otf_iou_edge_index")
    assert "obj_edge_affinity" not in bare[1] and "obj_pos" not in bare[1]
    assert OnTheFlyInstanceGraph(level=-1)(nag) is nag


def _random_instance(seed, n_cl, n_obj, nc):
    rng = np.random.default_rng(seed)
    obj_y = rng.integers(0, nc, n_obj)
    obj_y[rng.random(n_obj) < 0.2] = nc
    k = rng.integers(1, 7, n_cl)
    cl = np.repeat(np.arange(n_cl), k)
    ob = np.concatenate([rng.choice(n_obj, kk, replace=False) for kk in k])
    dup = rng.integers(0, cl.size, cl.size // 5)
    cl, ob = np.concatenate([cl, cl[dup]]), np.concatenate([ob, ob[dup]])
    p = rng.permutation(cl.size)
    cl, ob = cl[p], ob[p]
    return cl, ob * 7 + 2, rng.integers(1, 3000, cl.size), obj_y[ob]


def test_larger_seeded_case_against_the_loop_oracle(dev):
    from superpoint_transformer_amd.instance import InstanceData
    nc = 9
    cl, ob, cnt, y = _random_instance(5, 700, 150, nc)
    ref = O.instance_from_dense(cl, ob, cnt, y)
    d = InstanceData(*(torch.from_numpy(t).to(dev) for t in (cl, ob, cnt, y)), dense=True)
    for a, b in zip((d.pointers, d.obj, d.count, d.y), ref):
        assert np.array_equal(a.cpu().numpy(), b)
    for a, b in zip(d.major(nc), O.instance_major(ref, nc)):
        assert np.array_equal(a.cpu().numpy(), b)
    rng = np.random.default_rng(6)
    ei = rng.integers(0, 700, (2, 6000))
    e, aff = d.instance_graph(torch.from_numpy(ei).to(dev), nc)
    e_ref, aff_ref = O.instance_graph(ref, ei, nc)
    assert np.array_equal(e.cpu().numpy(), e_ref)
    np.testing.assert_allclose(aff.cpu().numpy(), aff_ref, rtol=1e-6)
    cm, pm, crop = d.search_void(nc)
    for a, b in zip((cm, pm, crop), O.instance_search_void(ref, nc)):
        assert np.array_equal(a.cpu().numpy(), b)
    pos = rng.standard_normal((700, 3)).astype(np.float32)
    p, ids = d.estimate_centroid(torch.from_numpy(pos).to(dev))
    p_ref, ids_ref = O.instance_estimate_centroid(ref, pos)
    assert np.array_equal(ids.cpu().numpy(), ids_ref)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref, rtol=1e-4, atol=1e-5)


def test_overlaps_follow_selection_and_batching_of_a_nag(dev):
    """``Data.select`` re-indexes ``obj`` with the nodes (data.py:437-439) and
    ``NAG.from_nag_list`` keeps the objects of different items apart (csr.py:715-731)."""
    from superpoint_transformer_amd.data import NAG, Data
    g = torch.Generator().manual_seed(3)
    n0, n1 = 600, 90
    si = torch.randint(0, n1, (n0,), generator=g)
    si[:n1] = torch.arange(n1)
    d = build(dev)
    lvl0 = Data(pos=torch.randn(n0, 3, generator=g).to(dev), super_index=si.to(dev))
    from superpoint_transformer_amd.data import Cluster
    lvl1 = Data(pos=T("cluster_pos", dev), obj=d,
                sub=Cluster(si.to(dev), torch.arange(n0, device=dev), dense=True))
    nag = NAG([lvl0, lvl1])
    idx = T("select_idx", dev)
    sel = nag.select(1, idx)
    same_inst(sel[1].obj, "select_")
    both = NAG.from_nag_list([nag, sel])
    ref = type(d).from_list([d, sel[1].obj])
    for a, b in zip(both[1].obj.values + [both[1].obj.pointers], ref.values + [ref.pointers]):
        assert torch.equal(a, b)
    assert int(both[1].obj.obj[d.num_items:].min()) > int(d.obj.max())


def test_radius_centroid_adjacency_is_every_pair_within_the_radius(dev):
    from superpoint_transformer_amd.data import NAG, Data
    from superpoint_transformer_amd.transforms import OnTheFlyInstanceGraph
    g = torch.Generator().manual_seed(11)
    pos = torch.rand(400, 3, generator=g) * 6
    batch = (torch.arange(400) >= 250).long()
    r = 0.9
    lvl1 = Data(pos=pos.to(dev), batch=batch.to(dev))
    nag = OnTheFlyInstanceGraph(level=1, adjacency_mode="radius-centroid", k_max=30, radius=r)(
        NAG([Data(pos=torch.zeros(1, 3, device=dev)), lvl1]))
    d2 = ((pos[:, None].double() - pos[None].double()) ** 2).sum(-1)
    ok = (d2 < r * r) & (batch[:, None] == batch[None]) & torch.triu(torch.ones(400, 400, dtype=torch.bool), 1)
    # pairs closer than 1e-6 to the radius may fall either side in f32
    margin = (d2.sqrt() - r).abs() < 1e-5
    got = torch.zeros(400, 400, dtype=torch.bool)
    e = nag[1].obj_edge_index.cpu()
    got[e[0], e[1]] = True
    assert bool((e[0] < e[1]).all())
    assert torch.equal(got | margin, ok | margin)
    key = e[0] * 400 + e[1]
    assert bool((key[1:] > key[:-1]).all())                      # sorted, duplicate-free


def test_knn_1_graph_keeps_the_smaller_distance_once(dev):
    from superpoint_transformer_amd.neighbors import knn_1, knn_1_graph
    g = torch.Generator().manual_seed(12)
    xyz = (torch.rand(3000, 3, generator=g) * 4).to(dev)
    nn, dist = knn_1(xyz, 8, r_max=0.5)
    e, d = knn_1_graph(xyz, 8, r_max=0.5)
    src = torch.arange(3000, device=dev).repeat_interleave(8)
    tgt, dd = nn.flatten(), dist.flatten()
    m = tgt >= 0
    lo, hi = torch.minimum(src[m], tgt[m]), torch.maximum(src[m], tgt[m])
    ref = {}
    for a, b, x in zip(lo.tolist(), hi.tolist(), dd[m].tolist()):
        if a != b:
            ref[(a, b)] = min(ref.get((a, b), float("inf")), x)
    keys = sorted(ref)
    assert e.t().tolist() == [list(k) for k in keys]
    assert d.tolist() == [ref[k] for k in keys]
    e2, _ = knn_1_graph(xyz, 8, r_max=0.5, trim=False)
    assert e2.shape[1] == int(m.sum())                           # directed, no duplicates in a kNN list


def test_segment_sampling_weights(dev):
    """sampling.py:771-798: uniform + cube-root-of-size + rarest-class terms."""
    from superpoint_transformer_amd.data import NAG, Data
    from superpoint_transformer_amd.transforms import (SampleRadiusSubgraphs, SampleSegments,
                                                       segment_sampling_weights)
    g = torch.Generator().manual_seed(2)
    n0, n1, nc = 5000, 120, 6
    si = torch.randint(0, n1, (n0,), generator=g)
    si[:n1] = torch.arange(n1)
    hist = torch.randint(0, 40, (n1, nc + 1), generator=g) * (torch.rand(n1, nc + 1, generator=g) < 0.4)
    hist[:, 2] = 0                                                # a class nobody holds
    nag = NAG([Data(pos=torch.randn(n0, 3, generator=g).to(dev), super_index=si.to(dev)),
               Data(pos=torch.randn(n1, 3, generator=g).to(dev), y=hist.to(dev))])
    size = np.bincount(si.numpy(), minlength=n1)
    for by_size in (False, True):
        for by_class in (False, True):
            w = O.segment_sampling_weights(size, hist.numpy(), by_size, by_class)
            got = segment_sampling_weights(nag, 1, by_size, by_class)
            np.testing.assert_allclose(got.cpu().numpy(), w, rtol=2e-6)
    # and against the weights the reference's own SampleSegments hands to torch.multinomial
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampling_weights.npz"))
    gsi, gy = torch.from_numpy(gold["super_index"]), torch.from_numpy(gold["y"])
    gnag = NAG([Data(pos=torch.zeros(gsi.numel(), 3, device=dev), super_index=gsi.to(dev)),
                Data(pos=torch.zeros(gy.shape[0], 3, device=dev), y=gy.to(dev))])
    for by_size in (False, True):
        for by_class in (False, True):
            got = segment_sampling_weights(gnag, 1, by_size, by_class)
            np.testing.assert_allclose(got.cpu().numpy(), gold[f"w_{int(by_size)}{int(by_class)}"],
                                       rtol=5e-6)
    out = SampleSegments(ratio=0.25, by_size=True, by_class=True)(nag)
    assert out[1].num_nodes == n1 - int(n1 * 0.25)
    sub = SampleRadiusSubgraphs(r=1.0, k=3, i_level=1, by_size=True, by_class=True)(nag)
    assert 0 < sub[1].num_nodes <= n1
