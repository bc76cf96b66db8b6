"""The oracle's restatement of the cluster-object overlap structure (src/data/instance.py)
against the fixture the reference's own InstanceData produced
(tests/golden/make_golden_instance.py)."""
import os

import numpy as np

from oracle import spt_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "instance_data.npz"))
NC = int(G["num_classes"])


def _inst(prefix=""):
    return tuple(G[prefix + k].astype(np.int64) for k in ("pointers", "obj", "count", "y"))


def test_dense_constructor_merges_duplicates():
    got = O.instance_from_dense(G["in_cluster"], G["in_obj"], G["in_count"], G["in_y"])
    for a, b in zip(got, _inst()):
        assert np.array_equal(a, b)


def test_major_with_and_without_void_labels():
    for tag, nc in (("nc", NC), ("all", None)):
        o, c, y = O.instance_major(_inst(), nc)
        assert np.array_equal(o, G[f"major_{tag}_obj"])
        assert np.array_equal(c, G[f"major_{tag}_count"])
        assert np.array_equal(y, G[f"major_{tag}_y"])
    # the fixture exercises the second-best (non-void) branch
    assert (G["major_nc_obj"] != G["major_all_obj"]).any()


def test_merge():
    ptr, obj, count, y = _inst()
    parent = G["merge_idx"][O.instance_indices(ptr)]
    got = O.instance_from_dense(parent, obj, count, y)
    for a, k in zip(got, ("merge_pointers", "merge_obj", "merge_count", "merge_y")):
        assert np.array_equal(a, G[k])


def test_iou_and_size():
    iou, a, b = O.instance_iou_and_size(_inst())
    assert np.array_equal(a, G["a_size"]) and np.array_equal(b, G["b_size"])
    np.testing.assert_allclose(iou, G["iou"], rtol=1e-6)
    iou, a, b = O.instance_iou_and_size(_inst("rv_"), G["rv_cropped"])
    assert np.array_equal(a, G["rv_a_size"]) and np.array_equal(b, G["rv_b_size"])
    np.testing.assert_allclose(iou, G["rv_iou"], rtol=1e-6)


def test_estimate_centroid():
    for mode in ("iou", "product-iou", "overlap"):
        pos, ids = O.instance_estimate_centroid(_inst(), G["cluster_pos"], mode)
        assert np.array_equal(ids, G[f"centroid_{mode}_idx"])
        np.testing.assert_allclose(pos, G[f"centroid_{mode}_pos"], rtol=1e-5, atol=1e-6)


def test_instance_graph():
    for tag, smooth in (("smooth", True), ("hard", False)):
        for ctag, nc in (("nc", NC), ("all", None)):
            e, aff = O.instance_graph(_inst(), G["edge_index"], nc, smooth)
            assert np.array_equal(e, G[f"graph_{tag}_{ctag}_edge_index"])
            np.testing.assert_allclose(aff, G[f"graph_{tag}_{ctag}_affinity"], rtol=1e-6)
    assert np.array_equal(G["otf_iou_edge_index"], G["graph_smooth_nc_edge_index"])


def test_search_void():
    cm, pm, crop = O.instance_search_void(_inst(), NC)
    assert np.array_equal(cm, G["void_cluster_mask"])
    assert np.array_equal(pm, G["void_pair_mask"])
    assert np.array_equal(crop, G["void_cropped"])
    assert cm.any() and not cm.all()
