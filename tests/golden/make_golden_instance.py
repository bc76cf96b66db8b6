"""Golden fixture for the cluster-object overlap structure of the panoptic path,
produced by the REFERENCE'S OWN code:

  * ``src/data/csr.py`` and ``src/data/instance.py`` are imported verbatim by path
    (InstanceData: dense constructor, select, major, merge, iou_and_size, estimate_centroid,
    instance_graph, search_void, remove_void, target_label_histogram, oracle; CSRBatch.from_list);
  * ``OnTheFlyInstanceGraph._process`` (src/transforms/instance.py:133-257) and
    ``Data.estimate_instance_centroid`` (src/data/data.py:941-974) are cut out of their files
    with ``ast`` - unmodified - and run on duck-typed NAG / Data stores (their modules pull
    torch_geometric's Data, h5py and the metrics at import time).

Stand-ins: torch_scatter / PyG coalesce, remove_self_loops, consecutive_cluster = the oracle's
restatements (ties of scatter_max -> first occurrence, the CPU kernel rule).

Usage (build container only): python tests/golden/make_golden_instance.py
"""
import ast
import importlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_select as mgs  # noqa: E402

REF = mg.REF
NUM_CLASSES = 5


def load_reference():
    U, csr, cluster = mgs.load_reference()
    graph = _graph_on_existing_hooks()
    U.to_trimmed = graph.to_trimmed
    inst = importlib.import_module("src.data.instance")
    return U, csr, inst


def _graph_on_existing_hooks():
    """src/utils/graph.py on the hooks already installed by make_golden_select (installing
    them twice would replace the ``src.utils`` package the csr module was bound to)."""
    from oracle import spt_oracle as O
    tgu = sys.modules["torch_geometric.utils"]
    tgu.coalesce = O.coalesce
    tgu.remove_self_loops = O.remove_self_loops
    U = sys.modules["src.utils"]
    for name in ("scatter", "neighbors", "edge", "sparse"):
        m = sys.modules.get(f"src.utils.{name}") or importlib.import_module(f"src.utils.{name}")
        if hasattr(m, "coalesce"):
            m.coalesce = O.coalesce
        if hasattr(m, "consecutive_cluster"):
            m.consecutive_cluster = O.consecutive_cluster
    edge = importlib.import_module("src.utils.edge")
    edge.consecutive_cluster = O.consecutive_cluster
    U.edge_wise_points = edge.edge_wise_points
    sys.modules["src.utils.scatter"].edge_wise_points = edge.edge_wise_points
    return importlib.import_module("src.utils.graph")


def cut(path, cls, name):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    return ast.Module(body=[fn], type_ignores=[])


def synth_pairs(gen, n_clusters, n_obj, void_frac=0.25, dup=40):
    """Dense (cluster, obj, count, y) pairs WITH duplicates; object ids sparse (3 * k + 1),
    objects semantically pure, some of them void (label NUM_CLASSES or -1)."""
    obj_y = torch.randint(0, NUM_CLASSES, (n_obj,), generator=gen)
    void = torch.rand(n_obj, generator=gen) < void_frac
    obj_y[void] = torch.where(torch.rand(int(void.sum()), generator=gen) < 0.5,
                              torch.tensor(NUM_CLASSES), torch.tensor(-1))
    cl, ob = [], []
    for c in range(n_clusters):
        k = int(torch.randint(1, 5, (1,), generator=gen))
        o = torch.randperm(n_obj, generator=gen)[:k]
        cl.append(torch.full((k,), c))
        ob.append(o)
    cl, ob = torch.cat(cl), torch.cat(ob)
    extra = torch.randint(0, cl.numel(), (dup,), generator=gen)          # duplicated pairs
    cl, ob = torch.cat([cl, cl[extra]]), torch.cat([ob, ob[extra]])
    perm = torch.randperm(cl.numel(), generator=gen)
    cl, ob = cl[perm], ob[perm]
    count = torch.randint(1, 120, (cl.numel(),), generator=gen)
    # a few clusters dominated by a void object with < 50 % of their points, a few with more
    return cl, ob * 3 + 1, count, obj_y[ob]


def synth_edges(gen, n, m):
    s = torch.randint(0, n, (m,), generator=gen)
    t = torch.randint(0, n, (m,), generator=gen)
    loops = torch.arange(0, n, 7)
    return torch.stack([torch.cat([s, t[:m // 3], loops]), torch.cat([t, s[:m // 3], loops])])


def main():
    U, csr, inst = load_reference()
    ID = inst.InstanceData
    gen = torch.Generator().manual_seed(20240924)
    out = {}

    n_cl, n_obj = 90, 23
    cl, ob, cnt, y = synth_pairs(gen, n_cl, n_obj)
    out.update(in_cluster=cl, in_obj=ob, in_count=cnt, in_y=y, num_classes=NUM_CLASSES)
    d = ID(cl.clone(), ob.clone(), cnt.clone(), y.clone(), dense=True)
    out.update(pointers=d.pointers, obj=d.obj, count=d.count, y=d.y)

    for tag, nc in (("nc", NUM_CLASSES), ("all", None)):
        o, c, yy = d.major(num_classes=nc)
        out.update({f"major_{tag}_obj": o, f"major_{tag}_count": c, f"major_{tag}_y": yy})

    idx = torch.randperm(n_cl, generator=gen)[:37]
    s = d.select(idx)
    out.update(select_idx=idx, select_pointers=s.pointers, select_obj=s.obj,
               select_count=s.count, select_y=s.y)

    parent = torch.randint(0, 17, (n_cl,), generator=gen)
    parent[:17] = torch.arange(17)
    m = d.merge(parent)
    out.update(merge_idx=parent, merge_pointers=m.pointers, merge_obj=m.obj,
               merge_count=m.count, merge_y=m.y)

    iou, a_size, b_size = d.iou_and_size()
    out.update(iou=iou, a_size=a_size, b_size=b_size)

    pos = torch.randn(n_cl, 3, generator=gen).float()
    out["cluster_pos"] = pos
    for mode in ("iou", "product-iou", "overlap"):
        p, oi = d.estimate_centroid(pos, mode=mode)
        out[f"centroid_{mode}_pos"], out[f"centroid_{mode}_idx"] = p, oi

    ei = synth_edges(gen, n_cl, 400)
    out["edge_index"] = ei
    for tag, smooth in (("smooth", True), ("hard", False)):
        for ctag, nc in (("nc", NUM_CLASSES), ("all", None)):
            e2, aff = d.instance_graph(ei.clone(), num_classes=nc, smooth_affinity=smooth)
            out[f"graph_{tag}_{ctag}_edge_index"], out[f"graph_{tag}_{ctag}_affinity"] = e2, aff

    cm, pm, crop = d.search_void(NUM_CLASSES)
    out.update(void_cluster_mask=cm, void_pair_mask=pm, void_cropped=crop)
    r, keep = d.remove_void(NUM_CLASSES)
    out.update(rv_pointers=r.pointers, rv_obj=r.obj, rv_count=r.count, rv_y=r.y,
               rv_cropped=r.pair_cropped_count, rv_keep=keep)
    iou2, a2, b2 = r.iou_and_size()
    out.update(rv_iou=iou2, rv_a_size=a2, rv_b_size=b2)
    out["label_hist"] = d.target_label_histogram(NUM_CLASSES)
    sc, oy, od = d.oracle(NUM_CLASSES)
    out.update(oracle_scores=sc, oracle_y=oy, oracle_pointers=od.pointers, oracle_obj=od.obj,
               oracle_count=od.count)

    # batching: obj indices of the second item shifted past the first item's largest
    cl2, ob2, cnt2, y2 = synth_pairs(gen, 31, 9, dup=5)
    d2 = ID(cl2, ob2, cnt2, y2, dense=True)
    b = inst.InstanceBatch.from_list([d, d2])
    out.update(b2_pointers=d2.pointers, b2_obj=d2.obj, b2_count=d2.count, b2_y=d2.y,
               batch_pointers=b.pointers, batch_obj=b.obj, batch_count=b.count, batch_y=b.y)

    # OnTheFlyInstanceGraph._process, adjacency_mode='available', on a duck NAG of two levels
    # (level 1 carries the overlaps, the graph and the positions; centroid_level = 1)
    ns = {"torch": torch, "to_trimmed": U.to_trimmed,
          "consecutive_cluster": sys.modules["torch_geometric.nn.pool.consecutive"].consecutive_cluster,
          "cluster_radius_nn_graph": None, "knn_1_graph": None}
    exec(compile(cut("src/transforms/instance.py", "OnTheFlyInstanceGraph", "_process"),
                 "instance.py", "exec"), ns)
    exec(compile(cut("src/data/data.py", "Data", "estimate_instance_centroid"),
                 "data.py", "exec"), ns)

    class Duck(mgs.DuckData):
        estimate_instance_centroid = ns["estimate_instance_centroid"]

        def __getattr__(self, k):
            if k == "obj":
                return object.__getattribute__(self, "_s").get("obj")
            return mgs.DuckData.__getattr__(self, k)

    lvl0 = Duck(pos=torch.randn(10, 3, generator=gen).float())
    lvl1 = Duck(pos=pos, edge_index=ei.clone(), obj=d)
    nag = mgs.DuckNAG([lvl0, lvl1])
    nag.has_atoms = True
    for ctag, cmode in (("iou", "iou"),):
        t = types.SimpleNamespace(level=1, num_classes=NUM_CLASSES, adjacency_mode="available",
                                  k_max=30, radius=1, use_batch=True, centroid_mode=cmode,
                                  centroid_level=1, smooth_affinity=True)
        res = ns["_process"](t, nag)
        out.update({f"otf_{ctag}_edge_index": res[1].obj_edge_index,
                    f"otf_{ctag}_affinity": res[1].obj_edge_affinity,
                    f"otf_{ctag}_obj_pos": res[1].obj_pos})
    mg.save("instance_data.npz", **out)


if __name__ == "__main__":
    main()
