"""Host-side pieces the GPU path leans on, runnable without a GPU: the fused layers' run tables, the synthetic NAG generator
(bench.py's scenes: the stored cluster CSR ``sub`` of every level, multi-cloud batches, the
``--graph local`` / ``--order morton`` variants) and the ``Cluster`` / ``Data`` behaviour the
CSR adoption of round 4 relies on (src/data/cluster.py:19-77, src/data/nag.py:878-898)."""
import pytest
import torch

from superpoint_transformer_amd import synthetic
from superpoint_transformer_amd.data import Cluster, Data

SIZES = (6_000, 400, 150, 3_000, 1_800, 1)
SIZES_B3 = (9_000, 600, 210, 4_500, 2_700, 3)


def _check_sub(levels):
    for lo, hi in zip(levels[:-1], levels[1:]):
        si, sub = lo["super_index"], hi["sub"]
        n_parent = hi["pos"].shape[0]
        assert sub.pointers.numel() == n_parent + 1 and int(sub.pointers[-1]) == si.numel()
        # the same partition: the children listed for cluster c all point at c ...
        owner = torch.repeat_interleave(torch.arange(n_parent), sub.pointers[1:] - sub.pointers[:-1])
        assert torch.equal(si[sub.points], owner)
        # ... every child exactly once, ascending inside a cluster (= the stable sort of super_index)
        assert torch.equal(sub.points, torch.argsort(si, stable=True))
        assert sub.ascending


@pytest.mark.parametrize("graph,order", [("random", "storage"), ("local", "storage"),
                                         ("random", "morton"), ("local", "morton")])
def test_synthetic_nag_levels_are_consistent(graph, order):
    nag = synthetic.make_nag(sizes=SIZES, seed=5, graph=graph, order=order)
    lv = nag.levels
    n0, n1, n2 = SIZES[:3]
    assert [l["pos"].shape[0] for l in lv] == [n0, n1, n2]
    _check_sub(lv)
    # node_size = points below every node (NodeSize, src/transforms/graph.py:1475-1498)
    assert torch.equal(lv[1]["node_size"], torch.bincount(lv[0]["super_index"], minlength=n1))
    assert int(lv[2]["node_size"].sum()) == n0
    for l, n in ((lv[1], n1), (lv[2], n2)):
        ei = l["edge_index"]
        assert ei.shape[0] == 2 and int(ei.min()) >= 0 and int(ei.max()) < n
        assert l["edge_attr"].shape[0] == ei.shape[1]
        # both directions and every self-loop are present (the doubled graph + NAGAddSelfLoops)
        key = set((ei[0] * n + ei[1]).tolist())
        assert all(int(s) * n + int(s) in key for s in range(0, n, max(n // 50, 1)))
        assert all((int(t) * n + int(s)) in key for s, t in ei[:, :200].t().tolist())
    # children jitter around their parents: positions survive any relabelling
    d = (lv[0]["pos"] - lv[1]["pos"][lv[0]["super_index"]]).norm(dim=1)
    assert float(d.mean()) < 1.0


def test_default_scene_is_the_same_scene_as_before():
    """The default generator (random graph, storage order) is what every committed bench line was
    measured on: same seed -> same arrays, with or without the optional arguments spelled out."""
    a = synthetic.make_nag(sizes=SIZES, seed=1234)
    b = synthetic.make_nag(sizes=SIZES, seed=1234, graph="random", order="storage")
    for la, lb in zip(a.levels, b.levels):
        for k in ("pos", "super_index", "edge_index", "edge_attr", "batch"):
            if la.get(k) is not None:
                assert torch.equal(la[k], lb[k])
    c = synthetic.make_nag(sizes=SIZES, seed=1235)
    assert not torch.equal(a.levels[1]["edge_index"], c.levels[1]["edge_index"])


def test_morton_order_puts_graph_neighbours_close_in_memory():
    far = synthetic.make_nag(sizes=SIZES, seed=7, graph="local", order="storage").levels[1]["edge_index"]
    near = synthetic.make_nag(sizes=SIZES, seed=7, graph="local", order="morton").levels[1]["edge_index"]
    gap = lambda ei: float((ei[0] - ei[1]).abs().double().median())
    assert gap(near) < 0.25 * gap(far)


@pytest.mark.parametrize("graph", ["random", "local"])
def test_multi_cloud_batch_keeps_clouds_contiguous_and_edges_inside_them(graph):
    nag = synthetic.make_nag(sizes=SIZES_B3, seed=11, graph=graph)
    assert nag.num_clouds == 3
    lv = nag.levels
    _check_sub(lv)
    for l in lv:
        b = l["batch"]
        assert bool((b[1:] >= b[:-1]).all()) and int(b.max()) == 2          # NAGBatch.from_nag_list
    assert torch.equal(lv[0]["batch"], lv[1]["batch"][lv[0]["super_index"]])
    for l in lv[1:]:
        ei, b = l["edge_index"], l["batch"]
        assert torch.equal(b[ei[0]], b[ei[1]])                              # no edge between clouds
        eb = b[ei[0]]
        # the edge MLP's norm index: a few sorted runs (the fused layers' run table holds 16)
        runs = 1 + int((eb[1:] != eb[:-1]).sum())
        assert runs <= 16


def test_cluster_ascending_flag():
    ptr = torch.tensor([0, 2, 2, 5])
    asc = Cluster(ptr, torch.tensor([4, 7, 0, 1, 9]))
    assert asc.ascending                                  # worked out on first use
    assert not Cluster(ptr, torch.tensor([7, 4, 0, 1, 9])).ascending
    assert Cluster(ptr, torch.tensor([7, 4, 0, 1, 9]), ascending=True).ascending   # the caller vouches
    # empty clusters and the boundary between clusters do not count as descents
    assert Cluster(torch.tensor([0, 0, 1, 1, 3]), torch.tensor([5, 0, 2])).ascending
    assert asc.clone().ascending and asc.to("cpu").ascending


def test_data_item_access_is_strict():
    d = Data(pos=torch.zeros(3, 3), x=torch.ones(3, 2))
    assert d["pos"].shape == (3, 3) and d.get("x") is not None
    assert d.get("nope") is None and d.get("nope", 7) == 7
    with pytest.raises(KeyError):
        d["nope"]


def test_bf16_storage_switch_round_trips():
    from superpoint_transformer_amd import precision
    prev = precision.set_bf16_activation_storage(False)
    try:
        with precision.matrix_precision("bf16"):
            assert precision.bf16_activation_storage() is False
            assert precision.set_bf16_activation_storage(True) is False
            assert precision.bf16_activation_storage() is True          # bf16 mode + the switch
        with precision.matrix_precision("f32"):
            assert precision.bf16_activation_storage() is False         # f32 storage in the f32 modes
    finally:
        precision.set_bf16_activation_storage(prev)
    with pytest.raises(ValueError):
        precision.set_matrix_precision("fp8")


def test_graph_runs_of_sorted_and_piecewise_sorted_indices():
    """``ops.graph_runs``: the run table one launch of the fused layer kernels covers - pure host
    logic over the batch index (the GPU tests launch with it; here its shape is pinned)."""
    from superpoint_transformer_amd import ops
    rows = 1200
    one = ops.graph_runs(None, None, rows)
    assert one.n == 1 and one.sorted_batch and one.rows_per_graph() == [rows]
    b = torch.arange(rows) * 3 // rows
    r = ops.graph_runs(b, 3, rows)
    assert r.sorted_batch and r.g == [0, 1, 2] and r.r0[0] == 0 and r.r1[-1] == rows
    assert r.r1[:-1] == r.r0[1:] and r.rows_per_graph() == [400, 400, 400]
    assert ops.graph_runs(b, 3, rows) is r                              # memoised on the tensor
    n, r0, r1, g = r.c_arrays()
    assert n == 3 and list(r0) == r.r0 and list(r1) == r.r1 and list(g) == r.g
    # the edge MLP's norm index of a 4-cloud batch: sorted inside each third of [i<j | j>i | loops]
    third = torch.arange(rows // 3) * 4 // (rows // 3)
    p = torch.cat([third, third, third])
    r = ops.graph_runs(p, 4, rows)
    assert r is not None and not r.sorted_batch and r.n == 12
    assert r.g == sorted(r.g)                                           # runs grouped by graph
    assert r.rows_per_graph() == [int((p == k).sum()) for k in range(4)]
    # more runs than the kernels' table holds, an id outside [0, B): no table (layer-by-layer route)
    assert ops.graph_runs(torch.randint(0, 3, (rows,), generator=torch.Generator().manual_seed(1)),
                          3, rows) is None
    assert ops.graph_runs(torch.arange(rows) * 3 // rows + 1, 3, rows) is None
    assert ops.graph_runs(b.clone(), ops.MAX_FUSED_RUNS + 1, rows) is None
    # a derived index (norm_index[edge_index[0]]): the table is remembered on the holder
    holder = torch.zeros(2, rows, dtype=torch.long)
    d1 = p.clone()
    r1 = ops.graph_runs_via(d1, 4, holder, holder, p)
    d2 = p.clone()                                                      # rebuilt every forward
    assert ops.graph_runs_via(d2, 4, holder, holder, p) is r1
    holder.add_(0)                                                      # in-place change: new version
    assert ops.graph_runs_via(p.clone(), 4, holder, holder, p) is not r1


def test_run_tables_from_host_knowledge_and_forget():
    """Round 5: the run tables of the fused layers are host knowledge of the batch layout (the
    reference's ``Batch.ptr``): where a batch's maker leaves the clouds' node ranges on the batch
    vector (`_spt_host_ptr`) and the runs of constant cloud id on the edge index
    (`_spt_host_runs`), they are built WITHOUT reading the device tensor back - and `csr.forget`
    (what a benchmark calls every step to stand for a fresh batch) drops the memoised tables, so
    a batch without that knowledge pays its read-back every step, as real training does."""
    from superpoint_transformer_amd import csr, ops
    nag = synthetic.make_nag(sizes=SIZES_B3, seed=2)
    for lv in nag.levels:
        b = lv["batch"]
        hp = b._spt_host_ptr
        assert hp[0] == 0 and hp[-1] == b.numel() and len(hp) == 4
        assert hp == [0] + torch.cumsum(torch.bincount(b, minlength=3), 0).tolist()
        r = ops.graph_runs(b, 3, b.numel())
        assert r.sorted_batch and r.r0 == hp[:-1] and r.r1 == hp[1:] and r.g == [0, 1, 2]
        assert ops.graph_ranges(b, 3, b.numel()) == hp
        assert hasattr(b, "_spt_graph_runs")
        csr.forget(b)
        assert not hasattr(b, "_spt_graph_runs") and not hasattr(b, "_spt_graph_ranges")
        assert hasattr(b, "_spt_host_ptr")                            # knowledge of the batch stays
    # a poisoned device tensor shows that the host route never looks at the values
    b = nag.levels[1]["batch"]
    fake = torch.full_like(b, 7)
    fake._spt_host_ptr = list(b._spt_host_ptr)
    assert ops.graph_runs(fake, 3, fake.numel()).g == [0, 1, 2]
    # ... and a stale pointer list (wrong length / wrong total) is not trusted
    stale = b.clone()
    stale._spt_host_ptr = [0, 5, 9]
    assert ops.graph_runs(stale, 3, stale.numel()).r1[-1] == stale.numel()
    for i in (1, 2):
        lv = nag.levels[i]
        ei, bb = lv["edge_index"], lv["batch"]
        (B, ne), runs = ei._spt_host_runs
        assert B == 3 and ne == ei.shape[1] and len(runs) == 9        # 3 clouds x [i<j | j>i | loops]
        eb = bb[ei[0]]
        for a, e, g in runs:
            assert bool((eb[a:e] == g).all())
        fake = torch.full_like(eb, 9)                                 # never read on the host route
        r = ops.graph_runs_via(fake, 3, ei, ei, bb)
        assert r.n == 9 and r.rows_per_graph() == [int((eb == k).sum()) for k in range(3)]


def test_grouped_order_is_the_morton_transform_layout():
    """``order="grouped"`` of the synthetic generator = what ``transforms.MortonOrder`` produces:
    children contiguous under their parents at every level, clouds contiguous."""
    from superpoint_transformer_amd.transforms import morton_code
    nag = synthetic.make_nag(sizes=SIZES_B3, seed=4, graph="local", order="grouped")
    lv = nag.levels
    _check_sub(lv)
    for l in lv[:2]:
        si = l["super_index"]
        assert bool((si[1:] >= si[:-1]).all())                        # grouped by parent
    for l in lv:
        b = l["batch"]
        assert bool((b[1:] >= b[:-1]).all())
    assert torch.equal(lv[1]["sub"].points, torch.arange(SIZES_B3[0]))   # the pool's view: identity
    code = morton_code(torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [1.0, 0.0, 0.0]]), bits=2)
    assert code.tolist() == [0, 63, 9]                               # x is the lowest bit of a triple


def test_scene_mix_table_of_the_bench():
    import bench
    assert len(bench.SCENE_MIX) == 6 and abs(sum(bench.SCENE_MIX) / 6 - 1.0) < 0.01
    assert max(bench.SCENE_MIX) == 1.73                               # Area 5


def test_product_ops_refuse_cpu_tensors_loudly():
    """There is no CPU fallback anywhere in the product: an op handed a CPU tensor raises at once
    (the message says so) instead of computing something else somewhere else; argument errors that
    need no device are reported before that."""
    import pytest
    import torch
    from superpoint_transformer_amd import neighbors as NB
    from superpoint_transformer_amd import ops
    xyz = torch.rand(64, 3)
    nn = torch.randint(0, 64, (64, 5))
    idx = torch.randint(0, 8, (64,))
    for call in (lambda: NB.knn_1(xyz, 5, 0.5),
                 lambda: NB.knn_1_features(xyz, 5, 0.5),
                 lambda: NB.geometric_features(xyz, nn),
                 lambda: NB.frnn_grid_points(xyz, xyz, 4, 0.5),
                 lambda: ops.segment_reduce(torch.rand(64, 8), idx, 8, "max")):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    with pytest.raises(ValueError, match="k \\+ 1 <= 64"):
        NB.knn_1_features(xyz, 64, 0.5)
