"""On-the-fly horizontal edge features + self loops.

Golden vector: tests/golden/horizontal_edge_features.npz, produced by executing
the REFERENCE's own `_on_the_fly_horizontal_edge_features` source on level 1 of
its demo room (tests/golden/make_golden_edge_features.py).  CPU: the oracle
matches it; GPU: the fused HIP kernel matches oracle and fixture."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spt_oracle as O


def _inputs(g):
    t = lambda k: torch.from_numpy(g["in__" + k])
    return (t("edge_index"), t("edge_attr"), t("pos"), t("normal"), t("log_length"),
            t("log_surface"), t("log_volume"), t("log_size"))


def test_oracle_matches_reference_function_output():
    g = load_golden("horizontal_edge_features.npz")
    ei, attr = O.horizontal_edge_features(*_inputs(g), add_self_loops=False)
    assert torch.equal(ei, torch.from_numpy(g["edge_index"]))
    torch.testing.assert_close(attr, torch.from_numpy(g["edge_attr"]), rtol=1e-6, atol=1e-7)
    assert attr.shape[1] == 18


def test_oracle_self_loops_are_zero_rows_at_the_end():
    g = load_golden("horizontal_edge_features.npz")
    ins = _inputs(g)
    n, e = ins[2].shape[0], ins[0].shape[1]
    ei, attr = O.horizontal_edge_features(*ins, add_self_loops=True)
    assert ei.shape == (2, 2 * e + n) and attr.shape == (2 * e + n, 18)
    assert torch.equal(ei[:, 2 * e:], torch.arange(n).repeat(2, 1))
    assert attr[2 * e:].abs().sum() == 0


@pytest.mark.gpu
def test_fused_kernel_matches_reference_fixture_and_oracle(dev):
    from superpoint_transformer_amd import transforms as T
    g = load_golden("horizontal_edge_features.npz")
    ins = _inputs(g)
    e = ins[0].shape[1]
    ei, attr = T.horizontal_edge_features(*[t.to(dev) for t in ins], add_self_loops=True)
    rei, rattr = O.horizontal_edge_features(*ins, add_self_loops=True)
    assert torch.equal(ei.cpu(), rei)                                   # indices bit-exact
    torch.testing.assert_close(attr.cpu(), rattr, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(attr.cpu()[:2 * e], torch.from_numpy(g["edge_attr"]),
                               rtol=1e-6, atol=1e-7)                    # the reference's own output
    ei2, attr2 = T.horizontal_edge_features(*[t.to(dev) for t in ins], add_self_loops=False)
    assert torch.equal(ei2.cpu(), torch.from_numpy(g["edge_index"]))
    assert torch.equal(attr2, attr[:2 * e])


@pytest.mark.gpu
def test_fused_kernel_degenerate_edges(dev):
    """Zero offsets / coincident centroids: 0/0 directions become 0 like the
    reference's NaN clean-up (graph.py:1206-1208, 1254-1256)."""
    from superpoint_transformer_amd import transforms as T
    n = 5
    pos = torch.tensor([[0., 0, 0], [0, 0, 0], [1, 2, 3], [1, 2, 3], [4, 4, 4]])
    normal = torch.nn.functional.normalize(torch.randn(n, 3, generator=torch.Generator().manual_seed(1)), dim=1)
    se = torch.tensor([[0, 2, 0], [1, 3, 4]])
    ea = torch.zeros(3, 7)
    ea[2, :3] = torch.tensor([1.0, -2.0, 0.5])
    ll = torch.arange(n).float().view(-1, 1)
    args = (se, ea, pos, normal, ll, ll * 2, ll * 3, ll * 4)
    ei, attr = T.horizontal_edge_features(*[t.to(dev) for t in args])
    rei, rattr = O.horizontal_edge_features(*args)
    assert torch.equal(ei.cpu(), rei)
    assert not attr.isnan().any()
    torch.testing.assert_close(attr.cpu(), rattr, rtol=1e-6, atol=1e-7)
    # empty graph with self loops only
    ei, attr = T.horizontal_edge_features(se[:, :0].to(dev), ea[:0].to(dev), *[t.to(dev) for t in args[2:]])
    assert ei.shape == (2, n) and attr.abs().sum() == 0


def _vertical_inputs(g):
    def lv(pre):
        return {k[len(pre):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pre)}
    return lv("child__"), lv("parent__")


def test_oracle_vertical_features_match_reference_function_output():
    """graph.py:1335-1416 executed from the reference's source on levels 1 -> 2 of the demo room
    (one child moved onto its parent's centroid: the 0/0 direction)."""
    g = load_golden("vertical_edge_features.npz")
    c, p = _vertical_inputs(g)
    logs = ("log_length", "log_surface", "log_volume", "log_size")
    out = O.vertical_edge_features(c["pos"], c["normal"], [c[k] for k in logs], p["pos"],
                                   p["normal"], [p[k] for k in logs], c["super_index"])
    torch.testing.assert_close(out, torch.from_numpy(g["v_edge_attr"]), rtol=1e-6, atol=1e-7)
    assert out.shape[1] == 9 and not out.isnan().any()


@pytest.mark.gpu
def test_vertical_edge_feature_kernel_matches_reference_fixture(dev):
    from superpoint_transformer_amd import transforms as T
    g = load_golden("vertical_edge_features.npz")
    c, p = _vertical_inputs(g)
    out = T.vertical_edge_features({k: v.to(dev) for k, v in c.items()},
                                   {k: v.to(dev) for k, v in p.items()})
    torch.testing.assert_close(out.cpu(), torch.from_numpy(g["v_edge_attr"]), rtol=1e-6, atol=1e-7)
    assert float(out[3, :3].abs().sum()) == 0.0          # the degenerate child: direction 0


def test_sampling_weights_oracle_matches_the_reference():
    """sampling.py:771-798 (fixture: the weights the reference's own SampleSegments._process
    hands to torch.multinomial, tests/golden/make_golden_sampling_weights.py)."""
    import numpy as np
    from conftest import load_golden
    from oracle import spt_oracle as O
    g = load_golden("sampling_weights.npz")
    size = np.bincount(g["super_index"], minlength=g["y"].shape[0])
    for by_size in (False, True):
        for by_class in (False, True):
            w = O.segment_sampling_weights(size, g["y"], by_size, by_class)
            np.testing.assert_allclose(w, g[f"w_{int(by_size)}{int(by_class)}"], rtol=1e-6)
    assert np.abs(g["w_11"] - g["w_00"]).max() > 1e-5            # the terms do something


def test_oversampling_reproduces_the_reference_draw_for_draw():
    """neighbors.py:420-488: the port consumes the generator like the reference (one uniform
    per missing entry, row-major), so under the fixture's seed the CPU result is identical to
    what the reference's own function returned (tests/golden/make_golden_oversample.py)."""
    from conftest import load_golden
    from superpoint_transformer_amd.neighbors import oversample_partial_neighborhoods
    g = load_golden("oversample.npz")
    nb, d = torch.from_numpy(g["neighbors"]).clone(), torch.from_numpy(g["distances"]).clone()
    torch.manual_seed(int(g["seed"]))
    out_nb, out_d = oversample_partial_neighborhoods(nb, d, int(g["k"]))
    assert torch.equal(out_nb, torch.from_numpy(g["out_neighbors"]))
    assert torch.equal(out_d, torch.from_numpy(g["out_distances"]))
    empty = (torch.from_numpy(g["neighbors"]) >= 0).sum(1) == 0
    assert bool((out_nb[empty] == -1).all()) and bool((out_nb[~empty] >= 0).all())


def test_subgraph_seeds_reproduce_the_reference_draw_for_draw():
    """sampling.py:922-948: seeds spread over the clouds of the batch (shuffled cloud order,
    k // clouds each, the remainder to the last) or drawn from the whole level.  The port
    consumes the generator like the reference, so under the fixture's seed the CPU draw equals
    what the reference's own ``BaseSampleSubgraphs._process`` drew
    (tests/golden/make_golden_seeds.py); the weights are the oracle's (pinned separately)."""
    import numpy as np
    from conftest import load_golden
    from oracle import spt_oracle as O
    from superpoint_transformer_amd.data import Data
    from superpoint_transformer_amd.transforms import SampleRadiusSubgraphs
    g = load_golden("subgraph_seeds.npz")
    batch = torch.from_numpy(g["batch"])
    size = np.bincount(g["super_index"], minlength=batch.numel())
    for tag in "abc":
        k, use_batch, by_size, by_class = (int(v) for v in g[f"{tag}_cfg"])
        w = torch.from_numpy(O.segment_sampling_weights(size, g["y"], bool(by_size), bool(by_class)))
        t = SampleRadiusSubgraphs(k=k, use_batch=bool(use_batch))
        data = Data(pos=torch.zeros(batch.numel(), 3), batch=batch)
        torch.manual_seed(int(g["seed"]))
        seeds = t._seeds(data, k, w)
        assert torch.equal(seeds, torch.from_numpy(g[f"{tag}_seeds"])), tag


def test_restrict_size_keeps_the_edges_the_reference_keeps():
    """sampling.py:1405-1423, edge branch, draw for draw under the fixture's seed (CPU
    generator; tests/golden/make_golden_restrict.py) - and the level spellings of
    src/utils/list.py:46-91."""
    from conftest import load_golden
    from superpoint_transformer_amd.data import NAG, Data
    from superpoint_transformer_amd.transforms import NAGRestrictSize, _per_level
    g = load_golden("restrict_size.npz")
    n = int(g["num_nodes"])
    lvl = Data(pos=torch.zeros(n, 3), edge_index=torch.from_numpy(g["in_edge_index"]),
               edge_attr=torch.from_numpy(g["in_edge_attr"]), edge_w=torch.from_numpy(g["in_edge_w"]))
    nag = NAG([Data(pos=torch.zeros(5, 3)), lvl])
    torch.manual_seed(int(g["seed"]))
    out = NAGRestrictSize(level="1+", num_nodes=0, num_edges=int(g["num_edges"]))(nag)
    for k in ("edge_index", "edge_attr", "edge_w"):
        assert torch.equal(out[1][k], torch.from_numpy(g["out_" + k])), k
    assert out[1].num_edges == 700 and out[0].num_nodes == 5
    assert _per_level("1+", -1, 9, 4) == [-1, 9, 9, 9] and _per_level("2-", -1, 9, 4) == [9, 9, -1, -1]
    assert _per_level("all", -1, 9, 3, start=1) == [-1, 9, 9] and _per_level(2, 0, 9, 3) == [0, 0, 9]
    with pytest.raises(ValueError):
        _per_level("1x", 0, 1, 3)
