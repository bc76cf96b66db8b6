"""GPU parity of the cluster radius graph through the C ABI.

Bars: edge_index BIT-EXACT against the reference's own output
(tests/golden/cluster_graph.npz) and the oracle, incl. the intermediate trimmed
graph; anchor point indices bit-exact (the distance expression is evaluated in f32
exactly like the reference: sqrt(dx*dx + dy*dy + dz*dz), ties -> lowest point
index); anchor distances to 1e-6 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu

G = load_golden("cluster_graph.npz")


def t(name):
    return torch.from_numpy(G[name])


@pytest.mark.parametrize("c", [0, 1, 2])
def test_graph_matches_the_reference(c, dev):
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    k_max, gap, trim, cycles = G[f"c{c}_cfg"].tolist()
    batch = t(f"c{c}_batch").to(dev) if f"c{c}_batch" in G else None
    ei, d = cluster_radius_nn_graph(t(f"c{c}_pos").to(dev), t(f"c{c}_idx").to(dev), int(k_max),
                                    gap, batch, bool(trim), int(cycles))
    assert torch.equal(ei.cpu(), t(f"c{c}_edge_index"))
    assert torch.allclose(d.cpu(), t(f"c{c}_dist"), atol=0, rtol=1e-6)


@pytest.mark.parametrize("c", [0, 1, 2])
def test_anchors_match_the_reference(c, dev):
    from superpoint_transformer_amd.neighbors import scatter_nearest_neighbor
    pos, idx, g = t(f"c{c}_pos"), t(f"c{c}_idx"), t(f"c{c}_snn_edges")
    cand, cidx = scatter_nearest_neighbor(pos.to(dev), idx.to(dev), g.to(dev),
                                          cycles=int(G[f"c{c}_cfg"][3]))
    assert torch.equal(cidx.cpu(), t(f"c{c}_snn_idx"))
    assert torch.equal(cand.cpu(), torch.vstack((pos[cidx[0].cpu()], pos[cidx[1].cpu()])))


@pytest.mark.parametrize("seed,nseg,hi,k_max,gap,trim", [
    (1, 300, 60, 20, 0.5, True), (2, 50, 400, 10, 0.3, True), (3, 500, 8, 30, 1.0, False),
    (4, 40, 30, 63, 2.0, True), (5, 200, 20, 5, 0.0, True),
    # k_max above one 64-neighbour search: chained continuation searches (reference default 100)
    (6, 400, 12, 100, 3.0, True), (7, 150, 25, 140, 6.0, False)])
def test_graph_and_intermediates_match_the_oracle(seed, nseg, hi, k_max, gap, trim, dev):
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(1, hi, (nseg,), generator=g)
    idx = torch.repeat_interleave(torch.arange(nseg), sizes)
    idx = idx[torch.randperm(idx.numel(), generator=g)]
    centre = torch.rand(nseg, 3, generator=g) * torch.tensor([12.0, 12.0, 3.0])
    pos = (centre[idx] + (torch.rand(idx.numel(), 3, generator=g) - 0.5) *
           (torch.rand(nseg, 3, generator=g) * 2.0 + 0.05)[idx]).float()
    ei, d, mid = cluster_radius_nn_graph(pos.to(dev), idx.to(dev), k_max, gap, None, trim, 3,
                                         return_intermediate=True)
    rei, rd, rmid = O.cluster_radius_nn_graph(pos, idx, k_max, gap, None, trim, 3)
    assert torch.equal(mid["trimmed"].cpu(), rmid["trimmed"])
    assert torch.equal(mid["anchors"].cpu(), rmid["anchors"])
    assert torch.equal(ei.cpu(), rei)
    assert torch.allclose(d.cpu(), rd, atol=0, rtol=1e-6)
    if trim:
        assert bool((ei[0] < ei[1]).all())
    key = ei[0] * nseg + ei[1]
    assert bool((key[1:] > key[:-1]).all())                 # sorted by (s, t), no duplicates


def test_euclidean_convention_switch(dev):
    """squared=False feeds Euclidean centre distances to the radius-sum filter."""
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    pos, idx = t("c0_pos"), t("c0_idx")
    ei, d, mid = cluster_radius_nn_graph(pos.to(dev), idx.to(dev), 12, 0.4, squared=False,
                                         return_intermediate=True)
    rei, rd, rmid = O.cluster_radius_nn_graph(pos, idx, 12, 0.4, squared=False)
    assert torch.equal(mid["trimmed"].cpu(), rmid["trimmed"])
    assert torch.equal(ei.cpu(), rei)


def test_default_k_max_on_the_reference_fixture(dev):
    """k_max = 100 (the reference's default, neighbors.py:491) on the fixture cloud: more
    neighbours than one search holds; the graph equals the oracle's."""
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    pos, idx = t("c0_pos"), t("c0_idx")
    ei, d = cluster_radius_nn_graph(pos.to(dev), idx.to(dev), k_max=100, gap=1.0)
    rei, rd, _ = O.cluster_radius_nn_graph(pos, idx, 100, 1.0)
    assert torch.equal(ei.cpu(), rei)
    assert torch.allclose(d.cpu(), rd, atol=0, rtol=1e-6)


def test_k_max_beyond_four_searches_is_refused(dev):
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    with pytest.raises(NotImplementedError):
        cluster_radius_nn_graph(t("c0_pos").to(dev), t("c0_idx").to(dev), k_max=300, gap=1.0)


def test_graph_at_scene_scale(dev):
    """3 M points in 85 714 clusters (level-1 shape of a 3 M-point tile): properties
    that do not need the oracle."""
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    g = torch.Generator(dev).manual_seed(0)
    nseg, n = 85_714, 3_000_000
    idx = torch.randint(0, nseg, (n,), device=dev, generator=g)
    centre = torch.rand(nseg, 3, device=dev, generator=g) * torch.tensor([120.0, 120.0, 3.0], device=dev)
    pos = centre[idx] + (torch.rand(n, 3, device=dev, generator=g) - 0.5) * 0.6
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    ei, d, mid = cluster_radius_nn_graph(pos, idx, 30, 0.3, return_intermediate=True)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"cluster_radius_nn_graph: {ev[0].elapsed_time(ev[1]):.1f} ms, "
          f"{mid['trimmed'].shape[1]} candidate edges -> {ei.shape[1]} edges")
    assert ei.shape[1] > 1000
    assert bool((ei[0] < ei[1]).all()) and bool((d <= 0.3).all())
    key = ei[0] * nseg + ei[1]
    assert bool((key[1:] > key[:-1]).all())
    a = mid["anchors"]
    tr = mid["trimmed"]
    assert torch.equal(idx[a[0]], tr[0]) and torch.equal(idx[a[1]], tr[1])
    dn = (pos[a[0]] - pos[a[1]]).norm(dim=1)
    assert torch.allclose(dn, mid["d_nn"], rtol=1e-5, atol=1e-7)


def test_graph_degenerate_inputs(dev):
    from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph
    # two far-apart clusters: no edge survives
    pos = torch.cat([torch.rand(50, 3), torch.rand(40, 3) + 100.0]).to(dev)
    idx = torch.cat([torch.zeros(50), torch.ones(40)]).long().to(dev)
    ei, d = cluster_radius_nn_graph(pos, idx, k_max=5, gap=0.5)
    assert ei.shape == (2, 0) and d.numel() == 0
    # two touching clusters: exactly the edge (0, 1)
    pos = torch.cat([torch.rand(50, 3), torch.rand(40, 3) + torch.tensor([1.0, 0.0, 0.0])]).to(dev)
    ei, d = cluster_radius_nn_graph(pos, idx, k_max=5, gap=0.5)
    assert ei.tolist() == [[0], [1]] and float(d[0]) <= 0.5
