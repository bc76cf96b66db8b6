"""CPU: the oracle's restatement of cluster_radius_nn_graph / scatter_nearest_neighbor
against the fixture produced by the reference's own functions
(tests/golden/make_golden_cluster_graph.py)."""
import os

import numpy as np
import torch

from oracle import spt_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cluster_graph.npz"))


def t(name):
    return torch.from_numpy(G[name])


def test_cluster_radius_nn_graph_matches_the_reference():
    for c in range(3):
        k_max, gap, trim, cycles = G[f"c{c}_cfg"].tolist()
        batch = t(f"c{c}_batch") if f"c{c}_batch" in G.files else None
        ei, d, _ = O.cluster_radius_nn_graph(t(f"c{c}_pos"), t(f"c{c}_idx"), int(k_max), gap,
                                             batch, bool(trim), int(cycles))
        assert torch.equal(ei, t(f"c{c}_edge_index")), c
        assert torch.allclose(d, t(f"c{c}_dist"), atol=0, rtol=1e-6), c


def test_scatter_nearest_neighbor_matches_the_reference():
    for c in range(3):
        cycles = int(G[f"c{c}_cfg"][3])
        got = O.scatter_nearest_neighbor(t(f"c{c}_pos"), t(f"c{c}_idx"), t(f"c{c}_snn_edges"), cycles)
        assert torch.equal(got, t(f"c{c}_snn_idx")), c
