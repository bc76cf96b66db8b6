"""Golden fixture for the cluster radius graph, produced by the
REFERENCE'S OWN ``cluster_radius_nn_graph`` / ``scatter_nearest_neighbor`` /
``to_trimmed`` / ``edge_wise_points`` (src/utils/neighbors.py, scatter.py, graph.py,
edge.py imported verbatim by path).

Stand-ins for what cannot run here (all "[third-party restated]" in
oracle/spt_oracle.py): torch_scatter, torch_geometric's coalesce /
remove_self_loops / consecutive_cluster, and the FRNN CUDA search (exhaustive
float32 search with FRNN's contract).  ``Tensor.cuda`` is made a no-op for the call
because knn_1 moves CPU inputs to the GPU (neighbors.py:84-86).

Usage (build container only): python tests/golden/make_golden_cluster_graph.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import spt_oracle as O  # noqa: E402


def frnn_stand_in(points1, points2, K=None, r=None, **kw):
    d, i = O.frnn_grid_points(points1[0], points2[0], int(K.view(-1)[0]), float(r.view(-1)[0]))
    return d.unsqueeze(0), i.unsqueeze(0), None, None


def scene(gen, nseg, lo, hi, spread):
    sizes = torch.randint(lo, hi, (nseg,), generator=gen)
    idx = torch.repeat_interleave(torch.arange(nseg), sizes)
    idx = idx[torch.randperm(idx.numel(), generator=gen)]
    centre = torch.rand(nseg, 3, generator=gen) * torch.tensor([9.0, 9.0, 3.0])
    ext = torch.rand(nseg, 3, generator=gen) * spread + 0.05
    pos = centre[idx] + (torch.rand(idx.numel(), 3, generator=gen) - 0.5) * ext[idx]
    return pos.float(), idx


def main():
    U, _ = mg.install_reference_import_hooks()
    tgu = sys.modules["torch_geometric.utils"]
    tgu.coalesce = O.coalesce
    tgu.remove_self_loops = O.remove_self_loops
    sys.modules["torch_geometric.nn.pool.consecutive"].consecutive_cluster = O.consecutive_cluster
    for name in ("scatter", "neighbors", "edge"):
        m = sys.modules.get(f"src.utils.{name}") or importlib.import_module(f"src.utils.{name}")
        if hasattr(m, "coalesce"):
            m.coalesce = O.coalesce
        if hasattr(m, "consecutive_cluster"):
            m.consecutive_cluster = O.consecutive_cluster
    edge = importlib.import_module("src.utils.edge")
    edge.consecutive_cluster = O.consecutive_cluster
    U.edge_wise_points = edge.edge_wise_points
    sys.modules["src.utils.scatter"].edge_wise_points = edge.edge_wise_points
    graph = importlib.import_module("src.utils.graph")
    U.to_trimmed = graph.to_trimmed
    nbm = sys.modules["src.utils.neighbors"]
    nbm.frnn.frnn_grid_points = frnn_stand_in
    torch.Tensor.cuda = lambda self, *a, **k: self

    gen = torch.Generator().manual_seed(4242)
    out = {}
    cases = [dict(nseg=120, lo=1, hi=40, spread=2.5, k_max=12, gap=0.4, trim=True, cycles=3),
             dict(nseg=80, lo=5, hi=150, spread=4.0, k_max=30, gap=1.0, trim=True, cycles=3),
             dict(nseg=60, lo=2, hi=30, spread=1.5, k_max=8, gap=0.8, trim=False, cycles=2)]
    for c, cfg in enumerate(cases):
        pos, idx = scene(gen, cfg["nseg"], cfg["lo"], cfg["hi"], cfg["spread"])
        batch = (torch.arange(cfg["nseg"]) >= cfg["nseg"] // 2).long() if c == 1 else None
        ei, d = nbm.cluster_radius_nn_graph(
            pos, idx, k_max=cfg["k_max"], gap=cfg["gap"], batch=batch, trim=cfg["trim"],
            cycles=cfg["cycles"], chunk_size=37 if c == 0 else None)
        out[f"c{c}_pos"], out[f"c{c}_idx"] = pos, idx
        out[f"c{c}_edge_index"], out[f"c{c}_dist"] = ei, d
        if batch is not None:
            out[f"c{c}_batch"] = batch
        out[f"c{c}_cfg"] = np.asarray([cfg["k_max"], cfg["gap"], int(cfg["trim"]), cfg["cycles"]],
                                      dtype=np.float64)
        # scatter_nearest_neighbor alone, on a hand-made coalesced graph
        s = torch.randint(0, cfg["nseg"], (300,), generator=gen)
        t = torch.randint(0, cfg["nseg"], (300,), generator=gen)
        g = O.coalesce(torch.stack([s, t])[:, s != t])
        cand, cidx = U.scatter_nearest_neighbor(pos, idx, g, cycles=cfg["cycles"], chunk_size=None)
        out[f"c{c}_snn_edges"], out[f"c{c}_snn_idx"] = g, cidx
        print(f"case {c}: {pos.shape[0]} points, {cfg['nseg']} clusters -> {ei.shape[1]} edges")
    mg.save("cluster_graph.npz", **out)


if __name__ == "__main__":
    main()
