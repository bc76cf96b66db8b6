"""Golden fixture for ``subedges()`` (src/utils/graph.py:99-463), produced by
the REFERENCE'S OWN function: src/utils/{graph,edge,sparse,scatter,geometry,tensor}.py are
imported verbatim by path on the hooks of make_golden.py / make_golden_cluster_graph.py
(torch_scatter, PyG coalesce / remove_self_loops / consecutive_cluster = oracle restatements).

Scene: segments of planar-ish patches with real neighbours (adjacent patches), so that the
half-space and bounding-box filters, the top-k selection and the ordering along the first
principal component all do something.  Saved: inputs, the trimmed edge_index, ST_pairs, ST_uid.

Usage (build container only): python tests/golden/make_golden_subedges.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import spt_oracle as O  # noqa: E402


def load_reference_graph():
    U, _ = mg.install_reference_import_hooks()
    tgu = sys.modules["torch_geometric.utils"]
    tgu.coalesce = O.coalesce
    tgu.remove_self_loops = O.remove_self_loops
    sys.modules["torch_geometric.nn.pool.consecutive"].consecutive_cluster = O.consecutive_cluster
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


def scene(gen, grid=5, per=(60, 260)):
    """grid x grid tiles of a wavy floor + a few wall patches; one segment per tile."""
    pts, idx = [], []
    s = 0
    for i in range(grid):
        for j in range(grid):
            n = int(torch.randint(per[0], per[1], (1,), generator=gen))
            xy = torch.rand(n, 2, generator=gen) + torch.tensor([float(i), float(j)])
            z = 0.05 * torch.sin(3 * xy[:, 0]) + 0.02 * torch.randn(n, generator=gen)
            pts.append(torch.cat([xy, z.view(-1, 1)], dim=1))
            idx.append(torch.full((n,), s))
            s += 1
    for w in range(4):                                  # walls standing on the floor
        n = int(torch.randint(per[0], per[1], (1,), generator=gen))
        y = torch.rand(n, generator=gen) * 2 + w
        zz = torch.rand(n, generator=gen) * 1.5
        x = torch.full((n,), 1.0 + w) + 0.01 * torch.randn(n, generator=gen)
        pts.append(torch.stack([x, y, zz], dim=1))
        idx.append(torch.full((n,), s))
        s += 1
    pos = torch.cat(pts).float()
    index = torch.cat(idx)
    perm = torch.randperm(pos.shape[0], generator=gen)
    return pos[perm], index[perm], s


def main():
    graph = load_reference_graph()
    gen = torch.Generator().manual_seed(777)
    pos, index, nseg = scene(gen)
    # segment graph: pairs of segments whose bounding boxes come within 0.15 of each other,
    # given in both directions and with duplicates (subedges() trims them itself)
    lo = torch.stack([pos[index == s].min(0).values for s in range(nseg)])
    hi = torch.stack([pos[index == s].max(0).values for s in range(nseg)])
    gap = (torch.maximum(lo[:, None], lo[None]) - torch.minimum(hi[:, None], hi[None])).clamp(min=0)
    near = (gap.norm(dim=2) < 0.15) & ~torch.eye(nseg, dtype=torch.bool)
    ei = near.nonzero().t().contiguous()
    ei = torch.cat([ei, ei[:, :7]], dim=1)
    out = {}
    for c, kw in enumerate((dict(ratio=0.2, k_min=20, cycles=3, margin=0.2),
                            dict(ratio=0.5, k_min=5, cycles=2, margin=0.05))):
        e2, pairs, uid = graph.subedges(pos.clone(), index.clone(), ei.clone(), **kw)
        out[f"c{c}_edge_index"], out[f"c{c}_pairs"], out[f"c{c}_uid"] = e2, pairs, uid
        out[f"c{c}_cfg"] = np.asarray([kw["ratio"], kw["k_min"], kw["cycles"], kw["margin"]])
        print(f"case {c}: {e2.shape[1]} trimmed edges, {pairs.shape[1]} subedge pairs")
    mg.save("subedges.npz", pos=pos, index=index, edge_index=ei, **out)


if __name__ == "__main__":
    main()
