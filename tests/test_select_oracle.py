"""CPU: the oracle's restatement of NAG / Data / Cluster selection against the
fixture produced by the reference's own code (tests/golden/make_golden_select.py)."""
import os

import numpy as np
import torch

from oracle import spt_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nag_select.npz"))


def levels_of(prefix):
    out = []
    for l in range(3):
        pre = f"{prefix}_L{l}_"
        d = {k[len(pre):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(pre)}
        if "sub_pointers" in d:
            d["sub"] = (d.pop("sub_pointers"), d.pop("sub_points"))
        out.append(d)
    return out


def canon(ptr, pts):
    """Points of each cluster in ascending order: the reference builds
    ``Cluster(index, arange, dense=True)`` with an UNSTABLE torch.sort
    (src/utils/sparse.py:33), so the order inside a cluster is unspecified."""
    seg = torch.repeat_interleave(torch.arange(ptr.numel() - 1), ptr[1:] - ptr[:-1])
    return pts[torch.argsort(seg * (int(pts.max()) + 1 if pts.numel() else 1) + pts)]


def same_level(a, b):
    assert set(a) == set(b), (set(a), set(b))
    for k in a:
        if k == "sub":
            assert torch.equal(a[k][0], b[k][0]), k
            assert torch.equal(canon(*a[k]), canon(*b[k])), k
        else:
            assert torch.equal(a[k], b[k]), k


def test_nag_select_matches_the_reference():
    levels = levels_of("in")
    for lvl in range(3):
        got = O.nag_select(levels, lvl, torch.from_numpy(G[f"sel{lvl}_idx"]))
        ref = levels_of(f"sel{lvl}")
        for a, b in zip(got, ref):
            same_level(a, b)


def test_cluster_select_matches_the_reference():
    sub = levels_of("in")[1]["sub"]
    (ptr, pts), (idx_sub, sub_super) = O.cluster_select(sub[0], sub[1], torch.from_numpy(G["cl_idx"]))
    for a, k in ((ptr, "cl_pointers"), (pts, "cl_points"), (idx_sub, "cl_idx_sub"),
                 (sub_super, "cl_sub_super")):
        assert torch.equal(a, torch.from_numpy(G[k])), k
