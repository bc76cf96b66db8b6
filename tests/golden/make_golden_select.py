"""Golden fixture for NAG selection / re-indexing, produced by the
REFERENCE'S OWN code:

  * ``src/data/csr.py`` and ``src/data/cluster.py`` are imported verbatim by path
    (CSRData.select / index_select_pointers, Cluster.select, Cluster.to_super_index);
  * ``Data.select`` (src/data/data.py:286-470) and ``NAG.select``
    (src/data/nag.py:306-399) cannot be imported (their modules subclass
    torch_geometric's Data and pull h5py / metrics at import time), so their
    FunctionDefs are cut out of the files with ``ast`` - unmodified - and executed as
    methods of duck-typed attribute stores.

Stand-ins: h5py (inert), torch_geometric.data.storage.recursive_apply(_),
consecutive_cluster = oracle restatement of the published PyG function.

Usage (build container only): python tests/golden/make_golden_select.py
"""
import ast
import copy
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

REF = mg.REF


def cut(path, cls, name):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    return ast.Module(body=[fn], type_ignores=[])


def load_reference():
    U, _ = mg.install_reference_import_hooks()
    h5 = types.ModuleType("h5py")
    h5.File = h5.Group = h5.Dataset = type("Inert", (), {})          # annotations only
    sys.modules["h5py"] = h5
    st = types.ModuleType("torch_geometric.data.storage")

    def recursive_apply(x, f):
        if torch.is_tensor(x):
            return f(x)
        if isinstance(x, (list, tuple)):
            return type(x)(recursive_apply(v, f) for v in x)
        if isinstance(x, dict):
            return {k: recursive_apply(v, f) for k, v in x.items()}
        return x.apply(f) if hasattr(x, "apply") else x
    st.recursive_apply = recursive_apply
    st.recursive_apply_ = lambda x, f: recursive_apply(x, f)
    tgd = types.ModuleType("torch_geometric.data")
    tgd.storage = st
    sys.modules["torch_geometric.data"] = tgd
    sys.modules["torch_geometric.data.storage"] = st
    sys.modules["torch_geometric.nn.pool.consecutive"].consecutive_cluster = O.consecutive_cluster
    mem = importlib.import_module("src.utils.memory")
    U.human_readable_memory = getattr(mem, "human_readable_memory", None)
    for name in ("save_tensor", "load_tensor", "from_flat_tensor", "check_incremental_keys"):
        if not hasattr(U, name):
            setattr(U, name, None)
    pkg = types.ModuleType("src.data")
    pkg.__path__ = [os.path.join(REF, "src", "data")]
    sys.modules["src.data"] = pkg
    csr = importlib.import_module("src.data.csr")
    cluster = importlib.import_module("src.data.cluster")
    return U, csr, cluster


class DuckData:
    """Attribute store with the handful of properties Data.select touches."""

    def __init__(self, **kw):
        object.__setattr__(self, "_s", {})
        object.__setattr__(self, "_n", None)
        for k, v in kw.items():
            setattr(self, k, v)

    def __getattr__(self, k):
        s = object.__getattribute__(self, "_s")
        if k in s:
            return s[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        if k == "num_nodes":
            object.__setattr__(self, "_n", v)
        else:
            self._s[k] = v

    __getitem__ = __getattr__
    __setitem__ = __setattr__

    def __iter__(self):
        return iter(list(self._s.items()))

    keys = property(lambda self: list(self._s))
    device = property(lambda self: torch.device("cpu"))
    has_edges = property(lambda self: "edge_index" in self._s)
    num_edges = property(lambda self: self.edge_index.shape[1] if self.has_edges else 0)
    is_super = property(lambda self: "sub" in self._s)
    is_sub = property(lambda self: "super_index" in self._s)
    edge_keys = property(lambda self: [k for k in self._s if k.startswith("edge_") and k not in ("edge_index", "edge_attr")])
    v_edge_keys = property(lambda self: [k for k in self._s if k.startswith("v_edge_")])

    @property
    def num_nodes(self):
        if self._n is not None:
            return self._n
        for k in ("pos", "x", "super_index"):
            if k in self._s:
                return self._s[k].shape[0]
        return self._s["sub"].num_clusters

    def clone(self):
        return copy.deepcopy(self)


class DuckNAG:
    def __init__(self, data_list, start_i_level=0):
        self._list = data_list
        self.start_i_level = start_i_level

    def __getitem__(self, i):
        return self._list[i]

    absolute_num_levels = property(lambda self: len(self._list))
    num_points = property(lambda self: [d.num_nodes for d in self._list])
    device = property(lambda self: torch.device("cpu"))

    def assert_level_in_nag(self, i):
        assert 0 <= i < len(self._list)

    def clone(self):
        return copy.deepcopy(self)


def synth_nag(gen, Cluster, n0=4000, n1=300, n2=40):
    sizes = [n0, n1, n2]
    levels = []
    supers = []
    for lo, hi in ((n0, n1), (n1, n2)):
        si = torch.randint(0, hi, (lo,), generator=gen)
        si[:hi] = torch.randperm(hi, generator=gen)          # every cluster non-empty
        supers.append(si[torch.randperm(lo, generator=gen)])
    for l, n in enumerate(sizes):
        d = DuckData(pos=torch.randn(n, 3, generator=gen).float(),
                     x=torch.randn(n, 4, generator=gen).float())
        if l < 2:
            d.super_index = supers[l]
        if l > 0:
            d.sub = Cluster(supers[l - 1], torch.arange(sizes[l - 1]), dense=True)
            m = n * 6
            s = torch.randint(0, n, (m,), generator=gen)
            t = torch.randint(0, n, (m,), generator=gen)
            keep = s != t
            d.edge_index = torch.stack([s[keep], t[keep]])
            d.edge_attr = torch.randn(int(keep.sum()), 7, generator=gen).float()
            d.edge_w = torch.rand(int(keep.sum()), generator=gen).float()
            d.v_edge_attr = torch.randn(n, 2, generator=gen).float()
        levels.append(d)
    return DuckNAG(levels)


def dump(prefix, nag, out):
    for l, d in enumerate(nag._list):
        for k, v in d:
            if torch.is_tensor(v):
                out[f"{prefix}_L{l}_{k}"] = v
            else:                                            # Cluster
                out[f"{prefix}_L{l}_{k}_pointers"] = v.pointers
                out[f"{prefix}_L{l}_{k}_points"] = v.points


def main():
    U, csr, cluster = load_reference()
    ns = {"torch": torch, "np": np, "copy": copy, "tensor_idx": U.tensor_idx, "is_arange": U.is_arange,
          "has_duplicates": U.has_duplicates, "src": sys.modules["src"], "CSRData": csr.CSRData,
          "Cluster": cluster.Cluster, "consecutive_cluster": O.consecutive_cluster,
          "Data": DuckData, "NAG": DuckNAG, "Union": None, "List": None, "Tuple": None}
    exec(compile(cut("src/data/data.py", "Data", "select"), "data.py", "exec"), ns)
    DuckData.select = ns["select"]
    exec(compile(cut("src/data/nag.py", "NAG", "select"), "nag.py", "exec"), ns)
    DuckNAG.select = ns["select"]

    gen = torch.Generator().manual_seed(99)
    nag = synth_nag(gen, cluster.Cluster)
    out = {}
    dump("in", nag, out)
    n = nag.num_points
    picks = {
        0: torch.randperm(n[0], generator=gen)[:1500],                 # shuffled subset of points
        1: torch.randperm(n[1], generator=gen)[:120],                  # shuffled subset of segments
        2: torch.sort(torch.randperm(n[2], generator=gen)[:9])[0],     # sorted subset at the top
    }
    for lvl, idx in picks.items():
        sel = nag.select(lvl, idx)
        out[f"sel{lvl}_idx"] = idx
        dump(f"sel{lvl}", sel, out)
        print(f"select(level {lvl}, {idx.numel()} nodes) -> {sel.num_points}")
    # Cluster.select in isolation (level 1's cluster), with a shuffled idx
    c = nag[1].sub
    idx = torch.randperm(c.num_clusters, generator=gen)[:77]
    c2, (idx_sub, sub_super) = c.select(idx)
    out.update(cl_idx=idx, cl_pointers=c2.pointers, cl_points=c2.points, cl_idx_sub=idx_sub,
               cl_sub_super=sub_super)
    mg.save("nag_select.npz", **out)


if __name__ == "__main__":
    main()
