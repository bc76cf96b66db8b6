"""Golden fixture for ``NAGRestrictSize._restrict_level`` (src/transforms/sampling.py:1405-1423),
edge branch: the REFERENCE'S OWN static method cut out of its file with ``ast`` - unmodified -
on a duck-typed level under ``torch.manual_seed(SEED)`` (CPU generator).  The port makes the
same ``torch.multinomial`` call, so it must keep the same edges in the same order.

Usage (build container only): python tests/golden/make_golden_restrict.py
"""
import ast
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SEED = 77


def main():
    tree = ast.parse(open(os.path.join(mg.REF, "src/transforms/sampling.py")).read())
    cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "NAGRestrictSize")
    fn = next(n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name == "_restrict_level")
    fn.decorator_list = []
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "sampling.py", "exec"), ns)

    gen = torch.Generator().manual_seed(3)
    n, e = 300, 2000

    class Level:
        def __init__(self):
            self.store = dict(edge_index=torch.randint(0, n, (2, e), generator=gen),
                              edge_attr=torch.randn(e, 5, generator=gen),
                              edge_w=torch.rand(e, generator=gen))
        num_nodes = n
        num_edges = property(lambda self: self.store["edge_index"].shape[1])
        has_edge_attr = True
        edge_keys = ["edge_w"]

        def __getattr__(self, k):
            return self.__dict__["store"][k]

        def __setattr__(self, k, v):
            if k == "store":
                object.__setattr__(self, k, v)
            else:
                self.store[k] = v

        __getitem__ = __getattr__
        __setitem__ = __setattr__

    class Nag:
        device = torch.device("cpu")

        def __init__(self):
            self.levels = [None, Level()]

        def __getitem__(self, i):
            return self.levels[i]

    nag = Nag()
    out = {k: v.clone() for k, v in nag[1].store.items()}
    torch.manual_seed(SEED)
    res = ns["_restrict_level"](nag, 1, -1, 700)
    mg.save("restrict_size.npz", seed=SEED, num_edges=700, num_nodes=n,
            **{"in_" + k: v for k, v in out.items()},
            **{"out_" + k: v for k, v in res[1].store.items()})


if __name__ == "__main__":
    main()
