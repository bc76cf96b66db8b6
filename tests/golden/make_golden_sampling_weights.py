"""Golden fixture for the segment sampling weights (``by_size`` / ``by_class``,
src/transforms/sampling.py:771-798), produced by the REFERENCE'S OWN ``SampleSegments._process``
cut out of its file with ``ast`` - unmodified - and run on a duck-typed two-level NAG; the
``torch`` it sees is a proxy whose ``multinomial`` records the weights it is handed and stops
the run (the draw itself is random and not part of the contract).

Usage (build container only): python tests/golden/make_golden_sampling_weights.py
"""
import ast
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF


class Captured(Exception):
    pass


class TorchProxy:
    def __init__(self):
        self.weights = None

    def __getattr__(self, name):
        return getattr(torch, name)

    def multinomial(self, weights, num, replacement=False):
        self.weights = weights.clone()
        raise Captured


def main():
    tree = ast.parse(open(os.path.join(REF, "src/transforms/sampling.py")).read())
    cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SampleSegments")
    fn = next(n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name == "_process")
    proxy = TorchProxy()
    ns = {"torch": proxy}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "sampling.py", "exec"), ns)

    gen = torch.Generator().manual_seed(31)
    n0, n1, nc = 6000, 150, 7
    si = torch.randint(0, n1, (n0,), generator=gen)
    si[:n1] = torch.arange(n1)
    hist = torch.randint(0, 60, (n1, nc + 1), generator=gen) * (torch.rand(n1, nc + 1, generator=gen) < 0.35)
    hist[:, 4] = 0                                           # a class nobody holds
    size = torch.bincount(si, minlength=n1)

    class Level:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class Nag:
        start_i_level, end_i_level, device = 0, 1, torch.device("cpu")

        def __getitem__(self, i):
            return [Level(num_nodes=n0, y=None), Level(num_nodes=n1, y=hist)][i]

        def get_sub_size(self, i_level, low=0):
            return size

    out = dict(super_index=si, y=hist)
    import types
    for by_size in (False, True):
        for by_class in (False, True):
            t = types.SimpleNamespace(ratio=0.3, by_size=by_size, by_class=by_class)
            try:
                ns["_process"](t, Nag())
            except Captured:
                pass
            out[f"w_{int(by_size)}{int(by_class)}"] = proxy.weights
    mg.save("sampling_weights.npz", **out)


if __name__ == "__main__":
    main()
