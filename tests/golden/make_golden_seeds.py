"""Golden fixture for the seed draw of the subgraph samplers (src/transforms/sampling.py:872-948),
produced by the REFERENCE'S OWN ``BaseSampleSubgraphs._process`` cut out of its file with ``ast``
- unmodified - under ``torch.manual_seed(SEED)`` on the CPU generator, on a duck-typed NAG of
three clouds; ``_sample_subgraphs_from_seeds`` (abstract in the base class) records the seeds it
is handed and stops the run.  The ``torch`` the function sees is a proxy that also records the
weights given to each ``multinomial`` call.

Usage (build container only): python tests/golden/make_golden_seeds.py
"""
import ast
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF
SEED = 99


class Captured(Exception):
    pass


def main():
    tree = ast.parse(open(os.path.join(REF, "src/transforms/sampling.py")).read())
    cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BaseSampleSubgraphs")
    fn = next(n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name == "_process")
    ns = {"torch": torch, "NAGBatch": None}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "sampling.py", "exec"), ns)

    gen = torch.Generator().manual_seed(5)
    n1, n0, nc = 400, 9000, 6
    batch = torch.sort(torch.randint(0, 3, (n1,), generator=gen)).values
    si = torch.randint(0, n1, (n0,), generator=gen)
    si[:n1] = torch.arange(n1)
    size = torch.bincount(si, minlength=n1)
    hist = torch.randint(0, 50, (n1, nc + 1), generator=gen) * (torch.rand(n1, nc + 1, generator=gen) < 0.4)

    class Level:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class Nag:
        start_i_level, end_i_level, absolute_num_levels, device = 0, 1, 2, torch.device("cpu")

        def __getitem__(self, i):
            return [Level(num_nodes=n0, y=None, batch=None), Level(num_nodes=n1, y=hist, batch=batch)][i]

        def get_sub_size(self, i_level, low=0):
            return size

    out = dict(batch=batch, super_index=si, y=hist, seed=SEED)
    seen = {}

    def capture(nag, i_level, idx_seed):
        seen["idx"] = idx_seed.clone()
        raise Captured

    for tag, (k, use_batch, by_size, by_class) in {
            "a": (7, True, True, True), "b": (2, True, False, False), "c": (5, False, True, False)}.items():
        t = types.SimpleNamespace(i_level=1, k=k, by_size=by_size, by_class=by_class,
                                  use_batch=use_batch, disjoint=False,
                                  _sample_subgraphs_from_seeds=capture)
        torch.manual_seed(SEED)
        try:
            ns["_process"](t, Nag())
        except Captured:
            pass
        out[f"{tag}_seeds"] = seen["idx"]
        out[f"{tag}_cfg"] = torch.tensor([k, int(use_batch), int(by_size), int(by_class)])
    mg.save("subgraph_seeds.npz", **out)


if __name__ == "__main__":
    main()
