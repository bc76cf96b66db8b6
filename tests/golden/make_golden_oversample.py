"""Golden fixture for ``oversample_partial_neighborhoods`` (src/utils/neighbors.py:420-488),
produced by the REFERENCE'S OWN function (module imported verbatim on the hooks of
make_golden.py) under ``torch.manual_seed(SEED)`` on the CPU generator: a table of sorted
neighbourhoods with 0..k found entries.  The port draws the same number of uniforms in the same
order, so under the same seed it must reproduce the table exactly.

Usage (build container only): python tests/golden/make_golden_oversample.py
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SEED = 4321


def main():
    mg.install_reference_import_hooks()
    NB = importlib.import_module("src.utils.neighbors")
    gen = torch.Generator().manual_seed(8)
    n, k = 3000, 14
    found = torch.randint(0, k + 1, (n,), generator=gen)
    nb = torch.randint(0, n, (n, k), generator=gen)
    d = torch.rand(n, k, generator=gen).sort(dim=1).values
    missing = torch.arange(k).view(1, -1) >= found.view(-1, 1)
    nb[missing], d[missing] = -1, -1
    torch.manual_seed(SEED)
    out_nb, out_d = NB.oversample_partial_neighborhoods(nb.clone(), d.clone(), k)
    mg.save("oversample.npz", neighbors=nb, distances=d, k=k, seed=SEED,
            out_neighbors=out_nb, out_distances=out_d)


if __name__ == "__main__":
    main()
