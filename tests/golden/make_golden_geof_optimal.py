"""Golden fixture for the optimal-neighbourhood geometric features (k_step >= 0,
src/utils/geometry.py:248-338), produced by the REFERENCE'S OWN ``geometric_features_torch``
imported verbatim on the hooks of make_golden.py, run in float64 (default dtype switched like
make_golden.py does for scatter_pca's ``cov``): planar / linear / volumetric patches, 24
brute-force neighbours within a radius that leaves part of the neighbourhoods partial.

Usage (build container only): python tests/golden/make_golden_geof_optimal.py
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    U, _ = mg.install_reference_import_hooks()
    G = importlib.import_module("src.utils.geometry")
    gen = torch.Generator().manual_seed(99)
    npts, k = 600, 24
    base = mg.rnd(gen, npts, 3)
    base[:200, 2] *= 0.01
    base[200:350, 1:] *= 0.01
    xyz = base.float()
    nn_idx, _ = U.knn_brute_force(xyz, xyz, k + 1, r_max=0.6)
    nn_idx = nn_idx[:, 1:]
    nn_full = torch.cat((torch.arange(npts).view(-1, 1), nn_idx), dim=1)
    out = dict(xyz=xyz, nn=nn_idx)
    torch.set_default_dtype(torch.float64)
    for tag, (k_min, k_step, k_search) in {"a": (3, 4, 8), "b": (5, 1, 2), "c": (1, 7, 25)}.items():
        f = G.geometric_features_torch(xyz.double(), nn_full.clone(), k_min=k_min, k_step=k_step,
                                       k_min_search=k_search, chunk_size=None)
        f["verticality"] = f["verticality"] * 2          # geometry.py:121
        nrm = f["normal"].clone()
        nrm[nrm[:, 2] < 0] *= -1                         # geometry.py:124
        out[f"{tag}_feats"] = torch.cat(
            [f["linearity"], f["planarity"], f["scattering"], f["verticality"], nrm,
             f["length"], f["surface"], f["volume"], f["curvature"]], dim=1)
        out[f"{tag}_cfg"] = torch.tensor([k_min, k_step, k_search])
    torch.set_default_dtype(torch.float32)
    mg.save("geometric_features_optimal.npz", **out)


if __name__ == "__main__":
    main()
