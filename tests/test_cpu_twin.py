"""oracle/cpu/libspt_cpu.so (OpenMP CPU twins of the kNN / geometric-feature entries) against the
exhaustive Python oracle: the twin is what carries the bit-exact kNN check to full DALES / S3DIS
size on the GPU box (tests/test_fullsize_gpu.py) and bench.py's preprocess cpu_baseline, so it is
pinned here first - indices AND squared distances bit for bit, lattice ties, both radius
conventions, K beyond what the radius holds, queries outside the search cloud's box."""
import pytest
import torch

from oracle import cpu_twin as T
from oracle import spt_oracle as O


def _cloud(kind, n, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        return torch.rand(n, 3, generator=g)
    if kind == "lattice":                       # many exactly equal distances: the index tie rule
        return torch.randint(0, 12, (n, 3), generator=g).float() * 0.05
    if kind == "plane":                         # a voxelised surface, like the preprocessing input
        p = torch.rand(n, 3, generator=g) * torch.tensor([4.0, 4.0, 0.02])
        return (p / 0.03).round() * 0.03
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n,k,r", [("uniform", 3000, 8, 0.3), ("uniform", 1500, 40, 0.15),
                                        ("lattice", 2500, 45, 0.2), ("plane", 4000, 25, 0.5),
                                        ("plane", 3000, 46, 10.0)])
@pytest.mark.parametrize("inclusive", [False, True])
def test_grid_knn_twin_is_bit_exact_against_the_exhaustive_oracle(kind, n, k, r, inclusive):
    pos = _cloud(kind, n, 1)
    d, i = O.frnn_grid_points(pos, pos, k, r, strict=not inclusive)
    ii, dd = T.grid_knn(pos, pos, k, r, inclusive=inclusive)
    assert torch.equal(i, ii)
    assert torch.equal(d.float(), dd)
    # any cell size gives the same answer
    for cell in (r / 7.3, r * 2.1):
        i2, d2 = T.grid_knn(pos, pos, k, r, inclusive=inclusive, cell_size=cell)
        assert torch.equal(i2, ii) and torch.equal(d2, dd)


def test_grid_knn_twin_other_queries_and_missing_neighbours():
    search = _cloud("uniform", 2000, 2)
    query = torch.cat([_cloud("uniform", 500, 3), torch.tensor([[2.5, -1.0, 0.5], [-3.0, 9.0, 9.0]])])
    d, i = O.frnn_grid_points(query, search, 12, 0.12)
    ii, dd = T.grid_knn(query, search, 12, 0.12)
    assert torch.equal(i, ii) and torch.equal(d.float(), dd)
    assert (ii[-1] == -1).all() and (dd[-1] == -1).all()        # nothing within r of a far query
    assert (ii == -1).any()                                      # partial neighbourhoods exist
    _, de = T.grid_knn(query, search, 12, 0.12, squared=False)
    import numpy as np           # (torch's vectorised CPU sqrt is 1 ulp off in ~1 % of the values)
    assert np.array_equal(de[dd >= 0].numpy(), np.sqrt(dd[dd >= 0].numpy()))


@pytest.mark.parametrize("k_min", [1, 5])
def test_point_geof_twin_matches_the_oracle(k_min):
    pos = _cloud("plane", 3000, 5)
    nb, _ = O.knn_1(pos, 20, 0.12)                               # partial neighbourhoods (-1) included
    ref = O.geometric_features(pos.double(), nb, k_min=k_min)
    got = T.point_geof(pos, nb, k_min=k_min)
    # eigenvectors of (nearly) repeated eigenvalues are free: compare away from them, like
    # tests/test_neighbors_gpu.py does
    assert (ref[:, [0, 1, 2, 7, 8, 9, 10]].float() - got[:, [0, 1, 2, 7, 8, 9, 10]]).abs().max() < 1e-5
    ok = (ref[:, 1] > 0.05) & (ref[:, 0] + ref[:, 2] < 0.95)
    assert (ref[ok][:, 3].float() - got[ok][:, 3]).abs().max() < 1e-4
    # the normal is flipped to z >= 0: with nz = +-0 (a vertical plane of lattice points) the sign
    # is free - compare up to it
    a, b = ref[ok][:, 4:7].float(), got[ok][:, 4:7]
    assert torch.minimum((a - b).abs().max(dim=1).values, (a + b).abs().max(dim=1).values).max() < 1e-4
