"""``src.dependencies.FRNN.frnn`` (the un-vendored lxxue/FRNN CUDA extension):
``frnn_grid_points`` with the call convention of src/utils/neighbors.py:24-48."""
import torch

from .. import neighbors as _nb


def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=-1, r=-1, grid=None,
                     return_nn=False, return_sorted=True, radius_cell_ratio=2.0, **unused):
    """points1 [1,P1,3] queries, points2 [1,P2,3] search set, K and r 1-element
    tensors or numbers -> (dists [1,P1,K] squared, idxs [1,P1,K] int64 with -1
    padding, None, None)."""
    if points1.dim() != 3 or points1.shape[0] != 1 or points2.shape[0] != 1:
        raise NotImplementedError("HIP shim: batch size 1 (the reference's only use)")
    k = int(K.view(-1)[0]) if torch.is_tensor(K) else int(K)
    rr = float(r.view(-1)[0]) if torch.is_tensor(r) else float(r)
    q, s = points1[0], points2[0]
    same = points1.data_ptr() == points2.data_ptr() and points1.shape == points2.shape
    dist, idx = _nb.frnn_grid_points(q, q if same else s, k, rr, squared=True)
    return dist.unsqueeze(0), idx.unsqueeze(0), None, None
