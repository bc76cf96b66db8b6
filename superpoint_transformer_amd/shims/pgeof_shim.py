"""``pgeof`` (un-vendored C++/nanobind CPU library): numpy in, numpy out like
upstream (src/utils/geometry.py:142-162), computed on the MI355X."""
import numpy as np
import torch

from .. import neighbors as _nb


def compute_features(xyz, nn, nn_ptr, k_min=1, verbose=False):
    """xyz f32[N,3], nn u32[M], nn_ptr u32[N+1] -> f32[N,11]
    [linearity, planarity, scattering, verticality, nx, ny, nz, length, surface,
    volume, curvature] (raw: no verticality scaling, no normal flip)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    p = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32)).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(nn).astype(np.int64)).to(dev)
    ptr = torch.from_numpy(np.ascontiguousarray(nn_ptr).astype(np.int64)).to(dev)
    f = _nb.geometric_features_csr(p, v, ptr, k_min=int(k_min), add_self=False, raw=True)
    return f.cpu().numpy()


def compute_features_optimal(xyz, nn, nn_ptr, k_min=1, k_step=1, k_min_search=1, verbose=False):
    raise NotImplementedError(
        "optimal-neighbourhood search (k_step > 0) is off in the dataset configs "
        "(k_step=-1, configs/datamodule/semantic/default.yaml:121-127) and not built")
