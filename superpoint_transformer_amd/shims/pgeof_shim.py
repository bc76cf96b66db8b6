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
    """Per point the neighbourhood size (k_min_search, then every ``k_step``, up to all of its
    list) of lowest eigenentropy.  pgeof's C++ is not available to pin against: the search
    follows the reference's own torch restatement of it (src/utils/geometry.py:248-287), the
    CSR lists spread to a -1-padded [N, longest] table."""
    dev = torch.device("cuda", torch.cuda.current_device())
    p = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32)).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(nn).astype(np.int64)).to(dev)
    ptr = torch.from_numpy(np.ascontiguousarray(nn_ptr).astype(np.int64)).to(dev)
    n = ptr.numel() - 1
    sizes = ptr[1:] - ptr[:-1]
    width = int(sizes.max()) if n else 0
    row = torch.arange(n, device=dev).repeat_interleave(sizes)
    dense = torch.full((n, max(width, 1)), -1, dtype=torch.int64, device=dev)
    dense[row, torch.arange(v.numel(), device=dev) - ptr[:-1][row]] = v
    f = _nb.geometric_features(p, dense, k_min=int(k_min), add_self_as_neighbor=False, raw=True,
                               k_step=int(k_step), k_min_search=int(k_min_search))
    return f.cpu().numpy()
