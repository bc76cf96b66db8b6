"""Segment-level preprocessing on the device: the
per-segment sampler and the handcrafted segment features, on the same CSR views
and kernels as the training path.

Mirrors, with the reference's names and argument meaning:
  * ``sparse_sample``             src/utils/sparse.py:142-243
  * ``scatter_std``               torch_scatter.scatter_std as used by SegmentFeatures
  * ``scatter_mean_orientation``  src/utils/scatter.py:249-300
  * ``segment_features``          ``_compute_cluster_features``, src/transforms/graph.py:193-321
"""
import math

import torch

from . import _lib
from .csr import csr_of
from .neighbors import geometric_features_csr, GEOF_COLUMNS
from .ops import _workspace, segment_reduce

__all__ = ["sparse_sample", "scatter_std", "scatter_mean_orientation", "segment_features",
           "SEGMENT_BASE_FEATURES"]

# src/transforms/__init__ / src/utils/features: the keys SegmentFeatures computes
SEGMENT_BASE_FEATURES = ["linearity", "planarity", "scattering", "verticality", "curvature",
                         "log_length", "log_surface", "log_volume", "normal", "log_size"]


def _draw_seed():
    """64-bit seed from torch's CPU generator, so ``torch.manual_seed`` makes the
    sampling reproducible."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def sparse_sample(idx, n_max=32, n_min=1, mask=None, return_pointers=False, seed=None,
                  num_segments=None):
    """Indices of a random sample, without replacement, of at least ``n_min`` and at
    most ``n_max`` elements of every segment of ``idx`` (sparse.py:142-243).

    ``mask``: boolean [N] (or index tensor) of the elements that may be drawn.
    Returns ``idx_samples`` (grouped by segment), and the segment pointers when
    ``return_pointers``."""
    _lib.require_cuda(idx)
    if not 0 <= n_min <= n_max:
        raise ValueError("need 0 <= n_min <= n_max")
    dev = idx.device
    idx = idx.long().contiguous()
    n = idx.numel()
    if num_segments is None:
        num_segments = int(idx.max()) + 1 if n else 1
    m8 = None
    if mask is not None:
        mask = torch.as_tensor(mask, device=dev)
        if mask.dtype == torch.bool:
            m8 = mask.to(torch.uint8).contiguous()
        else:                                   # tensor_idx semantics: a list of positions
            m8 = torch.zeros(n, dtype=torch.uint8, device=dev)
            m8[mask.long()] = 1
    if seed is None:
        seed = _draw_seed()
    out_ptr = torch.empty(num_segments + 1, dtype=torch.int64, device=dev)
    out_idx = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    nbytes = _lib.lib.spt_sparse_sample_workspace_bytes(n, num_segments)
    ws = _workspace(nbytes, dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_sparse_sample(
            _lib.ptr(idx), n, num_segments, _lib.ptr(m8), int(n_max), int(n_min),
            int(seed) & (2 ** 64 - 1), _lib.ptr(out_ptr), _lib.ptr(out_idx), _lib.ptr(ws), nbytes,
            _lib.stream_ptr(dev))
    _lib.check(st, "spt_sparse_sample")
    total = int(out_ptr[-1])                    # the one host sync of the sampler
    idx_samples = out_idx[:total]
    if not return_pointers:
        return idx_samples
    return idx_samples, out_ptr


def scatter_std(x, idx, num_segments=None):
    """``torch_scatter.scatter_std(x, idx, dim=0)`` (unbiased; graph.py:285)."""
    _lib.require_cuda(x, idx)
    x2 = x.detach().float().contiguous()
    squeeze = x2.dim() == 1
    x2 = x2.view(x2.shape[0], -1)
    csr = csr_of(idx, num_segments)
    out = torch.empty((csr.num_seg, x2.shape[1]), dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device):
        st = _lib.lib.spt_segment_std_f32(
            _lib.ptr(x2), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), csr.num_seg, x2.shape[1],
            _lib.ptr(out), _lib.stream_ptr(x2.device))
    _lib.check(st, "spt_segment_std_f32")
    return out.view(-1) if squeeze else out


def scatter_pca(x, idx, num_segments=None):
    """Eigen-decomposition of every group's population covariance
    (src/utils/scatter.py:41-125): returns ``(eigenval [S,3] ascending, clamped at 0,
    eigenvec [S,3,3] with eigenvectors in columns)``; a group without rows gets (1,1,1) and
    the identity like the reference's NaN rule.  ``x`` [N,3]; ``idx`` [N] int64 (unsorted)."""
    _lib.require_cuda(x, idx)
    if x.dim() != 2 or x.shape[1] != 3:
        raise NotImplementedError("scatter_pca is built for 3-D points (the only use in the reference)")
    csr = csr_of(idx, num_segments)
    p = x.detach().float().contiguous()
    val = torch.empty((csr.num_seg, 3), dtype=torch.float32, device=p.device)
    vec = torch.empty((csr.num_seg, 3, 3), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        st = _lib.lib.spt_scatter_pca_f32(_lib.ptr(p), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr),
                                          csr.num_seg, _lib.ptr(val), _lib.ptr(vec),
                                          _lib.stream_ptr(p.device))
    _lib.check(st, "spt_scatter_pca_f32")
    return val, vec


def scatter_mean_orientation(orientation, idx, num_segments=None):
    """Mean orientation of the vectors of each segment, up to sign, expressed in the
    z+ half-space (scatter.py:249-300)."""
    _lib.require_cuda(orientation, idx)
    o = orientation.detach().float().contiguous()
    if o.dim() != 2 or o.shape[1] != 3:
        raise ValueError("orientation must be [N, 3]")
    csr = csr_of(idx, num_segments)
    out = torch.empty((csr.num_seg, 3), dtype=torch.float32, device=o.device)
    with torch.cuda.device(o.device):
        st = _lib.lib.spt_segment_mean_orientation_f32(
            _lib.ptr(o), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), csr.num_seg, _lib.ptr(out),
            _lib.stream_ptr(o.device))
    _lib.check(st, "spt_segment_mean_orientation_f32")
    return out


def segment_features(pos, super_index, num_segments, sub_size=None, n_max=32, n_min=5,
                     keys=None, mean_keys=None, std_keys=None, point_attrs=None, seed=None,
                     samples=None):
    """Handcrafted features of the segments of one level from the level-0 points
    (``_compute_cluster_features``, graph.py:193-321).

    ``pos`` [N0,3]; ``super_index`` [N0] = segment of each level-0 point
    (``nag.get_super_index(i_level)``); ``sub_size`` [S] number of level-0 points per
    segment (``nag.get_sub_size``; counted here when omitted); ``point_attrs``: dict of
    level-0 attributes for the ``mean_<key>`` / ``std_<key>`` outputs.  ``samples`` =
    ``(idx_samples, ptr)`` overrides the random draw (deterministic option the
    reference asks for at graph.py:212-214).  Returns a dict of [S, ·] tensors."""
    _lib.require_cuda(pos, super_index)
    keys = SEGMENT_BASE_FEATURES if keys is None else list(keys)
    point_attrs = point_attrs or {}
    mean_keys = list(point_attrs) if mean_keys is None else list(mean_keys)
    std_keys = list(point_attrs) if std_keys is None else list(std_keys)
    out = {}
    csr = csr_of(super_index, num_segments)
    if sub_size is None:
        sub_size = csr.counts().long()
    geof_needed = (set(keys) & set(SEGMENT_BASE_FEATURES)) - {"log_size"}
    if geof_needed:
        if samples is None:
            samples = sparse_sample(super_index, n_max=n_max, n_min=n_min, return_pointers=True,
                                    seed=seed, num_segments=num_segments)
        idx_samples, ptr = samples
        # k_min = 5: the default of geometric_features, which graph.py:244-247 does not override
        f = geometric_features_csr(pos, idx_samples, ptr, k_min=5, add_self=False, raw=False)
        col = {k: i for i, k in enumerate(GEOF_COLUMNS)}
        for key in geof_needed:
            if key == "normal":
                out[key] = f[:, 4:7]
            elif key.startswith("log_"):
                out[key] = torch.log(f[:, col[key[4:]]:col[key[4:]] + 1] + 1)
            else:
                out[key] = f[:, col[key]:col[key] + 1]
    if "log_size" in keys:
        out["log_size"] = (torch.log(sub_size.float() + 1).view(-1, 1) - math.log(2)) / 10
    for key in mean_keys:
        a = point_attrs[key]
        if key == "normal":
            out[f"mean_{key}"] = scatter_mean_orientation(a, super_index, num_segments)
        else:
            out[f"mean_{key}"] = segment_reduce(a.float(), super_index, num_segments, "mean")
    for key in std_keys:
        out[f"std_{key}"] = scatter_std(point_attrs[key], super_index, num_segments)
    return out
