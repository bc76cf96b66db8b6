"""``torch_scatter`` API (the subset the reference calls) on the
segment-CSR HIP kernels.  ``index`` is a 1-D int64 tensor reducing along dim 0;
``dim_size=None`` costs the same host sync as upstream (``index.max()+1``)."""
import torch

from .. import ops

__all__ = ["scatter", "scatter_sum", "scatter_add", "scatter_mean", "scatter_min",
           "scatter_max", "scatter_std"]


def _check(src, index, dim, out):
    if out is not None:
        raise NotImplementedError("out= is not supported by the HIP shim")
    if index.dim() != 1 or dim not in (0, -src.dim()):
        raise NotImplementedError(
            "the HIP shim reduces along dim 0 with a 1-D index (every call site of the "
            "reference's hot path does)")


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    if not src.is_floating_point():
        return ops.segment_sum_i64(src, index, dim_size)        # bit-exact (nag.py:97,108)
    return ops.segment_reduce(src, index, dim_size, "sum")


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    return ops.segment_reduce(src, index, dim_size, "mean")


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    o, a = ops.segment_reduce(src, index, dim_size, "min", return_arg=True)
    return o, a.long()


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    o, a = ops.segment_reduce(src, index, dim_size, "max", return_arg=True)
    return o, a.long()


def scatter_std(src, index, dim=-1, out=None, dim_size=None, unbiased=True):
    _check(src, index, dim, out)
    from ..csr import csr_of
    if unbiased and not src.requires_grad and src.dim() <= 2:
        from ..segment import scatter_std as segment_std      # one kernel, two passes in-cache
        return segment_std(src, index, dim_size).to(src.dtype)
    csr = csr_of(index, dim_size)
    cnt = csr.counts().to(src.dtype).view((-1,) + (1,) * (src.dim() - 1))
    mean = ops.segment_reduce(src, csr, None, "sum") / cnt.clamp(min=1)
    var = ops.segment_reduce((src - ops.gather_rows(mean, csr.idx)) ** 2, csr, None, "sum")
    denom = (cnt - 1 if unbiased else cnt).clamp(min=1)
    return (var / (denom + 1e-6)).sqrt()


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == "min":
        return scatter_min(src, index, dim, out, dim_size)[0]
    if reduce == "max":
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)
