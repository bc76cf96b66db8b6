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


def _float_src(src, what):
    """The segment kernels compute in f32: f16 / bf16 sources widen exactly, f64 would
    silently lose precision -> refused (no call site of the hot path passes f64)."""
    if src.dtype == torch.float64:
        raise NotImplementedError(f"HIP shim: {what} of a float64 source (the kernels are f32)")
    return src


def _int_minmax(src, index, dim_size, op):
    """Integer min / max on the f32 segment kernels, exact for |v| < 2^47 and without a host sync:
    v = hi * 2^24 + lo (hi = v >> 24 floors, 0 <= lo < 2^24), the order of v is the lexicographic
    order of (hi, lo) and both halves are exact in f32 - first the winning hi per segment, then
    the winning lo among the rows that carry it (the others see a sentinel)."""
    from ..csr import csr_of
    csr = csr_of(index, dim_size)
    v = src.long()
    hi, lo = (v >> 24).float(), (v & 0xFFFFFF).float()
    ohi = ops.segment_reduce(hi, csr, None, op)
    sentinel = float(1 << 24) if op == "min" else -1.0
    lo = torch.where(hi == ops.gather_rows(ohi, csr.idx), lo, torch.full_like(lo, sentinel))
    olo, arg = ops.segment_reduce(lo, csr, None, op, return_arg=True)
    empty = (csr.counts() == 0).view((-1,) + (1,) * (src.dim() - 1))
    out = ohi.long() * (1 << 24) + olo.long()
    return torch.where(empty, torch.zeros_like(out), out), arg


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    if not src.is_floating_point():
        o = ops.segment_sum_i64(src, index, dim_size)            # bit-exact (nag.py:97,108)
        return o if src.dtype == torch.bool else o.to(src.dtype)
    return ops.segment_reduce(_float_src(src, "scatter_sum"), index, dim_size, "sum").to(src.dtype)


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    if not src.is_floating_point():
        # torch_scatter: integer sum, then floor division by the clamped count
        from ..csr import csr_of
        csr = csr_of(index, dim_size)
        tot = ops.segment_sum_i64(src, csr, None)
        cnt = csr.counts().long().clamp(min=1).view((-1,) + (1,) * (src.dim() - 1))
        return torch.div(tot, cnt, rounding_mode="floor").to(src.dtype)
    return ops.segment_reduce(_float_src(src, "scatter_mean"), index, dim_size, "mean").to(src.dtype)


def _minmax(src, index, dim_size, op):
    if not src.is_floating_point():
        o, a = _int_minmax(src, index, dim_size, op)
        return o.to(src.dtype), a.long()
    o, a = ops.segment_reduce(_float_src(src, f"scatter_{op}"), index, dim_size, op, return_arg=True)
    return o.to(src.dtype), a.long()


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    return _minmax(src, index, dim_size, "min")


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    _check(src, index, dim, out)
    return _minmax(src, index, dim_size, "max")


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
