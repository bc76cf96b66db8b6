"""Functional, differentiable front-ends of the HIP segment kernels.

Every function here launches a kernel of libspt_hip.so on torch's current
stream; there is no eager-PyTorch path.
"""
import ctypes
import os

import torch

from . import _lib
from . import precision as _precision
from .csr import EdgeCSR, SegmentCSR, csr_of, edge_csr_of

_OPS = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3}

# Optional per-kernel HIP-event timers (bench.py's roofline leg): name -> list
# of (start, end) events recorded on the stream the kernel is launched on.
_TIMERS = {}
_TIMER_PREFIXES = []
_TIMER_CAP = 8192            # event pairs kept per timer (timers are opt-in measurement aids)


def enable_timer(name, prefix=False):
    """Record a HIP event pair (torch's current stream = the launch stream) around every call site
    named ``name`` (``prefix=True``: every site whose name starts with it, pooled under it)."""
    _TIMERS[name] = []
    if prefix and name not in _TIMER_PREFIXES:
        _TIMER_PREFIXES.append(name)


def timer_count(name):
    return len(_TIMERS.get(name) or [])


def timer_mean_ms(name):
    pairs = _TIMERS.get(name) or []
    if not pairs:
        return None
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)


def reset_timers():
    for k in _TIMERS:
        _TIMERS[k] = []


_TIMERS_PAUSED = False


def pause_timers(on=True):
    """Stop / resume recording event pairs (a stream capture cannot hold timing events; a second
    workload timed next to the first must not pool into its timers).  Returns the old setting."""
    global _TIMERS_PAUSED
    old, _TIMERS_PAUSED = _TIMERS_PAUSED, bool(on)
    return old


_CONST = {}


def _const_tensor(vals, dtype, dev):
    """Small host-known table as a device tensor, uploaded ONCE per (values, dtype, device): a
    fresh ``torch.tensor(list, device=...)`` is a pageable host-to-device copy - a stall in an
    eager step and not capturable into a graph."""
    key = (tuple(vals), dtype, dev.index)
    t = _CONST.get(key)
    if t is None:
        if len(_CONST) > 512:
            _CONST.clear()
        t = _CONST[key] = torch.tensor(list(vals), dtype=dtype, device=dev)
    return t


class _timed:
    def __init__(self, name):
        self.rec = None if _TIMERS_PAUSED else _TIMERS.get(name)
        if _TIMERS_PAUSED:
            return
        if self.rec is None and _TIMER_PREFIXES:
            for p in _TIMER_PREFIXES:
                if name.startswith(p):
                    self.rec = _TIMERS[p]
                    break

    def __enter__(self):
        if self.rec is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            if len(self.rec) >= _TIMER_CAP:      # a loop longer than a benchmark: keep the newest
                del self.rec[:_TIMER_CAP // 2]
            self.rec.append((self.a, b))


def _as_rows(x):
    """[n, ...] -> contiguous, 16B-aligned f32 [n, c] plus the trailing shape."""
    tail = x.shape[1:]
    x2 = x.reshape(x.shape[0], -1)
    if x2.dtype != torch.float32:
        x2 = x2.float()
    x2 = x2.contiguous()
    if x2.data_ptr() % 16:
        x2 = x2.clone()
    return x2, tail


def _seg_reduce_fwd(x2, csr, op, want_arg):
    n, c = x2.shape
    out = torch.empty((csr.num_seg, c), dtype=torch.float32, device=x2.device)
    arg = None
    if want_arg and op in (2, 3):
        arg = torch.empty((csr.num_seg, c), dtype=torch.int32, device=x2.device)
    with torch.cuda.device(x2.device), _timed(f"segcsr_reduce_fwd:{op}:{n}x{c}"):
        st = _lib.lib.spt_segcsr_reduce_f32(
            op, _lib.ptr(x2), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), n,
            csr.num_seg, c, _lib.ptr(out), _lib.ptr(arg), _lib.stream_ptr(x2.device))
    _lib.check(st, "spt_segcsr_reduce_f32")
    return out, arg


def _seg_reduce_bwd(gout, arg, csr, op, n):
    c = gout.shape[1]
    gx = torch.empty((n, c), dtype=torch.float32, device=gout.device)
    with torch.cuda.device(gout.device), _timed(f"segcsr_reduce_bwd:{op}:{n}x{c}"):
        st = _lib.lib.spt_segcsr_reduce_bwd_f32(
            op, _lib.ptr(gout), _lib.ptr(arg), _lib.ptr(csr.idx),
            _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), n, csr.num_seg, c, _lib.ptr(gx),
            _lib.stream_ptr(gout.device))
    _lib.check(st, "spt_segcsr_reduce_bwd_f32")
    return gx


def _gather_fwd(x2, idx):
    n = idx.numel()
    c = x2.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device):
        st = _lib.lib.spt_gather_rows_f32(
            _lib.ptr(x2), _lib.ptr(idx), n, x2.shape[0], c, _lib.ptr(out),
            _lib.stream_ptr(x2.device))
    _lib.check(st, "spt_gather_rows_f32")
    return out


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, csr, op, want_arg):
        _lib.require_cuda(x)
        x2, tail = _as_rows(x)
        if x2.shape[0] != csr.n:
            raise ValueError(
                f"src has {x2.shape[0]} rows but the index has {csr.n}")
        need_arg = op in (2, 3) and (want_arg or x.requires_grad)
        out, arg = _seg_reduce_fwd(x2, csr, op, need_arg)
        ctx.csr, ctx.op, ctx.n, ctx.tail, ctx.in_dtype = csr, op, x2.shape[0], tail, x.dtype
        ctx.save_for_backward(arg)
        out = out.reshape((csr.num_seg,) + tuple(tail)).to(x.dtype)
        if arg is not None:
            arg_out = arg.reshape((csr.num_seg,) + tuple(tail))
            ctx.mark_non_differentiable(arg_out)
            return out, arg_out
        return out, None

    @staticmethod
    def backward(ctx, gout, _garg):
        (arg,) = ctx.saved_tensors
        g2, _ = _as_rows(gout)
        gx = _seg_reduce_bwd(g2, arg, ctx.csr, ctx.op, ctx.n)
        return gx.reshape((ctx.n,) + tuple(ctx.tail)).to(ctx.in_dtype), None, None, None


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, csr_holder):
        _lib.require_cuda(x, idx)
        x2, tail = _as_rows(x)
        if idx.dtype != torch.int64:
            idx = idx.long()
        idx = idx.contiguous()
        ctx.idx, ctx.holder = idx, csr_holder
        ctx.num_src, ctx.tail, ctx.in_dtype = x2.shape[0], tail, x.dtype
        out = _gather_fwd(x2, idx)
        return out.reshape((idx.numel(),) + tuple(tail)).to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        g2, _ = _as_rows(gout)
        # backward of a gather is a segment sum over the CSR view of idx
        csr = csr_of(ctx.holder if ctx.holder is not None else ctx.idx, ctx.num_src)
        gx, _ = _seg_reduce_fwd(g2, csr, 0, False)
        return gx.reshape((ctx.num_src,) + tuple(ctx.tail)).to(ctx.in_dtype), None, None


def segment_reduce(x, index, num_seg=None, reduce="sum", return_arg=False):
    """out[s] = reduce_{i: index[i]==s} x[i] along dim 0 (empty -> 0).

    ``index`` is an int64 [n] tensor or a prebuilt :class:`SegmentCSR`.
    For min/max and ``return_arg=True`` also returns the int32 arg rows
    (``n`` for empty segments).
    """
    op = _OPS[reduce]
    csr = csr_of(index, num_seg)
    out, arg = _SegmentReduce.apply(x, csr, op, return_arg)
    if return_arg:
        return out, arg
    return out


def gather_rows(x, idx):
    """out[i] = x[idx[i]]  (IndexUnpool forward; backward = segment sum)."""
    return _GatherRows.apply(x, idx, idx)


class _EdgeAffinityFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, edge_index):
        _lib.require_cuda(x, edge_index)
        xc = x.detach().float().contiguous()
        ei = edge_index.long().contiguous()
        n, c = xc.shape
        e = ei.shape[1]
        out = torch.empty((e, 2 * c), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(xc.device):
            st = _lib.lib.spt_edge_affinity_features_f32(
                _lib.ptr(xc), n, c, _lib.ptr(ei[0]), _lib.ptr(ei[1]), e, _lib.ptr(out),
                _lib.stream_ptr(xc.device))
        _lib.check(st, "spt_edge_affinity_features_f32")
        ctx.save_for_backward(xc, ei)
        ctx.in_dtype = x.dtype
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, gout):
        xc, ei = ctx.saved_tensors
        n, c = xc.shape
        e = ei.shape[1]
        g = gout.float().contiguous()
        gend = torch.empty((2 * e, c), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(xc.device):
            st = _lib.lib.spt_edge_affinity_features_bwd_f32(
                _lib.ptr(xc), _lib.ptr(g), n, c, _lib.ptr(ei[0]), _lib.ptr(ei[1]), e,
                _lib.ptr(gend), _lib.stream_ptr(xc.device))
        _lib.check(st, "spt_edge_affinity_features_bwd_f32")
        # per-node sums over both endpoint lists: one CSR segment reduce (memoised on the
        # edge_index tensor's flattened view for the lifetime of the batch)
        gx, _ = _seg_reduce_fwd(gend, csr_of(ei.reshape(-1), n), 0, False)
        return gx.to(ctx.in_dtype), None


def edge_affinity_features(x, edge_index):
    """``cat(|x[a] - x[b]|, (x[a] + x[b]) / 2)`` for the edges ``(a, b) = edge_index``
    (src/models/panoptic.py:477-480), one fused pass each way."""
    return _EdgeAffinityFeatures.apply(x, edge_index)


def segment_sum_i64(x, index, num_seg=None):
    """Bit-exact int64 segment sum (NAG.get_sub_size chain)."""
    _lib.require_cuda(x)
    csr = csr_of(index, num_seg)
    tail = x.shape[1:]
    x2 = x.reshape(x.shape[0], -1).long().contiguous()
    c = x2.shape[1]
    out = torch.empty((csr.num_seg, c), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        st = _lib.lib.spt_segcsr_sum_i64(
            _lib.ptr(x2), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), x2.shape[0],
            csr.num_seg, c, _lib.ptr(out), _lib.stream_ptr(x.device))
    _lib.check(st, "spt_segcsr_sum_i64")
    return out.reshape((csr.num_seg,) + tuple(tail))


# ---------------------------------------------------------------------------
# UnitSphereNorm (src/nn/norm.py:67-138)
# ---------------------------------------------------------------------------
def _usn_args(pos, idx, w, num_super):
    _lib.require_cuda(pos)
    pos = pos.detach()
    if pos.dtype != torch.float32:
        pos = pos.float()
    pos = pos.contiguous()
    n = pos.shape[0]
    dev = pos.device
    if idx is None:
        num_seg = 1
        perm = None
        rowptr = _const_tensor([0, n], torch.int32, dev)
        idx_t = None
    else:
        csr = csr_of(idx, num_super)
        num_seg, perm, rowptr, idx_t = csr.num_seg, csr.perm, csr.rowptr, csr.idx
    wf = wi = None
    if w is not None:
        w = w.detach().contiguous()
        if w.dtype == torch.int64:
            wi = w
        elif w.is_floating_point():
            wf = w.float()
        else:
            wi = w.long()
    return pos, n, dev, num_seg, perm, rowptr, idx_t, wf, wi


def unit_sphere_norm(pos, idx, w=None, num_super=None):
    """Returns (pos_normalised [n,3], diameter [num_super,1]) like
    UnitSphereNorm.forward with log_diameter=False.  ``idx=None`` normalises
    all rows together (norm.py:86-110).  No gradient: positions are data."""
    pos, n, dev, num_seg, perm, rowptr, idx_t, wf, wi = _usn_args(pos, idx, w, num_super)
    out = torch.empty_like(pos)
    diam = torch.empty(num_seg, dtype=torch.float32, device=dev)
    center = torch.empty((num_seg, 3), dtype=torch.float32, device=dev)
    nb = _lib.lib.spt_unit_sphere_workspace_bytes(n, num_seg)
    ws = _workspace(nb, dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_unit_sphere_norm_ws_f32(
            _lib.ptr(pos), _lib.ptr(idx_t), _lib.ptr(perm), _lib.ptr(rowptr),
            _lib.ptr(wf), _lib.ptr(wi), n, num_seg, _lib.ptr(out), _lib.ptr(diam),
            _lib.ptr(center), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(st, "spt_unit_sphere_norm_ws_f32")
    return out, diam.view(-1, 1)


class _UnitSphereAssemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, idx, w, num_super):
        pos, n, dev, num_seg, perm, rowptr, idx_t, wf, wi = _usn_args(pos, idx, w, num_super)
        x2 = _f32c(x.detach())
        cx = x2.shape[1]
        out = torch.empty((n, 4 + cx), dtype=torch.float32, device=dev)
        diam = torch.empty(num_seg, dtype=torch.float32, device=dev)
        center = torch.empty((num_seg, 3), dtype=torch.float32, device=dev)
        nb = _lib.lib.spt_unit_sphere_workspace_bytes(n, num_seg)
        ws = _workspace(nb, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_unit_sphere_assemble_ws_f32(
                _lib.ptr(pos), _lib.ptr(idx_t), _lib.ptr(perm), _lib.ptr(rowptr), _lib.ptr(wf),
                _lib.ptr(wi), n, num_seg, _lib.ptr(x2), cx, _lib.ptr(out), _lib.ptr(diam),
                _lib.ptr(center), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "spt_unit_sphere_assemble_ws_f32")
        diam = diam.view(-1, 1)
        ctx.mark_non_differentiable(diam)
        ctx.x_dtype = x.dtype
        return out, diam

    @staticmethod
    def backward(ctx, gout, _gdiam):
        return gout[:, 4:].to(ctx.x_dtype), None, None, None, None


def unit_sphere_assemble_ok(x, pos):
    """Whether ``unit_sphere_assemble`` applies: f32 CUDA features of a multiple of 4 columns."""
    return (torch.is_tensor(x) and torch.is_tensor(pos) and x.is_cuda and x.dim() == 2
            and x.dtype == torch.float32 and x.shape[1] >= 4 and x.shape[1] % 4 == 0
            and x.shape[0] == pos.shape[0])


def unit_sphere_assemble(x, pos, idx, w=None, num_super=None):
    """``(cat([diameter[idx], pos_normalised, x], 1), diameter)``: UnitSphereNorm and the three
    fusion concatenations of a stage's input (src/nn/stage.py:249-271 with ``use_pos`` and
    ``use_diameter_parent``) in one writing pass.  The gradient reaches ``x`` (a column slice)."""
    return _UnitSphereAssemble.apply(x, pos, idx, w, num_super)


# ---------------------------------------------------------------------------
# GraphNorm (+ fused LeakyReLU)
# ---------------------------------------------------------------------------
_WS = {}


def _workspace(nbytes, dev):
    """Grow-only scratch buffer per (device, stream).  Kernels that share it
    are ordered on the stream, so reuse across calls is safe."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


class _GraphNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch, num_graphs, weight, bias, mean_scale, eps, slope):
        _lib.require_cuda(x)
        x2 = x.contiguous()
        if x2.dtype != torch.float32:
            x2 = x2.float()
        r, d = x2.shape
        dev = x2.device
        if batch is not None:
            batch = batch.contiguous()
            if batch.dtype != torch.int64:
                batch = batch.long()
        w, b, a = (t.detach().float().contiguous() for t in (weight, bias, mean_scale))
        y = torch.empty_like(x2)
        mean = torch.empty((num_graphs, d), dtype=torch.float32, device=dev)
        rstd = torch.empty((num_graphs, d), dtype=torch.float32, device=dev)
        nb = _lib.lib.spt_graphnorm_workspace_bytes(r, d, num_graphs)
        ws = _workspace(nb, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_graphnorm_fwd_f32(
                _lib.ptr(x2), _lib.ptr(batch), r, d, num_graphs, _lib.ptr(w),
                _lib.ptr(b), _lib.ptr(a), eps, slope, _lib.ptr(y), _lib.ptr(mean),
                _lib.ptr(rstd), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "spt_graphnorm_fwd_f32")
        ctx.save_for_backward(x2, batch, w, b, a, mean, rstd)
        ctx.meta = (num_graphs, slope, x.dtype)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        x2, batch, w, b, a, mean, rstd = ctx.saved_tensors
        num_graphs, slope, in_dtype = ctx.meta
        r, d = x2.shape
        dev = x2.device
        gy = gy.contiguous().float()
        gx = torch.empty_like(x2)
        gw = torch.empty(d, dtype=torch.float32, device=dev)
        gb = torch.empty(d, dtype=torch.float32, device=dev)
        ga = torch.empty(d, dtype=torch.float32, device=dev)
        nb = _lib.lib.spt_graphnorm_workspace_bytes(r, d, num_graphs)
        ws = _workspace(nb, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_graphnorm_bwd_f32(
                _lib.ptr(x2), _lib.ptr(gy), _lib.ptr(batch), r, d, num_graphs,
                _lib.ptr(w), _lib.ptr(b), _lib.ptr(a), _lib.ptr(mean), _lib.ptr(rstd),
                slope, _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ga),
                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "spt_graphnorm_bwd_f32")
        return gx.to(in_dtype), None, None, gw, gb, ga, None, None


def graph_norm(x, batch, weight, bias, mean_scale, eps=1e-5, num_graphs=None,
               act_slope=1.0):
    """GraphNorm(x, batch) [+ LeakyReLU(act_slope) when act_slope != 1].
    ``num_graphs=None`` costs a host sync (``batch.max()+1``) exactly like
    PyG's GraphNorm; pass it to stay asynchronous."""
    if batch is None:
        num_graphs = 1
    elif num_graphs is None:
        num_graphs = int(batch.max().item()) + 1 if batch.numel() else 1
    return _GraphNorm.apply(x, batch, int(num_graphs), weight, bias, mean_scale,
                            float(eps), float(act_slope))


# ---------------------------------------------------------------------------
# Fused edge attention (src/nn/attention.py:202-315)
# ---------------------------------------------------------------------------
def _f32c(t):
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_EA_SHARE = os.environ.get("SPT_EA_GRAD_SHARE", "1") != "0"


def share_edge_attr_grad(on=None):
    """Whether ``Stage.forward`` lets its blocks accumulate d edge_attr in one buffer (default on;
    ``SPT_EA_GRAD_SHARE=0`` in the environment turns it off).  Returns the previous setting."""
    global _EA_SHARE
    prev = _EA_SHARE
    if on is not None:
        _EA_SHARE = bool(on)
    return prev


class EdgeAttrGradShare:
    """One gradient buffer for the ``edge_attr`` of a stage: every transformer block of the
    stage reads the same ``edge_attr`` (src/nn/stage.py:137-141), so instead of each attention
    backward returning its own [E, F] tensor for autograd to sum, the blocks accumulate into one
    buffer (``spt_edge_attn_bwd_acc_f32``) and only the block that ran FIRST in the forward - the
    last one autograd reaches, every later block depends on its output - hands it to autograd.

    Valid for one forward pass over sequentially dependent blocks, which is what ``Stage.forward``
    creates it for; a backward that stops short of the first block (``autograd.grad`` on an
    intermediate block's inputs) would not see the edge_attr gradient - do not pass a share in
    that case.

    A backward that fails part-way, or one that re-enters a block before the first block has
    handed the buffer over (a retained graph walked again after a partial ``autograd.grad``),
    must not add onto what an earlier walk left behind: ``acquire`` starts a fresh buffer
    whenever the entering block has already been seen in the current walk, ``abort`` drops it."""

    __slots__ = ("tensor", "count", "buf", "seen")

    def __init__(self):
        self.tensor, self.count, self.buf, self.seen = None, 0, None, set()

    def acquire(self, rank, like):
        """(buffer, acc) for the block of forward order ``rank``: acc = 1 when a later block of
        THIS backward walk already started the sum."""
        if rank in self.seen:                # a second walk without a hand-over: stale state
            self.buf, self.seen = None, set()
        self.seen.add(rank)
        if self.buf is not None:
            return self.buf, 1
        self.buf = torch.empty_like(like)
        return self.buf, 0

    def release(self, rank):
        """The buffer if ``rank`` is the block that hands it to autograd (the first one of the
        forward), else None."""
        if rank != 0:
            return None
        buf, self.buf, self.seen = self.buf, None, set()
        return buf

    def abort(self):
        self.buf, self.seen = None, set()

    def enter(self, edge_attr):
        if self.tensor is None:
            self.tensor = edge_attr
        elif self.tensor is not edge_attr:
            raise ValueError("EdgeAttrGradShare: the blocks were given different edge_attr tensors")
        self.count += 1
        return self.count - 1


class _EdgeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, ecsr, edge_attr, Wk, bk, Wq, bq, Wv, bv, H, D, scale_mode, scale_a,
                share=None):
        _lib.require_cuda(qkv)
        q2 = _f32c(qkv)
        n, ld = q2.shape
        Dv = (ld - 2 * H * D) // H
        if 2 * H * D + H * Dv != ld or ecsr.n != n:
            raise ValueError(f"qkv has shape {tuple(q2.shape)}: not [n={ecsr.n}, 2*H*D + H*Dv]")
        dev = q2.device
        ea = _f32c(edge_attr)
        F = ea.shape[1] if ea is not None else 0
        if ea is not None and ea.shape[0] != ecsr.e:
            raise ValueError("edge_attr rows != number of edges")
        ps = [_f32c(t) for t in (Wk, bk, Wq, bq, Wv, bv)]
        out = torch.empty((n, H * Dv), dtype=torch.float32, device=dev)
        m = torch.empty((n, H), dtype=torch.float32, device=dev)
        z = torch.empty((n, H), dtype=torch.float32, device=dev)
        mode = _precision.attention_mode()       # per call; the backward runs in the same mode
        with torch.cuda.device(dev), _timed(f"edge_attn_fwd:{n}:{ecsr.e}"):
            st = _lib.lib.spt_edge_attn_fwd_ex_f32(
                _lib.ptr(q2), n, H, D, Dv, _lib.ptr(ecsr.erowptr), _lib.ptr(ecsr.eperm),
                _lib.ptr(ecsr.tgt_sorted), ecsr.e, _lib.ptr(ea), F,
                *[_lib.ptr(t) for t in ps], scale_mode, scale_a, _lib.ptr(out),
                _lib.ptr(m), _lib.ptr(z), mode, _lib.stream_ptr(dev))
        _lib.check(st, "spt_edge_attn_fwd_ex_f32")
        ctx.mode = mode
        ctx.save_for_backward(q2, ea, *[t for t in ps if t is not None], out, m, z)
        ctx.present = [t is not None for t in ps]
        ctx.has_ea = ea is not None
        ctx.meta = (ecsr, H, D, Dv, F, scale_mode, scale_a, qkv.dtype,
                    None if edge_attr is None else edge_attr.dtype)
        ctx.share = ctx.rank = None
        if (share is not None and edge_attr is not None and edge_attr.dtype == torch.float32
                and edge_attr.requires_grad and (Wk is not None or Wq is not None or Wv is not None)):
            ctx.share, ctx.rank = share, share.enter(edge_attr)
        return out.to(qkv.dtype)

    @staticmethod
    def backward(ctx, gout):
        ecsr, H, D, Dv, F, scale_mode, scale_a, q_dtype, ea_dtype = ctx.meta
        saved = list(ctx.saved_tensors)
        q2 = saved.pop(0)
        ea = saved.pop(0) if ctx.has_ea else None
        if not ctx.has_ea:
            saved.pop(0)
        ps = [saved.pop(0) if pres else None for pres in ctx.present]
        out, m, z = saved
        n, ld = q2.shape
        dev = q2.device
        g = _f32c(gout)
        gqkv = torch.empty_like(q2)
        share, acc = ctx.share, 0
        if share is not None:
            gea, acc = share.acquire(ctx.rank, ea)       # acc: a later block already started the sum
        else:
            gea = torch.empty_like(ea) if ea is not None else None
        gps = [torch.empty_like(t) if t is not None else None for t in ps]
        # the edge-lane formulation needs 512 B per edge + ~390 B per node of scratch on top of the
        # weight-gradient tables: asked for only when that formulation will run (the scratch
        # buffer is grow-only per device and stream)
        el = bool(ea is not None and ecsr.e > 0
                  and _lib.lib.spt_edge_attn_bwd_el_supported(H, D, Dv, F, ctx.mode))
        nb = (_lib.lib.spt_edge_attn_bwd_ex_workspace_bytes(n, ecsr.e, H, D, Dv, max(F, 1)) if el
              else _lib.lib.spt_edge_attn_bwd_workspace_bytes(H, D, Dv, max(F, 1)))
        ws = _workspace(nb, dev)
        src = tids = tperm = trowptr = None
        if el:
            src = ecsr.src_sorted()
            tids = ecsr.tile_ids(ctx.mode)
            if not ecsr.mirrored(ctx.mode):
                # (a mirrored edge list's target-order records need no sorted target view)
                tv = ecsr.target_view()
                tperm, trowptr = tv.perm, tv.rowptr
        with torch.cuda.device(dev), _timed(f"edge_attn_bwd:{n}:{ecsr.e}"):
            st = _lib.lib.spt_edge_attn_bwd_ex_f32(
                _lib.ptr(q2), n, H, D, Dv, _lib.ptr(ecsr.erowptr), _lib.ptr(ecsr.eperm),
                _lib.ptr(ecsr.tgt_sorted), _lib.ptr(src), _lib.ptr(tids), _lib.ptr(tperm),
                _lib.ptr(trowptr),
                ecsr.e, _lib.ptr(ea), F,
                *[_lib.ptr(t) for t in ps], scale_mode, scale_a, _lib.ptr(out),
                _lib.ptr(m), _lib.ptr(z), _lib.ptr(g), _lib.ptr(gqkv), _lib.ptr(gea), acc,
                *[_lib.ptr(t) for t in gps], ctx.mode, _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(dev))
        if st != 0 and share is not None:
            share.abort()                                # never leave a half-built sum behind
        _lib.check(st, "spt_edge_attn_bwd_ex_f32")
        if gea is not None and not (any(ctx.present[0::2])):
            gea = None
        if share is not None:
            gea = share.release(ctx.rank)                # the first block returns the sum
        return (gqkv.to(q_dtype), None, None if gea is None else gea.to(ea_dtype),
                *gps, None, None, None, None, None)


def edge_attention(qkv, edge_index, edge_attr=None, k_rpe=None, q_rpe=None, v_rpe=None,
                   num_heads=1, qk_dim=8, scale_mode=0, scale_a=1.0, ea_grad=None):
    """out[s] = sum_e softmax_e(<q_e, k_e>) v_e over the edges leaving s.

    ``qkv`` [N, 2*H*D + C] is the output of the block's qkv Linear;
    ``edge_index`` a [2,E] tensor or an :class:`EdgeCSR`; ``k_rpe`` etc. are
    (weight, bias) pairs of the RPE Linears or None.  Returns [N, C]
    (before out_proj).  ``ea_grad``: an :class:`EdgeAttrGradShare` common to the blocks of a
    stage (their d edge_attr accumulate into one buffer)."""
    ecsr = edge_csr_of(edge_index, qkv.shape[0])

    def wb(p):
        return (None, None) if p is None else (p[0], p[1])

    Wk, bk = wb(k_rpe)
    Wq, bq = wb(q_rpe)
    Wv, bv = wb(v_rpe)
    H, D = int(num_heads), int(qk_dim)
    share = ea_grad if torch.is_grad_enabled() else None
    split = _matrix_pipe_split(qkv, edge_attr, Wk, Wq, Wv, H, D)
    if split is None:
        padded = _matrix_pipe_pad(qkv, ecsr, edge_attr, Wk, bk, Wq, bq, Wv, bv, H, D)
        if padded is not None:
            # narrower head layouts (nano: 16 heads of qk_dim 2, value dim 1, 16-D edge encodings)
            # zero-padded to the built (16, 4, 4, 32) shape: the matrix-pipe kernels instead of
            # the generic VALU ones; plain torch ops do the padding, autograd their backward
            qp, eap, wk, bkp, wq, bqp, wv, bvp, dv = padded
            out = _EdgeAttention.apply(qp, ecsr, eap, wk, bkp, wq, bqp, wv, bvp, 16, 4,
                                       int(scale_mode), float(scale_a), None)
            return out.view(out.shape[0], 16, 4)[:, :, :dv].reshape(out.shape[0], 16 * dv)
        return _EdgeAttention.apply(qkv, ecsr, edge_attr, Wk, bk, Wq, bq, Wv, bv, H, D,
                                    int(scale_mode), float(scale_a), share)
    # wider head layouts (SPT-128: 16 heads of value dim 8; 32 heads) on the matrix-pipe kernels
    G, J, _ = split
    return _EdgeAttentionSplit.apply(qkv, ecsr, edge_attr, Wk, bk, Wq, bq, Wv, bv, H, G, J,
                                     int(scale_mode), float(scale_a), share)


class _EdgeAttentionSplit(torch.autograd.Function):
    """Wider head layouts on the matrix-pipe kernels, by decomposition.

    The MFMA / edge-lane kernels are built for 16 heads of qk_dim 4 and value dim 4.  Heads are
    independent attention problems and, within a head, the value dims only enter through
    out = sum_e a_e v_e (linear in v; its backward dc = a (<g, v> - <g, out>) is a SUM over the
    value dims), so H = 16 G heads of value dim 4 J decompose EXACTLY into G x J problems of the
    built shape: head group g, value slice j -> columns [q_g | k_g | v_g[:, :, 4j:4j+4]] with the
    matching rows of the RPE weights.  Every pass recomputes the softmax of its head group (the
    price: J x the q / k work).  SPT-128 (KITTI-360: H 16, value dim 8) = 2 passes, 32 heads x 4
    (ScanNet) = 2 passes.

    One autograd node for all passes: the operands of every pass are laid out by ONE gather
    ([G J, N, 192]), the passes write into slabs of one buffer, d edge_attr accumulates in place
    (``acc``) and the q / k / weight gradients of a head group's J passes are summed by one
    reduction each - a per-pass autograd graph (slices, cat, CopySlices, AccumulateGrad) cost
    ~40 five-microsecond launches per block, which is what a train batch's step time is made of."""

    @staticmethod
    def _cols(H, G, J, dev):
        key = (H, G, J, str(dev))
        c = _SPLIT_COLS.get(key)
        if c is None:
            QK, Dv = H * 4, 4 * J
            cols = []
            for gi in range(G):
                for j in range(J):
                    qc = torch.arange(64 * gi, 64 * gi + 64)
                    vc = (2 * QK + (16 * gi + torch.arange(16)).view(16, 1) * Dv + 4 * j
                          + torch.arange(4).view(1, 4)).reshape(64)
                    cols.append(torch.cat([qc, QK + qc, vc]))
            c = _SPLIT_COLS[key] = torch.stack(cols).to(dev)            # [G J, 192]
        return c

    @staticmethod
    def forward(ctx, qkv, ecsr, edge_attr, Wk, bk, Wq, bq, Wv, bv, H, G, J, scale_mode, scale_a,
                share=None):
        _lib.require_cuda(qkv)
        q2 = _f32c(qkv)
        n = q2.shape[0]
        dev = q2.device
        ea = _f32c(edge_attr)
        if ecsr.n != n or ea.shape[0] != ecsr.e:
            raise ValueError("qkv / edge_attr rows do not match the graph")
        P, Dv, F = G * J, 4 * J, 32
        qa = torch.empty((P, n, 192), dtype=torch.float32, device=dev)               # [P, n, 192]
        with torch.cuda.device(dev):
            st = _lib.lib.spt_attn_split_pack_f32(_lib.ptr(q2), n, G, J, _lib.ptr(qa), _lib.stream_ptr(dev))
        _lib.check(st, "spt_attn_split_pack_f32")
        Wk2, Wq2 = _f32c(Wk), _f32c(Wq)
        bk2, bq2 = _f32c(bk), _f32c(bq)
        Wva = _f32c(Wv).view(G, 16, J, 4, F).permute(0, 2, 1, 3, 4).contiguous()      # [G, J, 64, F]
        bva = None if bv is None else _f32c(bv).view(G, 16, J, 4).permute(0, 2, 1, 3).contiguous()
        oa = torch.empty((P, n, 64), dtype=torch.float32, device=dev)
        m = torch.empty((P, n, 16), dtype=torch.float32, device=dev)
        z = torch.empty((P, n, 16), dtype=torch.float32, device=dev)
        mode = _precision.attention_mode()
        with torch.cuda.device(dev):
            for gi in range(G):
                cs = slice(64 * gi, 64 * gi + 64)
                for j in range(J):
                    p = gi * J + j
                    with _timed(f"edge_attn_fwd:{n}:{ecsr.e}"):
                        st = _lib.lib.spt_edge_attn_fwd_ex_f32(
                            _lib.ptr(qa[p]), n, 16, 4, 4, _lib.ptr(ecsr.erowptr), _lib.ptr(ecsr.eperm),
                            _lib.ptr(ecsr.tgt_sorted), ecsr.e, _lib.ptr(ea), F,
                            _lib.ptr(Wk2[cs]), _lib.ptr(None if bk2 is None else bk2[cs]),
                            _lib.ptr(Wq2[cs]), _lib.ptr(None if bq2 is None else bq2[cs]),
                            _lib.ptr(Wva[gi, j]), _lib.ptr(None if bva is None else bva[gi, j]),
                            scale_mode, scale_a, _lib.ptr(oa[p]), _lib.ptr(m[p]), _lib.ptr(z[p]),
                            mode, _lib.stream_ptr(dev))
                    _lib.check(st, "spt_edge_attn_fwd_ex_f32")
        ctx.mode = mode
        ctx.save_for_backward(qa, ea, Wk2, Wq2, Wva, oa, m, z,
                              *[t for t in (bk2, bq2, bva) if t is not None])
        ctx.has_b = [t is not None for t in (bk2, bq2, bva)]
        ctx.meta = (ecsr, H, G, J, scale_mode, scale_a, qkv.dtype, edge_attr.dtype)
        ctx.share = ctx.rank = None
        if share is not None and edge_attr.dtype == torch.float32 and edge_attr.requires_grad:
            ctx.share, ctx.rank = share, share.enter(edge_attr)
        out = oa.view(G, J, n, 16, 4).permute(2, 0, 3, 1, 4).reshape(n, H * Dv)
        return out.to(qkv.dtype)

    @staticmethod
    def backward(ctx, gout):
        ecsr, H, G, J, scale_mode, scale_a, q_dtype, ea_dtype = ctx.meta
        saved = list(ctx.saved_tensors)
        qa, ea, Wk2, Wq2, Wva, oa, m, z = saved[:8]
        rest = saved[8:]
        bk2, bq2, bva = [rest.pop(0) if h else None for h in ctx.has_b]
        P, n = qa.shape[0], qa.shape[1]
        dev, F, Dv, QK = qa.device, 32, 4 * J, 4 * H
        ga = _f32c(gout).view(n, G, 16, J, 4).permute(1, 3, 0, 2, 4).contiguous().view(P, n, 64)
        gqa = torch.empty_like(qa)
        share, acc = ctx.share, 0
        if share is not None:
            gea, acc = share.acquire(ctx.rank, ea)
        else:
            gea = torch.empty_like(ea)
        gWk = torch.empty((P, 64, F), dtype=torch.float32, device=dev)
        gWq = torch.empty_like(gWk)
        gWv = torch.empty_like(gWk)
        gbk = torch.empty((P, 64), dtype=torch.float32, device=dev) if bk2 is not None else None
        gbq = torch.empty((P, 64), dtype=torch.float32, device=dev) if bq2 is not None else None
        gbv = torch.empty((P, 64), dtype=torch.float32, device=dev) if bva is not None else None
        el = bool(ecsr.e > 0 and _lib.lib.spt_edge_attn_bwd_el_supported(16, 4, 4, F, ctx.mode))
        nb = (_lib.lib.spt_edge_attn_bwd_ex_workspace_bytes(n, ecsr.e, 16, 4, 4, F) if el
              else _lib.lib.spt_edge_attn_bwd_workspace_bytes(16, 4, 4, F))
        ws = _workspace(nb, dev)
        src = tids = tperm = trowptr = None
        if el:
            src, tids = ecsr.src_sorted(), ecsr.tile_ids(ctx.mode)
            if not ecsr.mirrored(ctx.mode):
                tv = ecsr.target_view()
                tperm, trowptr = tv.perm, tv.rowptr
        with torch.cuda.device(dev):
            for gi in range(G):
                cs = slice(64 * gi, 64 * gi + 64)
                for j in range(J):
                    p = gi * J + j
                    with _timed(f"edge_attn_bwd:{n}:{ecsr.e}"):
                        st = _lib.lib.spt_edge_attn_bwd_ex_f32(
                            _lib.ptr(qa[p]), n, 16, 4, 4, _lib.ptr(ecsr.erowptr), _lib.ptr(ecsr.eperm),
                            _lib.ptr(ecsr.tgt_sorted), _lib.ptr(src), _lib.ptr(tids), _lib.ptr(tperm),
                            _lib.ptr(trowptr), ecsr.e, _lib.ptr(ea), F,
                            _lib.ptr(Wk2[cs]), _lib.ptr(None if bk2 is None else bk2[cs]),
                            _lib.ptr(Wq2[cs]), _lib.ptr(None if bq2 is None else bq2[cs]),
                            _lib.ptr(Wva[gi, j]), _lib.ptr(None if bva is None else bva[gi, j]),
                            scale_mode, scale_a, _lib.ptr(oa[p]), _lib.ptr(m[p]), _lib.ptr(z[p]),
                            _lib.ptr(ga[p]), _lib.ptr(gqa[p]), _lib.ptr(gea), acc,
                            _lib.ptr(gWk[p]), _lib.ptr(None if gbk is None else gbk[p]),
                            _lib.ptr(gWq[p]), _lib.ptr(None if gbq is None else gbq[p]),
                            _lib.ptr(gWv[p]), _lib.ptr(None if gbv is None else gbv[p]),
                            ctx.mode, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
                    if st != 0 and share is not None:
                        share.abort()
                    _lib.check(st, "spt_edge_attn_bwd_ex_f32")
                    acc = 1
        # q / k (and their encoders') gradients: the sum over a head group's J value slices
        gqkv = torch.empty((n, 2 * QK + H * Dv), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_attn_split_grad_f32(_lib.ptr(gqa), n, G, J, _lib.ptr(gqkv), _lib.stream_ptr(dev))
        _lib.check(st, "spt_attn_split_grad_f32")

        def over_j(t, tail):
            if t is None:
                return None
            t = t.view(G, J, *tail)
            return (t.sum(1) if J > 1 else t[:, 0]).reshape(G * tail[0], *tail[1:])

        def value_rows(t, tail):                       # [G, J, 16, 4, ...] -> rows h * Dv + 4 j + d
            if t is None:
                return None
            t = t.view(G, J, 16, 4, *tail)
            return t.permute(0, 2, 1, 3, *range(4, 4 + len(tail))).reshape(H * Dv, *tail)

        if share is not None:
            gea = share.release(ctx.rank)
        return (gqkv.to(q_dtype), None, None if gea is None else gea.to(ea_dtype),
                over_j(gWk, (64, F)), over_j(gbk, (64,)), over_j(gWq, (64, F)), over_j(gbq, (64,)),
                value_rows(gWv, (F,)), value_rows(gbv, ()), None, None, None, None, None, None)


_SPLIT_COLS = {}


PAD_MIN_EDGES = 1 << 16        # below that the generic kernels are a few microseconds anyway


def _matrix_pipe_pad(qkv, ecsr, edge_attr, Wk, bk, Wq, bq, Wv, bv, H, D):
    """Operands of a head layout NARROWER than the built (16, 4, 4, 32) one, zero-padded to it
    (H = 16, qk_dim <= 4, value dim <= 4, in_rpe_dim <= 32, all three RPE encoders, a
    matrix-pipe precision active, enough edges to matter); None otherwise.  Exact: padded q / k
    dims add zero to every dot product, padded value dims and edge-encoding columns meet zero
    weights, the padded output dims are dropped."""
    if H != 16 or D > 4 or edge_attr is None or Wk is None or Wq is None or Wv is None:
        return None
    F = edge_attr.shape[1]
    C = qkv.shape[1] - 2 * H * D
    if C <= 0 or C % H or F > 32 or ecsr.e < PAD_MIN_EDGES:
        return None
    Dv = C // H
    if Dv > 4 or (D == 4 and Dv == 4 and F == 32):
        return None
    if qkv.dtype != torch.float32 or edge_attr.dtype != torch.float32:
        return None
    mode = _precision.attention_mode()
    if mode < 0:
        mode = _lib.lib.spt_attn_use_mfma(-2)
    if (mode & 3) == 0:
        return None
    n, e = qkv.shape[0], edge_attr.shape[0]
    pad = torch.nn.functional.pad

    def heads(t, d):                                 # [rows, H * d] -> [rows, H * 4]
        return t if d == 4 else pad(t.reshape(t.shape[0], H, d), (0, 4 - d)).reshape(t.shape[0], H * 4)

    def enc(w, b, d):                                # ([H d, F], [H d]) -> ([H 4, 32], [H 4])
        w = pad(w.reshape(H, d, F), (0, 32 - F, 0, 4 - d)).reshape(H * 4, 32)
        b = None if b is None else pad(b.reshape(H, d), (0, 4 - d)).reshape(H * 4)
        return w.contiguous(), b
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    qp = torch.cat([heads(q, D), heads(k, D), heads(v, Dv)], dim=1)
    eap = edge_attr if F == 32 else pad(edge_attr, (0, 32 - F))
    wk, bkp = enc(Wk, bk, D)
    wq, bqp = enc(Wq, bq, D)
    wv, bvp = enc(Wv, bv, Dv)
    return qp, eap, wk, bkp, wq, bqp, wv, bvp, Dv


def _matrix_pipe_split(qkv, edge_attr, Wk, Wq, Wv, H, D):
    """(G, J, Dv) when a head layout other than the built (16, 4, 4) decomposes into G x J passes
    of it (H = 16 G, qk_dim 4, value dim 4 J, in_rpe_dim 32, all three RPE encoders, a matrix-pipe
    precision active); None for the built shape itself and for everything else (generic kernels)."""
    if D != 4 or H % 16 or edge_attr is None or Wk is None or Wq is None or Wv is None:
        return None
    if edge_attr.shape[1] != 32:
        return None
    C = qkv.shape[1] - 2 * H * D
    if C <= 0 or C % H:
        return None
    Dv = C // H
    if Dv % 4 or (H == 16 and Dv == 4):
        return None
    mode = _precision.attention_mode()
    if mode < 0:
        mode = _lib.lib.spt_attn_use_mfma(-2)          # query: any value < -1 leaves the default
    if (mode & 3) == 0:
        return None
    return H // 16, Dv // 4, Dv


# ---------------------------------------------------------------------------
# Dense Linear on tall-skinny operands (plumbing over rocBLAS / hipBLASLt)
# ---------------------------------------------------------------------------
_DW_CHUNK = 32768
_SKINNY_MIN_ROWS = 4096


def _skinny_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.dim() == 2 and x.shape[0] >= _SKINNY_MIN_ROWS
            and _lib.lib.spt_skinny_linear_supported(weight.shape[1], weight.shape[0]))


def _skinny_launch(x, weight, bias, mode=-1):
    """y = x W^T + b on the MFMA skinny-GEMM kernel (csrc/skinny_linear.hip); ``mode``: the matrix
    mode word of the call (``precision.skinny_mode()``; -1 = the library's default)."""
    x = x.contiguous()
    weight = weight.contiguous()
    rows, k = x.shape
    n = weight.shape[0]
    y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = _lib.lib.spt_skinny_linear_pre_m_f32(
            _lib.ptr(x), rows, k, _lib.ptr(weight), _lib.ptr(bias), n, _lib.ptr(y),
            None, None, None, None, 1, None, int(mode), _lib.stream_ptr(x.device))
    _lib.check(st, "spt_skinny_linear_pre_m_f32")
    return y


def _input_grad(g, weight, mode=-1):
    """dX = G W of a Linear (``weight`` [N_out, N_in] as the layer holds it): the skinny kernel reads
    the weight transposed while it stages the slab (``spt_skinny_linear_wt_f32``) - no transposed
    copy per backward call -, else the transposed copy on the same kernel, else the library."""
    n_out, n_in = weight.shape
    if (g.is_cuda and g.dtype == torch.float32 and weight.dtype == torch.float32 and g.dim() == 2
            and g.shape[0] >= _SKINNY_MIN_ROWS and n_in % 4 == 0 and n_in >= 64
            and _lib.lib.spt_skinny_linear_supported(n_out, n_in)):
        g = g.contiguous()
        w = weight.detach().contiguous()
        rows = g.shape[0]
        y = torch.empty((rows, n_in), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            st = _lib.lib.spt_skinny_linear_wt_m_f32(_lib.ptr(g), rows, n_out, _lib.ptr(w), n_in,
                                                     _lib.ptr(y), int(mode), _lib.stream_ptr(g.device))
        _lib.check(st, "spt_skinny_linear_wt_m_f32")
        return y
    wt = weight.detach().t().contiguous()           # [K, N]: dX = G (W^T)^T
    return _skinny_launch(g, wt, None, mode) if _skinny_ok(g, wt) else g @ weight


def _dw_batched(g, x):
    """dW = G^T X reduces over millions of rows into a <=192x192 output, a shape the
    library runs on a handful of workgroups - so it is issued as a BATCHED GEMM over row
    chunks (thousands of workgroups) followed by a small sum."""
    n = x.shape[0]
    nb = n // _DW_CHUNK
    main = nb * _DW_CHUNK
    gw = None
    if nb > 0:
        gm = g[:main].view(nb, _DW_CHUNK, g.shape[1])
        xm = x[:main].view(nb, _DW_CHUNK, x.shape[1])
        gw = torch.bmm(gm.transpose(1, 2), xm).sum(0)
    if main < n:
        tail = g[main:].t() @ x[main:]
        gw = tail if gw is None else gw + tail
    return gw


def _skinny_dw_ok(g, x):
    return (x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32
            and x.shape[0] >= _SKINNY_MIN_ROWS
            and _lib.lib.spt_skinny_dw_supported(x.shape[1], g.shape[1]))


def _skinny_dw(g, x, want_bias=False, mode=-1):
    """dW = G^T X (and, from the same pass, db = column sums of G) on the MFMA kernel of
    csrc/skinny_linear.hip (per-wave partials, fixed-order sum)."""
    g, x = g.contiguous(), x.detach().contiguous()
    rows, k = x.shape
    n = g.shape[1]
    gw = torch.empty((n, k), dtype=torch.float32, device=x.device)
    gb = torch.empty(n, dtype=torch.float32, device=x.device) if want_bias else None
    nb = _lib.lib.spt_skinny_dw_workspace_bytes(k, n)
    ws = _workspace(nb, x.device)
    with torch.cuda.device(x.device):
        st = _lib.lib.spt_skinny_dw_pre_m_f32(_lib.ptr(g), _lib.ptr(x), rows, n, k, _lib.ptr(gw),
                                              _lib.ptr(gb), None, None, None, None, 1, int(mode),
                                              _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
    _lib.check(st, "spt_skinny_dw_pre_m_f32")
    return (gw, gb) if want_bias else gw


class _TallLinear(torch.autograd.Function):
    """y = x W^T (+ b) for [rows >> features] operands.  Forward and dX run on the
    hand-written skinny-GEMM kernel when the shape is built (K in {32,64,128,192},
    N % 64 == 0 - the attention block's qkv / out_proj), else on the library; dW on the
    skinny dW kernel (K in {32, 64}) or a batched library GEMM over row chunks."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.smode = _precision.skinny_mode()      # the backward runs in the mode of the forward
        if _skinny_ok(x, weight):
            return _skinny_launch(x.detach(), weight.detach(), None if bias is None else bias.detach(),
                                  ctx.smode)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        if (x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32
                and weight.dtype == torch.float32 and x.shape[0] >= _SKINNY_MIN_ROWS
                and _lib.lib.spt_narrow_linear_bwd_supported(x.shape[1], g.shape[1])):
            # narrow head: dX, dW and db from one pass over (x, g)
            xc, wc = x.detach().contiguous(), weight.detach().contiguous()
            rows, k = xc.shape
            n = g.shape[1]
            dev = xc.device
            gx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
            gw = torch.empty((n, k), dtype=torch.float32, device=dev)
            gb = torch.empty(n, dtype=torch.float32, device=dev) if ctx.has_bias else None
            nb = _lib.lib.spt_narrow_linear_bwd_workspace_bytes(k, n)
            ws = _workspace(nb, dev)
            with torch.cuda.device(dev):
                st = _lib.lib.spt_narrow_linear_bwd_f32(
                    _lib.ptr(g), _lib.ptr(xc), _lib.ptr(wc), rows, n, k, _lib.ptr(gx),
                    _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            _lib.check(st, "spt_narrow_linear_bwd_f32")
            return gx, gw, gb
        gx = None
        smode = getattr(ctx, "smode", -1)
        if ctx.needs_input_grad[0]:
            gx = _input_grad(g, weight, smode)
        gw = gb = None
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if _skinny_dw_ok(g, x):
                gw = _skinny_dw(g, x, want_gb, smode)
                if want_gb:
                    gw, gb = gw
            else:
                gw = _dw_batched(g, x) if x.shape[0] >= 4 * _DW_CHUNK else g.t() @ x
        if want_gb and gb is None:
            gb = g.sum(0)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """``nn.Linear`` forward for tall-skinny operands (see _TallLinear); small or
    unsupported inputs go straight to the library."""
    if x.dim() != 2 or not x.is_cuda or x.shape[0] < _SKINNY_MIN_ROWS:
        return torch.nn.functional.linear(x, weight, bias)
    return _TallLinear.apply(x, weight, bias)


class _ResidualLinear(torch.autograd.Function):
    """y = residual + (x W^T + b): a transformer block's `shortcut + out_proj(.)`
    (src/nn/transformer.py:231-234, src/nn/attention.py:318-319) with the add in the skinny
    kernel's epilogue (bitwise `residual + linear(x)`).  The residual's gradient is the incoming
    gradient itself."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.smode = _precision.skinny_mode()
        xd, wd = x.detach().contiguous(), weight.detach().contiguous()
        rows, k = xd.shape
        n = wd.shape[0]
        y = torch.empty((rows, n), dtype=torch.float32, device=xd.device)
        with torch.cuda.device(xd.device):
            st = _lib.lib.spt_skinny_linear_pre_m_f32(
                _lib.ptr(xd), rows, k, _lib.ptr(wd),
                _lib.ptr(None if bias is None else bias.detach().contiguous()), n, _lib.ptr(y),
                None, None, None, None, 1, _lib.ptr(residual.detach().contiguous()),
                ctx.smode, _lib.stream_ptr(xd.device))
        _lib.check(st, "spt_skinny_linear_pre_m_f32")
        return y

    @staticmethod
    def backward(ctx, g):
        gx, gw, gb = _TallLinear.backward(ctx, g)
        return gx, gw, gb, g


_FUSE_PRENORM = os.environ.get("SPT_FUSE_PRENORM", "1") != "0"


def fuse_prenorm(on=None):
    """Switch of the folded pre-norm / residual route of the transformer blocks (default on;
    ``SPT_FUSE_PRENORM=0`` in the environment turns it off).  Returns the previous setting."""
    global _FUSE_PRENORM
    prev = _FUSE_PRENORM
    if on is not None:
        _FUSE_PRENORM = bool(on)
    return prev


def linear_residual_ok(residual, weight):
    """Whether ``linear_residual`` runs fused for a Linear with this weight whose output is
    added to ``residual`` [rows, N]: tall f32 rows, (K, N) built."""
    return bool(_FUSE_PRENORM and residual.dim() == 2 and residual.is_cuda
                and residual.dtype == torch.float32 and weight.dtype == torch.float32
                and residual.shape[0] >= _SKINNY_MIN_ROWS and residual.shape[1] == weight.shape[0]
                and _lib.lib.spt_skinny_pre_supported(weight.shape[1], weight.shape[0], 1))


def linear_residual(x, weight, bias, residual):
    """``residual + F.linear(x, weight, bias)`` with the add folded into the Linear's kernel where
    the shape is built (tall f32 operands, K in {32, 64, 128}, N a multiple of 64)."""
    if (x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] == residual.shape[0]
            and linear_residual_ok(residual, weight)):
        return _ResidualLinear.apply(x, weight, bias, residual)
    return residual + linear(x, weight, bias)


class _NormLinear(torch.autograd.Function):
    """qkv = Linear(GraphNorm(x)) with the norm applied INSIDE the Linear's read of x
    (`spt_graphnorm_stats_f32` + `spt_skinny_linear_pre_f32`): the pre-norm of a transformer block
    (src/nn/transformer.py:231-234) without the normalised [rows, C] tensor ever being written.
    Returns ``(y, x_res)``: ``x_res`` is x itself, handed back so that the block's residual
    branch hangs on THIS node - the backward then receives both gradients of x and the norm's
    backward kernel adds them in its own pass (`spt_graphnorm_bwd_acc_f32`), instead of autograd
    summing two [rows, C] tensors with one more launch.  Values are bitwise those of
    ``linear(graph_norm(x))``."""

    @staticmethod
    def forward(ctx, x, batch, num_graphs, gn_w, gn_b, gn_a, eps, weight, bias):
        _lib.require_cuda(x)
        xd = x.detach().contiguous()
        rows, d = xd.shape
        dev = xd.device
        B = int(num_graphs)
        if batch is not None:
            batch = batch.contiguous()
            if batch.dtype != torch.int64:
                batch = batch.long()
        w, b, a = (t.detach().float().contiguous() for t in (gn_w, gn_b, gn_a))
        wd = weight.detach().contiguous()
        bd = None if bias is None else bias.detach().contiguous()
        n = wd.shape[0]
        mean, rstd, am, sc = (torch.empty((B, d), dtype=torch.float32, device=dev) for _ in range(4))
        y = torch.empty((rows, n), dtype=torch.float32, device=dev)
        nb = _lib.lib.spt_graphnorm_workspace_bytes(rows, d, B)
        ws = _workspace(nb, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_graphnorm_stats_f32(
                _lib.ptr(xd), _lib.ptr(batch), rows, d, B, _lib.ptr(w), _lib.ptr(a), float(eps),
                _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(am), _lib.ptr(sc), _lib.ptr(ws),
                ws.numel(), _lib.stream_ptr(dev))
            _lib.check(st, "spt_graphnorm_stats_f32")
            smode = _precision.skinny_mode()
            st = _lib.lib.spt_skinny_linear_pre_m_f32(
                _lib.ptr(xd), rows, d, _lib.ptr(wd), _lib.ptr(bd), n, _lib.ptr(y), _lib.ptr(am),
                _lib.ptr(sc), _lib.ptr(b), _lib.ptr(batch), B, None, smode, _lib.stream_ptr(dev))
        _lib.check(st, "spt_skinny_linear_pre_m_f32")
        ctx.save_for_backward(xd, batch, w, b, a, mean, rstd, am, sc, wd)
        ctx.meta = (B, bias is not None)
        ctx.smode = smode                          # the backward runs in the mode of the forward
        # an unused output arrives as None in the backward instead of a dense zero [rows, C]
        # tensor that the norm's backward kernel would read for nothing
        ctx.set_materialize_grads(False)
        return y, x

    @staticmethod
    def backward(ctx, gy, gres):
        xd, batch, w, b, a, mean, rstd, am, sc, wd = ctx.saved_tensors
        B, has_bias = ctx.meta
        if gy is None:                       # only the residual branch was used
            return gres, None, None, None, None, None, None, None, None
        rows, d = xd.shape
        n = wd.shape[0]
        dev = xd.device
        gy = gy.contiguous()
        # gradient wrt the normalised rows: dX of the Linear
        gxn = _input_grad(gy, wd, ctx.smode)
        # weight / bias gradient against the rows normalised on the fly
        gw = torch.empty((n, d), dtype=torch.float32, device=dev)
        gb = torch.empty(n, dtype=torch.float32, device=dev) if has_bias else None
        ws = _workspace(_lib.lib.spt_skinny_dw_workspace_bytes(d, n), dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_skinny_dw_pre_m_f32(
                _lib.ptr(gy), _lib.ptr(xd), rows, n, d, _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(am),
                _lib.ptr(sc), _lib.ptr(b), _lib.ptr(batch), B, ctx.smode, _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(dev))
        _lib.check(st, "spt_skinny_dw_pre_m_f32")
        # the norm's backward, with the residual branch's gradient added in its apply pass
        gx = torch.empty_like(xd)
        g_w, g_b, g_a = (torch.empty(d, dtype=torch.float32, device=dev) for _ in range(3))
        if gres is not None:
            gres = gres.contiguous()
        ws = _workspace(_lib.lib.spt_graphnorm_workspace_bytes(rows, d, B), dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_graphnorm_bwd_acc_f32(
                _lib.ptr(xd), _lib.ptr(gxn), _lib.ptr(batch), rows, d, B, _lib.ptr(w), _lib.ptr(b),
                _lib.ptr(a), _lib.ptr(mean), _lib.ptr(rstd), 1.0, _lib.ptr(gres), _lib.ptr(gx),
                _lib.ptr(g_w), _lib.ptr(g_b), _lib.ptr(g_a), _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(dev))
        _lib.check(st, "spt_graphnorm_bwd_acc_f32")
        return gx, None, None, g_w, g_b, g_a, None, gw, gb


def norm_linear_ok(x, batch, num_graphs, weight):
    """Whether ``norm_linear`` runs fused: tall f32 rows, a built (K, N), few graphs."""
    if not _FUSE_PRENORM or (batch is not None and num_graphs is None):
        return False
    B = 1 if batch is None else int(num_graphs)
    return bool(x.dim() == 2 and x.is_cuda and x.dtype == torch.float32
                and weight.dtype == torch.float32 and x.shape[0] >= _SKINNY_MIN_ROWS
                and _lib.lib.spt_skinny_pre_supported(x.shape[1], weight.shape[0], B)
                and _lib.lib.spt_skinny_dw_supported(x.shape[1], weight.shape[0]))


_BCHK_ATTR = "_spt_batch_checked"


def _check_batch_ids(batch, B):
    """The folded pre-norm kernels index LDS tables by ``batch[row]``: ids outside [0, B) must fail
    here, as they do on the unfused ``graph_norm`` route, not read tables out of range.  One
    memoised device reduction per batch vector (none when its maker left the host ranges)."""
    if batch is None or _host_ptr(batch, B) is not None:
        return
    key = (batch._version, int(B), batch.data_ptr(), batch.numel())
    if getattr(batch, _BCHK_ATTR, None) == key:
        return
    if batch.numel():
        lo, hi = torch.aminmax(batch)
        if int(lo) < 0 or int(hi) >= int(B):
            raise ValueError(f"graph ids of `batch` must lie in [0, {int(B)}): found [{int(lo)}, {int(hi)}]")
    try:
        setattr(batch, _BCHK_ATTR, key)
    except Exception:
        pass


def norm_linear(x, batch, num_graphs, gn_weight, gn_bias, gn_mean_scale, eps, weight, bias):
    """``(linear(graph_norm(x, batch), weight, bias), x_res)`` - see ``_NormLinear``.  Callers
    check ``norm_linear_ok`` first."""
    B = 1 if batch is None else int(num_graphs)
    _check_batch_ids(batch, B)
    return _NormLinear.apply(x, batch, B, gn_weight, gn_bias, gn_mean_scale, float(eps), weight, bias)


# ---------------------------------------------------------------------------
# Cross-entropy of the heads' logits (csrc/loss.hip)
# ---------------------------------------------------------------------------
class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        _lib.require_cuda(logits, target)
        lg = _f32c(logits)
        tg = target.long().contiguous()
        rows, c = lg.shape
        dev = lg.device
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)          # loss | count
        nb = _lib.lib.spt_cross_entropy_workspace_bytes(rows)
        ws = _workspace(nb, dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_cross_entropy_fwd_f32(
                _lib.ptr(lg), _lib.ptr(tg), rows, c, int(ignore_index), _lib.ptr(lse),
                _lib.ptr(out), _lib.ptr(out[1:]), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "spt_cross_entropy_fwd_f32")
        ctx.save_for_backward(lg, tg, lse, out)
        ctx.meta = (int(ignore_index), logits.dtype)
        return out[0].to(logits.dtype)

    @staticmethod
    def backward(ctx, gout):
        lg, tg, lse, out = ctx.saved_tensors
        ignore_index, dtype = ctx.meta
        rows, c = lg.shape
        dev = lg.device
        g = gout.detach().float().reshape(1).contiguous()
        glog = torch.empty_like(lg)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_cross_entropy_bwd_f32(
                _lib.ptr(lg), _lib.ptr(tg), _lib.ptr(lse), rows, c, ignore_index, _lib.ptr(g),
                _lib.ptr(out[1:]), _lib.ptr(glog), _lib.stream_ptr(dev))
        _lib.check(st, "spt_cross_entropy_bwd_f32")
        return glog.to(dtype), None, None


def cross_entropy(logits, target, ignore_index=-100):
    """``torch.nn.functional.cross_entropy(logits, target, ignore_index=...)`` (mean reduction, no
    class weights) for [rows, C <= 32] logits on the HIP kernels; anything else goes to torch."""
    if (logits.is_cuda and logits.dim() == 2 and 1 <= logits.shape[1] <= 32
            and logits.shape[0] >= 1 and target.dim() == 1):
        return _CrossEntropy.apply(logits, target, ignore_index)
    return torch.nn.functional.cross_entropy(logits, target, ignore_index=ignore_index)


# ---------------------------------------------------------------------------
# Fused MLP: [bias-free Linear -> GraphNorm -> LeakyReLU] x L  (src/nn/mlp.py:8-94)
# ---------------------------------------------------------------------------
_GPTR_ATTR = "_spt_graph_ranges"


def _host_ptr(batch, num_graphs):
    """Row ranges [B+1] of the clouds of a batch vector when its maker left them on the tensor
    (``NAG.from_nag_list`` / the synthetic batches: ``Batch.ptr`` of the reference) - host
    knowledge, no device read-back - else None."""
    hp = getattr(batch, "_spt_host_ptr", None)
    if (hp is None or len(hp) != int(num_graphs) + 1 or hp[0] != 0 or hp[-1] != batch.numel()
            or batch._version != 0):
        return None
    return list(hp)


def graph_ranges(batch, num_graphs, rows):
    """Host row ranges [B+1] of the graphs when ``batch`` is sorted (clouds of a
    NAGBatch are contiguous), else None.  One host sync per batch tensor (memoised
    on the tensor, like the CSR views)."""
    if batch is None or (num_graphs is not None and int(num_graphs) == 1):
        return [0, rows]                                 # one graph: nothing to look at
    memo = getattr(batch, _GPTR_ATTR, None)
    key = (batch._version, int(num_graphs), batch.data_ptr(), batch.numel())
    if memo is not None and memo[0] == key:
        return memo[1]
    hp = _host_ptr(batch, num_graphs)
    if hp is not None:
        return hp
    if batch.numel() and not bool((batch[1:] >= batch[:-1]).all()):
        ranges = None
    else:
        counts = torch.bincount(batch, minlength=int(num_graphs)).tolist()
        ranges = [0]
        for c_ in counts:
            ranges.append(ranges[-1] + c_)
    try:
        setattr(batch, _GPTR_ATTR, (key, ranges))
    except Exception:
        pass
    return ranges


MAX_FUSED_RUNS = 16          # FMLP_MAX_RUNS of csrc/common.hpp


class GraphRuns:
    """Row ranges of constant graph id, sorted by graph: what one launch of the fused layer
    kernels covers (``spt_fused_linear_*_runs_f32``).  ``sorted_batch``: one run per graph in
    row order (a batch whose clouds are contiguous)."""

    __slots__ = ("r0", "r1", "g", "B", "rows", "sorted_batch", "_c")

    def __init__(self, runs, B, rows):
        runs = sorted(runs, key=lambda t: (t[2], t[0]))
        self.r0 = [int(t[0]) for t in runs]
        self.r1 = [int(t[1]) for t in runs]
        self.g = [int(t[2]) for t in runs]
        self.B, self.rows = int(B), int(rows)
        self.sorted_batch = (len(runs) <= self.B and
                             all(self.r1[i] <= self.r0[i + 1] for i in range(len(runs) - 1)))
        self._c = None

    @property
    def n(self):
        return len(self.r0)

    def c_arrays(self):
        """(nruns, int64[] r0, int64[] r1, int32[] graph) for the C ABI (host arrays)."""
        if self._c is None:
            import ctypes
            n = self.n
            self._c = (n, (ctypes.c_int64 * n)(*self.r0), (ctypes.c_int64 * n)(*self.r1),
                       (ctypes.c_int32 * n)(*self.g))
        return self._c

    def rows_per_graph(self):
        out = [0] * self.B
        for a, b, g in zip(self.r0, self.r1, self.g):
            out[g] += b - a
        return out


def graph_runs(batch, num_graphs, rows):
    """``GraphRuns`` of a per-row graph index, or None when it has more than 16 runs / graphs
    (the caller then takes the layer-by-layer route).  ``batch`` may be sorted (one run per
    cloud) or piecewise sorted (the edge MLP's norm index).  One host sync per batch tensor
    (memoised on the tensor, like the CSR views)."""
    if batch is None or (num_graphs is not None and int(num_graphs) == 1):
        return GraphRuns([(0, rows, 0)], 1, rows)
    B = int(num_graphs)
    if B > MAX_FUSED_RUNS:
        return None
    memo = getattr(batch, _GRUN_ATTR, None)
    key = (batch._version, B, batch.data_ptr(), batch.numel())
    if memo is not None and memo[0] == key:
        return memo[1]
    runs = None
    hp = _host_ptr(batch, B)
    if batch.numel() == 0:
        runs = GraphRuns([], B, 0)
    elif hp is not None:
        runs = GraphRuns([(hp[g], hp[g + 1], g) for g in range(B) if hp[g + 1] > hp[g]], B,
                         batch.numel())
    else:
        # run starts = positions where the id changes; read back only when there are few
        change = (batch[1:] != batch[:-1])
        n_change = int(change.sum())
        if n_change + 1 <= MAX_FUSED_RUNS:
            starts = [0] + (torch.nonzero(change).flatten() + 1).tolist()
            ends = starts[1:] + [int(batch.numel())]
            ids = batch[torch.tensor(starts, device=batch.device)].tolist()
            if all(0 <= g < B for g in ids):
                runs = GraphRuns(list(zip(starts, ends, ids)), B, batch.numel())
    try:
        setattr(batch, _GRUN_ATTR, (key, runs))
    except Exception:
        pass
    return runs


_GRUN_ATTR = "_spt_graph_runs"


def graph_runs_via(batch, num_graphs, holder, *deps):
    """``graph_runs(batch, ...)`` for a ``batch`` that is DERIVED from longer-lived tensors (the
    edge MLP's ``norm_index[edge_index[0]]`` is rebuilt every forward): the run table - host
    knowledge of the batch layout, like the row ranges of ``batch`` itself - is memoised on
    ``holder`` (the edge_index) keyed by the versions of ``deps``, so only the first forward on a
    batch pays the read-back."""
    key = tuple((d._version, d.data_ptr(), d.numel()) for d in deps) + (int(num_graphs),)
    memo = getattr(holder, _GRUN_ATTR + "_via", None)
    host = getattr(holder, "_spt_host_runs", None)
    if memo is not None and memo[0] == key:
        runs = memo[1]
    elif (host is not None and holder._version == 0 and host[0] == (int(num_graphs), batch.numel())
          and len(host[1]) <= MAX_FUSED_RUNS):
        # the batch's maker left the runs on the edge index (host knowledge of the batch layout,
        # like Batch.ptr: per-cloud edge counts are known where the clouds are concatenated)
        runs = GraphRuns(list(host[1]), int(num_graphs), batch.numel())
    else:
        runs = graph_runs(batch, num_graphs, batch.numel())
        try:
            setattr(holder, _GRUN_ATTR + "_via", (key, runs))
        except Exception:
            pass
    try:
        setattr(batch, _GRUN_ATTR, ((batch._version, int(num_graphs), batch.data_ptr(), batch.numel()), runs))
    except Exception:
        pass
    return runs


def fused_mlp_supported(dims):
    return all(_lib.lib.spt_fused_linear_supported(int(k), int(n))
               for k, n in zip(dims[:-1], dims[1:]))


_ST_H, _ST_X = 8, 16          # SPT_FMLP_H_BF16 / SPT_FMLP_X_BF16 of include/spt_hip.h


def _fmlp_forward(x, batch, runs, eps_list, slope_list, params, apply_last=True, fmode=-1,
                  store16=False):
    """Forward of the fused layer chain (``runs``: a ``GraphRuns``; ONE launch per layer whatever
    the number of graphs).  Returns (y or None, saved tensors, hs[-1], tables of
    the last GraphNorm): with ``apply_last=False`` the last norm + activation are left to the
    consumer (the fused max-pool)."""
    L = len(eps_list)
    Ws = [params[4 * i].detach().float().contiguous() for i in range(L)]
    gnw = [params[4 * i + 1].detach().float().contiguous() for i in range(L)]
    gnb = [params[4 * i + 2].detach().float().contiguous() for i in range(L)]
    gms = [params[4 * i + 3].detach().float().contiguous() for i in range(L)]
    x2 = x.detach().contiguous()
    if x2.dtype != torch.float32:
        x2 = x2.float()
    R = x2.shape[0]
    dev = x2.device
    B = runs.B
    nr, c_r0, c_r1, c_g = runs.c_arrays()
    sp = _lib.stream_ptr(dev)
    hs, tabs = [], []
    cur, pre = x2, None
    y = None
    with torch.cuda.device(dev):
        for l in range(L):
            N, K = Ws[l].shape
            # store16: raw layer outputs kept as bf16 (bf16 matrix mode: 3), read as bf16 by the
            # next layer (its x), the pool and the backward
            h = torch.empty((R, N), dtype=torch.bfloat16 if store16 else torch.float32, device=dev)
            lmode = (3 | _ST_H | (_ST_X if l else 0)) if store16 else fmode
            total = torch.empty((B, 2 * N + 1), dtype=torch.float64, device=dev)
            nb = _lib.lib.spt_fused_linear_workspace_bytes(K, N)
            ws = _workspace(nb, dev)
            pa = ps = pb = None
            if pre is not None:
                pa, ps, pb = pre
            mean = torch.empty((B, N), dtype=torch.float32, device=dev)
            rstd, am, sc = (torch.empty_like(mean) for _ in range(3))
            if FUSE_POST:
                # the layer's totals AND its norm's tables out of one post launch
                norm = _lib.GnFwdTables(_lib.ptr(gnw[l]), _lib.ptr(gms[l]), float(eps_list[l]),
                                        _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(am), _lib.ptr(sc))
                with _timed(f"fused_linear_fwd:{K}x{N}:{R}"):
                    st = _lib.lib.spt_fused_linear_fwd_runs_gn_f32(
                        _lib.ptr(cur), nr, c_r0, c_r1, c_g, B, K, _lib.ptr(Ws[l]), N,
                        _lib.ptr(pa), _lib.ptr(ps), _lib.ptr(pb),
                        float(slope_list[l - 1]) if l else 1.0, _lib.ptr(h),
                        _lib.ptr(total), lmode, _lib.ptr(ws), ws.numel(), ctypes.addressof(norm), sp)
                _lib.check(st, "spt_fused_linear_fwd_runs_gn_f32")
            else:
                with _timed(f"fused_linear_fwd:{K}x{N}:{R}"):
                    st = _lib.lib.spt_fused_linear_fwd_runs_f32(
                        _lib.ptr(cur), nr, c_r0, c_r1, c_g, B, K, _lib.ptr(Ws[l]), N,
                        _lib.ptr(pa), _lib.ptr(ps), _lib.ptr(pb),
                        float(slope_list[l - 1]) if l else 1.0, _lib.ptr(h),
                        _lib.ptr(total), lmode, _lib.ptr(ws), ws.numel(), sp)
                _lib.check(st, "spt_fused_linear_fwd_runs_f32")
                st = _lib.lib.spt_graphnorm_tables_f32(
                    _lib.ptr(total), B, N, _lib.ptr(gnw[l]), _lib.ptr(gms[l]), float(eps_list[l]),
                    _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(am), _lib.ptr(sc), sp)
                _lib.check(st, "spt_graphnorm_tables_f32")
            hs.append(h)
            tabs.append((mean, rstd, am, sc))
            cur, pre = h, (am, sc, gnb[l])
        if apply_last:
            assert not store16, "bf16 storage: the last norm is applied by the consumer (the pool)"
            N = Ws[-1].shape[0]
            y = torch.empty((R, N), dtype=torch.float32, device=dev)
            st = _lib.lib.spt_graphnorm_apply_f32(
                _lib.ptr(hs[-1]), _lib.ptr(batch) if B > 1 else None, R, N, B,
                _lib.ptr(tabs[-1][2]), _lib.ptr(tabs[-1][3]), _lib.ptr(gnb[-1]),
                float(slope_list[-1]), _lib.ptr(y), sp)
            _lib.check(st, "spt_graphnorm_apply_f32")
    saved = (x2, batch, *hs, *[t for tab in tabs for t in tab], *Ws, *gnw, *gnb, *gms)
    return y, saved, hs[-1], (tabs[-1][2], tabs[-1][3], gnb[-1])


# Round 6: a fused layer's table sums and the GraphNorm table kernel that consumes them as ONE post
# launch (spt_fused_linear_*_gn_f32: 2 launches per layer and direction instead of 3 / 4; the same
# sums in the same order, the same formulas - tests pin the two routes on each other bitwise).
FUSE_POST = os.environ.get("SPT_FUSE_POST", "1") != "0"


def fuse_post(on=None):
    global FUSE_POST
    old = FUSE_POST
    if on is not None:
        FUSE_POST = bool(on)
    return old


def _fmlp_backward(saved, meta, gy, top_total=None, pooled=None, top_tabs=None):
    """Backward of the fused layer chain from the gradient of its (normalised) output.
    ``top_total``: statistics of the top GraphNorm's backward when the caller already has
    them (the max-pool route computes them from the pool's sparse gradient).
    ``pooled = (gout, arg, csr)``: the top layer consumes the pool's gradient directly
    (``spt_fused_linear_bwd_pooled_runs_f32``), ``gy`` is then None."""
    L, runs, slopes, in_dtype, need_gx0 = meta[:5]
    fmode = meta[5] if len(meta) > 5 else -1          # the matrix mode the forward ran in
    store16 = bool(meta[6]) if len(meta) > 6 else False   # layer outputs stored as bf16
    sv = list(saved)
    x2, batch = sv[0], sv[1]
    hs = sv[2:2 + L]
    tabs = [tuple(sv[2 + L + 4 * i: 2 + L + 4 * i + 4]) for i in range(L)]
    o = 2 + 5 * L
    Ws, gnw, gnb, gms = sv[o:o + L], sv[o + L:o + 2 * L], sv[o + 2 * L:o + 3 * L], sv[o + 3 * L:o + 4 * L]
    R = x2.shape[0]
    dev = x2.device
    B = runs.B
    nr, c_r0, c_r1, c_g = runs.c_arrays()
    sp = _lib.stream_ptr(dev)
    g_cur = gy.contiguous().float() if gy is not None else None
    grads = [None] * (4 * L)
    gx0 = None
    with torch.cuda.device(dev):
        # statistics of the top GraphNorm's backward need their own pass over (h_L, gy)
        N = Ws[-1].shape[0]
        mean, rstd, am, sc = tabs[-1]
        if top_tabs is not None:
            total = None                 # the caller's post launch already wrote the top norm's tables
        elif top_total is not None:
            total = top_total
        else:
            assert not store16, "bf16 storage: the top statistics come from the sparse route"
            total = torch.empty((B, 2 * N + 1), dtype=torch.float64, device=dev)
            ws = _workspace(_lib.lib.spt_graphnorm_workspace_bytes(R, N, B), dev)
            st = _lib.lib.spt_graphnorm_bwd_stats_f32(
                _lib.ptr(hs[-1]), _lib.ptr(g_cur), _lib.ptr(batch) if B > 1 else None, R, N, B,
                _lib.ptr(am), _lib.ptr(sc), _lib.ptr(gnb[-1]), float(slopes[-1]), _lib.ptr(total),
                _lib.ptr(ws), ws.numel(), sp)
            _lib.check(st, "spt_graphnorm_bwd_stats_f32")
        cur_tabs = top_tabs
        for l in range(L - 1, -1, -1):
            N, K = Ws[l].shape
            mean, rstd, am, sc = tabs[l]
            if cur_tabs is not None:
                c1, c2, c3, gw_n, gb_n, ga_n = cur_tabs
            else:
                c1, c2, c3 = (torch.empty((B, N), dtype=torch.float32, device=dev) for _ in range(3))
                gw_n, gb_n, ga_n = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
                st = _lib.lib.spt_graphnorm_bwd_tables_f32(
                    _lib.ptr(total), B, N, _lib.ptr(gnw[l]), _lib.ptr(gms[l]), _lib.ptr(mean),
                    _lib.ptr(rstd), _lib.ptr(c1), _lib.ptr(c2), _lib.ptr(c3), _lib.ptr(gw_n),
                    _lib.ptr(gb_n), _lib.ptr(ga_n), sp)
                _lib.check(st, "spt_graphnorm_bwd_tables_f32")
            xprev = hs[l - 1] if l else x2
            pre = tabs[l - 1] if l else None
            want_gx = l > 0 or need_gx0
            gx = torch.empty((R, K), dtype=torch.float32, device=dev) if want_gx else None
            gW = torch.empty((N, K), dtype=torch.float32, device=dev)
            # the previous layer's backward tables: written by THIS call's post launch (FUSE_POST),
            # or by the next iteration's table call from the totals this call leaves
            nxt_tabs, pn = None, None
            if l and FUSE_POST:
                nxt_tabs = (*(torch.empty((B, K), dtype=torch.float32, device=dev) for _ in range(3)),
                            *(torch.empty(K, dtype=torch.float32, device=dev) for _ in range(3)))
                pn = _lib.GnBwdTables(_lib.ptr(gnw[l - 1]), _lib.ptr(gms[l - 1]), _lib.ptr(tabs[l - 1][0]),
                                      _lib.ptr(tabs[l - 1][1]), *[_lib.ptr(t) for t in nxt_tabs])
            ptot = (torch.empty((B, 2 * K + 1), dtype=torch.float64, device=dev)
                    if (l and pn is None) else None)
            ws = _workspace(_lib.lib.spt_fused_linear_workspace_bytes(K, N), dev)
            pa = ps = pb = None
            if pre is not None:
                pa, ps, pb = pre[2], pre[3], gnb[l - 1]
            lmode = (3 | _ST_H | (_ST_X if l else 0)) if store16 else fmode
            if pooled is not None and l == L - 1:
                p_gout, p_arg, p_csr = pooled
                args = (_lib.ptr(p_gout), _lib.ptr(p_arg), _lib.ptr(p_csr.perm),
                        _lib.ptr(p_csr.pos_seg()), _lib.ptr(hs[l]), nr, c_r0, c_r1, c_g, B, N,
                        _lib.ptr(am), _lib.ptr(sc), _lib.ptr(gnb[l]), float(slopes[l]),
                        _lib.ptr(c1), _lib.ptr(c2), _lib.ptr(c3), _lib.ptr(xprev), K,
                        _lib.ptr(pa), _lib.ptr(ps), _lib.ptr(pb),
                        float(slopes[l - 1]) if l else 1.0, _lib.ptr(Ws[l]), _lib.ptr(gx),
                        _lib.ptr(gW), _lib.ptr(ptot), lmode, _lib.ptr(ws), ws.numel())
                with _timed(f"fused_linear_bwd_pooled:{K}x{N}:{R}"):
                    if pn is not None:
                        st = _lib.lib.spt_fused_linear_bwd_pooled_runs_gn_f32(*args, ctypes.addressof(pn), sp)
                    else:
                        st = _lib.lib.spt_fused_linear_bwd_pooled_runs_f32(*args, sp)
                _lib.check(st, "spt_fused_linear_bwd_pooled_runs_f32")
            else:
                args = (_lib.ptr(g_cur), _lib.ptr(hs[l]), nr, c_r0, c_r1, c_g, B, N,
                        _lib.ptr(am), _lib.ptr(sc), _lib.ptr(gnb[l]), float(slopes[l]),
                        _lib.ptr(c1), _lib.ptr(c2), _lib.ptr(c3), _lib.ptr(xprev), K,
                        _lib.ptr(pa), _lib.ptr(ps), _lib.ptr(pb),
                        float(slopes[l - 1]) if l else 1.0, _lib.ptr(Ws[l]), _lib.ptr(gx),
                        _lib.ptr(gW), _lib.ptr(ptot), lmode, _lib.ptr(ws), ws.numel())
                if pn is not None:
                    st = _lib.lib.spt_fused_linear_bwd_runs_gn_f32(*args, ctypes.addressof(pn), sp)
                else:
                    st = _lib.lib.spt_fused_linear_bwd_runs_f32(*args, sp)
                _lib.check(st, "spt_fused_linear_bwd_runs_f32")
            grads[4 * l], grads[4 * l + 1], grads[4 * l + 2], grads[4 * l + 3] = gW, gw_n, gb_n, ga_n
            if l:
                g_cur, total, cur_tabs = gx, ptot, nxt_tabs
            else:
                gx0 = gx
    return (None if gx0 is None else gx0.to(in_dtype)), grads


class _FusedMLP(torch.autograd.Function):
    """All layers of one MLP in one autograd node: raw layer outputs h_l are the
    only [rows, C] tensors kept; normalised activations are never materialised
    except the final output."""

    @staticmethod
    def forward(ctx, x, batch, runs, eps_list, slope_list, *params):
        fmode = _precision.fused_mode()
        y, saved, _, _ = _fmlp_forward(x, batch, runs, eps_list, slope_list, params, fmode=fmode)
        ctx.save_for_backward(*saved)
        ctx.meta = (len(eps_list), runs, list(slope_list), x.dtype, x.requires_grad, fmode)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        gx0, grads = _fmlp_backward(ctx.saved_tensors, ctx.meta, gy)
        return (gx0, None, None, None, None, *grads)


# the fused MLP + max-pool node lets the pool write the winners' raw values for its backward
# (tests switch it off to pin the two routes on each other)
POOL_RAW_OUTPUT = True

# Round 5: the top layer and the max-pool behind it as ONE algebraic unit (csrc/fused_pool.hip) -
# the layer's [rows, N] output is never written, read or recomputed: the forward pools the raw
# tile out of the matrix-pipe accumulators, the norm's statistics come from the Gram matrix of the
# layer's input, the backward needs (gout, arg, raw) and that input only.  Off: the round-4 route
# (layer output materialised, streaming segment-max, LDS-DMA backward); tests pin one on the other.
POOL_IN_FORWARD = os.environ.get("SPT_POOL_IN_FORWARD", "1") != "0"


def pool_in_forward(on=None):
    """Switch of the pool-fused top layer (default on; ``SPT_POOL_IN_FORWARD=0`` turns it off).
    Returns the previous setting."""
    global POOL_IN_FORWARD
    prev = POOL_IN_FORWARD
    if on is not None:
        POOL_IN_FORWARD = bool(on)
    return prev


class _FusedMLPMaxPool(torch.autograd.Function):
    """The fused MLP followed by a max-pool over segments, as ONE node: the last
    GraphNorm-apply + LeakyReLU run inside the pool's read of the raw h_L (the normalised
    [rows, C] tensor is never written).  Backward = the pool's scatter of the incoming
    gradient to the arg rows, then the MLP backward."""

    @staticmethod
    def forward(ctx, x, batch, runs, eps_list, slope_list, csr, seg_graph, *params):
        fmode = _precision.fused_mode()
        # bf16 mode: the raw layer outputs are STORED as bf16 where every kernel of the chain is
        # built for it (the point MLP in front of the L0 -> L1 pool)
        L = len(eps_list)
        dims = [params[0].shape[1]] + [params[4 * i].shape[0] for i in range(L)]
        store16 = bool(
            _precision.bf16_activation_storage()
            and all(_lib.lib.spt_fused_linear_storage_supported(int(k), int(n))
                    for k, n in zip(dims[:-1], dims[1:]))
            and all(int(k) % 32 == 0 for k in dims[1:-1])
            and 256 % int(dims[-1]) == 0
            and _lib.lib.spt_segcsr_max_affine_bf16_supported(int(dims[-1]), x.shape[0]))
        ctx.pool_fused = False
        mode_top = (3 | _ST_X) if store16 else fmode
        if (POOL_IN_FORWARD and L > 1 and runs.sorted_batch
                and _lib.lib.spt_fused_linear_pool_supported(int(dims[-2]), int(dims[-1]), mode_top)
                and (runs.B == 1 or graph_ranges(seg_graph, runs.B, csr.num_seg) is not None)):
            return _FusedMLPMaxPool._forward_pooled(ctx, x, batch, runs, eps_list, slope_list, csr,
                                                    seg_graph, params, fmode, store16, mode_top)
        _, saved, h_last, (am, sc, bs) = _fmlp_forward(x, batch, runs, eps_list, slope_list,
                                                        params, apply_last=False, fmode=fmode,
                                                        store16=store16)
        R, N = h_last.shape
        dev = h_last.device
        out = torch.empty((csr.num_seg, N), dtype=torch.float32, device=dev)
        arg = torch.empty((csr.num_seg, N), dtype=torch.int32, device=dev)
        # a backward will follow: the pool also writes the winners' raw values, which the top
        # GraphNorm's backward statistics then read as a stream instead of gathering h[arg]
        raw = None
        if (POOL_RAW_OUTPUT and any(ctx.needs_input_grad) and N <= 256 and 256 % N == 0
                and _lib.lib.spt_segcsr_max_affine_raw_supported(int(N), R)):
            raw = torch.empty((csr.num_seg, N), dtype=torch.float32, device=dev)
        # same timer key as the plain segment-max: this IS the L0 -> L1 pool launch
        with torch.cuda.device(dev), _timed(f"segcsr_reduce_fwd:3:{R}x{N}"):
            if raw is not None:
                st = _lib.lib.spt_segcsr_max_affine_raw_f32(
                    _lib.ptr(h_last), 1 if store16 else 0, _lib.ptr(csr.perm), _lib.ptr(csr.rowptr),
                    R, csr.num_seg, N, _lib.ptr(am), _lib.ptr(sc), _lib.ptr(bs),
                    float(slope_list[-1]), _lib.ptr(seg_graph), _lib.ptr(out), _lib.ptr(arg),
                    _lib.ptr(raw), _lib.stream_ptr(dev))
            else:
                seg_max = (_lib.lib.spt_segcsr_max_affine_bf16 if store16
                           else _lib.lib.spt_segcsr_max_affine_f32)
                st = seg_max(
                    _lib.ptr(h_last), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), R, csr.num_seg, N,
                    _lib.ptr(am), _lib.ptr(sc), _lib.ptr(bs), float(slope_list[-1]),
                    _lib.ptr(seg_graph), _lib.ptr(out), _lib.ptr(arg), _lib.stream_ptr(dev))
        _lib.check(st, "spt_segcsr_max_affine")
        ctx.has_raw = raw is not None
        if raw is not None:
            ctx.save_for_backward(arg, *saved, raw)
        else:
            ctx.save_for_backward(arg, *saved)
        ctx.csr = csr
        ctx.seg_graph = seg_graph
        ctx.meta = (len(eps_list), runs, list(slope_list), x.dtype, x.requires_grad,
                    3 if store16 else fmode, store16)
        return out.to(x.dtype)

    @staticmethod
    def _forward_pooled(ctx, x, batch, runs, eps_list, slope_list, csr, seg_graph, params, fmode,
                        store16, mode_top):
        """Layers 0 .. L-2 as a fused chain (their last norm + activation are applied inside the top
        layer's read), then the top layer with the pool in its epilogue
        (``spt_fused_linear_fwd_pool_runs_f32``): out, arg, raw per (segment, channel), the Gram
        totals and the norm's tables - no [rows, N] tensor."""
        L = len(eps_list)
        _, saved_sub, h_prev, (pam, psc, pbs) = _fmlp_forward(
            x, batch, runs, eps_list[:L - 1], slope_list[:L - 1], params[:4 * (L - 1)],
            apply_last=False, fmode=fmode, store16=store16)
        W = params[4 * (L - 1)].detach().float().contiguous()
        gnw, gnb, gms = (params[4 * (L - 1) + i].detach().float().contiguous() for i in (1, 2, 3))
        N, K = W.shape
        R = h_prev.shape[0]
        dev = h_prev.device
        B = runs.B
        S = csr.num_seg
        nr, c_r0, c_r1, c_g = runs.c_arrays()
        out = torch.empty((S, N), dtype=torch.float32, device=dev)
        # (the winners' ORIGINAL rows are not asked for: the fused backward and the sparse
        # statistics work on the CSR positions - `arg` would cost one scattered read of perm per
        # (segment, channel): 55 M of them at scene S, 0.17 ms)
        arg = None
        argpos = torch.empty((S, N), dtype=torch.int32, device=dev)
        raw = torch.empty((S, N), dtype=torch.float32, device=dev)
        glen = int(_lib.lib.spt_fused_linear_pool_gram_len(K))
        gram = torch.empty((B, glen), dtype=torch.float64, device=dev)
        mean, rstd, am, sc = (torch.empty((B, N), dtype=torch.float32, device=dev) for _ in range(4))
        nb = _lib.lib.spt_fused_linear_pool_workspace_bytes(K, N)
        ws = _workspace(nb, dev)
        # same timer key as the segment-max it replaces: this IS the L0 -> L1 pool of the step
        with torch.cuda.device(dev), _timed(f"fused_linear_fwd_pool:{K}x{N}:{R}"):
            st = _lib.lib.spt_fused_linear_fwd_pool_runs_f32(
                _lib.ptr(h_prev), _lib.ptr(csr.perm), _lib.ptr(csr.pos_seg()), _lib.ptr(csr.rowptr),
                _lib.ptr(seg_graph), S, R, nr, c_r0, c_r1, c_g, B, K, _lib.ptr(W), N,
                _lib.ptr(gnw), _lib.ptr(gnb), _lib.ptr(gms), float(eps_list[-1]),
                float(slope_list[-1]), _lib.ptr(pam), _lib.ptr(psc), _lib.ptr(pbs),
                float(slope_list[-2]), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(argpos), _lib.ptr(raw),
                _lib.ptr(gram),
                None, _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(am), _lib.ptr(sc), mode_top,
                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "spt_fused_linear_fwd_pool_runs_f32")
        ctx.pool_fused = True
        ctx.n_sub = len(saved_sub)
        ctx.save_for_backward(argpos, raw, gram, mean, rstd, am, sc, W, gnw, gnb, gms, h_prev, pam, psc,
                              pbs, argpos, *saved_sub)
        ctx.csr = csr
        ctx.seg_graph = seg_graph
        ctx.meta = (L, runs, list(slope_list), x.dtype, x.requires_grad,
                    3 if store16 else fmode, store16)
        ctx.mode_top = mode_top
        return out.to(x.dtype)

    @staticmethod
    def _backward_pooled(ctx, gout):
        (arg, raw, gram, mean, rstd, am, sc, W, gnw, gnb, gms, h_prev, pam, psc,
         pbs, argpos) = ctx.saved_tensors[:16]
        saved_sub = ctx.saved_tensors[16:]
        L, runs, slopes, in_dtype, need_gx0, fmode, store16 = ctx.meta
        N, K = W.shape
        R = h_prev.shape[0]
        dev = h_prev.device
        B = runs.B
        csr = ctx.csr
        S = csr.num_seg
        nr, c_r0, c_r1, c_g = runs.c_arrays()
        sp = _lib.stream_ptr(dev)
        gout = gout.contiguous().float()
        with torch.cuda.device(dev):
            # statistics of the top GraphNorm's backward from the pool's sparse gradient
            total = torch.empty((B, 2 * N + 1), dtype=torch.float64, device=dev)
            rows = _const_tensor(runs.rows_per_graph(), torch.int64, dev)
            nbs = _lib.lib.spt_graphnorm_bwd_stats_sparse_workspace_bytes(S, N, B)
            ws = _workspace(nbs, dev)
            # (`arg` = the CSR positions: the raw variant reads it for the empty-segment sentinel only)
            st = _lib.lib.spt_graphnorm_bwd_stats_sparse_raw_f32(
                _lib.ptr(raw), _lib.ptr(gout), _lib.ptr(arg), _lib.ptr(ctx.seg_graph),
                _lib.ptr(rows), S, R, N, B, _lib.ptr(am), _lib.ptr(sc), _lib.ptr(gnb),
                float(slopes[-1]), _lib.ptr(total), _lib.ptr(ws), nbs, sp)
            _lib.check(st, "spt_graphnorm_bwd_stats_sparse_raw_f32")
            c1, c2, c3 = (torch.empty((B, N), dtype=torch.float32, device=dev) for _ in range(3))
            gw_n, gb_n, ga_n = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(3))
            st = _lib.lib.spt_graphnorm_bwd_tables_f32(
                _lib.ptr(total), B, N, _lib.ptr(gnw), _lib.ptr(gms), _lib.ptr(mean), _lib.ptr(rstd),
                _lib.ptr(c1), _lib.ptr(c2), _lib.ptr(c3), _lib.ptr(gw_n), _lib.ptr(gb_n),
                _lib.ptr(ga_n), sp)
            _lib.check(st, "spt_graphnorm_bwd_tables_f32")
            gm = torch.empty((S, N), dtype=torch.float32, device=dev)
            gx = torch.empty((R, K), dtype=torch.float32, device=dev)
            gW = torch.empty((N, K), dtype=torch.float32, device=dev)
            # the tables of the norm BELOW (layer L-2's backward) out of this call's post launch
            Ls = L - 1
            sub_tabs, pn, ptot = None, None, None
            if FUSE_POST:
                o = 2 + 5 * Ls
                s_gnw, s_gms = saved_sub[o + Ls + Ls - 1], saved_sub[o + 3 * Ls + Ls - 1]
                s_mean, s_rstd = saved_sub[2 + Ls + 4 * (Ls - 1)], saved_sub[2 + Ls + 4 * (Ls - 1) + 1]
                sub_tabs = (*(torch.empty((B, K), dtype=torch.float32, device=dev) for _ in range(3)),
                            *(torch.empty(K, dtype=torch.float32, device=dev) for _ in range(3)))
                pn = _lib.GnBwdTables(_lib.ptr(s_gnw), _lib.ptr(s_gms), _lib.ptr(s_mean), _lib.ptr(s_rstd),
                                      *[_lib.ptr(t) for t in sub_tabs])
            else:
                ptot = torch.empty((B, 2 * K + 1), dtype=torch.float64, device=dev)
            nb = _lib.lib.spt_fused_linear_pool_workspace_bytes(K, N)
            ws = _workspace(nb, dev)
            args = (_lib.ptr(gout), _lib.ptr(raw), _lib.ptr(argpos), _lib.ptr(csr.perm),
                    _lib.ptr(csr.pos_seg()), _lib.ptr(ctx.seg_graph), S, nr, c_r0, c_r1, c_g, B, N,
                    _lib.ptr(am), _lib.ptr(sc), _lib.ptr(gnb), float(slopes[-1]), _lib.ptr(c1),
                    _lib.ptr(c2), _lib.ptr(c3), _lib.ptr(h_prev), K, _lib.ptr(pam), _lib.ptr(psc),
                    _lib.ptr(pbs), float(slopes[-2]), _lib.ptr(W), _lib.ptr(gram), _lib.ptr(gm),
                    _lib.ptr(gx), _lib.ptr(gW), _lib.ptr(ptot), ctx.mode_top, _lib.ptr(ws), ws.numel())
            with _timed(f"fused_linear_bwd_pool:{K}x{N}:{R}"):
                if pn is not None:
                    st = _lib.lib.spt_fused_linear_bwd_pool_runs_gn_f32(*args, ctypes.addressof(pn), sp)
                else:
                    st = _lib.lib.spt_fused_linear_bwd_pool_runs_f32(*args, sp)
            _lib.check(st, "spt_fused_linear_bwd_pool_runs_f32")
        # layers L-2 .. 0: the fused chain's own backward, entered with the gradient of its
        # normalised output and the statistics (or, FUSE_POST, the tables) of its top norm's backward
        meta_sub = (L - 1, runs, list(slopes[:L - 1]), in_dtype, need_gx0, fmode, store16)
        gx0, grads = _fmlp_backward(saved_sub, meta_sub, gx, top_total=ptot, top_tabs=sub_tabs)
        grads = list(grads) + [gW, gw_n, gb_n, ga_n]
        return (gx0, None, None, None, None, None, None, *grads)

    @staticmethod
    def backward(ctx, gout):
        if ctx.pool_fused:
            return _FusedMLPMaxPool._backward_pooled(ctx, gout)
        arg, saved, raw = ctx.saved_tensors[0], ctx.saved_tensors[1:], None
        if ctx.has_raw:
            saved, raw = saved[:-1], saved[-1]
        L, runs, slopes = ctx.meta[0], ctx.meta[1], ctx.meta[2]
        h_last = saved[2 + L - 1]
        am, sc = saved[2 + L + 4 * (L - 1) + 2], saved[2 + L + 4 * (L - 1) + 3]
        gnb_last = saved[2 + 5 * L + 2 * L + (L - 1)]
        R, N = h_last.shape
        dev = h_last.device
        B = runs.B
        gout = gout.contiguous().float()
        # statistics of the top GraphNorm's backward from the pool's SPARSE gradient (one
        # non-zero per (segment, channel)) instead of a pass over the dense [R, N] tensors
        total = None
        if N <= 256 and 256 % N == 0:
            total = torch.empty((B, 2 * N + 1), dtype=torch.float64, device=dev)
            rows = _const_tensor(runs.rows_per_graph(), torch.int64, dev)
            nb = _lib.lib.spt_graphnorm_bwd_stats_sparse_workspace_bytes(ctx.csr.num_seg, N, B)
            ws = _workspace(nb, dev)
            with torch.cuda.device(dev):
                if raw is not None:
                    st = _lib.lib.spt_graphnorm_bwd_stats_sparse_raw_f32(
                        _lib.ptr(raw), _lib.ptr(gout), _lib.ptr(arg), _lib.ptr(ctx.seg_graph),
                        _lib.ptr(rows), ctx.csr.num_seg, R, N, B, _lib.ptr(am), _lib.ptr(sc),
                        _lib.ptr(gnb_last), float(slopes[-1]), _lib.ptr(total), _lib.ptr(ws), nb,
                        _lib.stream_ptr(dev))
                else:
                    st = _lib.lib.spt_graphnorm_bwd_stats_sparse_ex_f32(
                        _lib.ptr(h_last), 1 if h_last.dtype == torch.bfloat16 else 0, _lib.ptr(gout),
                        _lib.ptr(arg), _lib.ptr(ctx.seg_graph),
                        _lib.ptr(rows), ctx.csr.num_seg, R, N, B, _lib.ptr(am), _lib.ptr(sc),
                        _lib.ptr(gnb_last), float(slopes[-1]), _lib.ptr(total), _lib.ptr(ws), nb,
                        _lib.stream_ptr(dev))
            _lib.check(st, "spt_graphnorm_bwd_stats_sparse")
        # The pool's gradient has one non-zero per (segment, channel): the top layer's backward
        # reads (gout, arg) through the pool's CSR order instead of a dense [R, N] tensor.  Needs
        # the top GraphNorm's statistics from the sparse route above, an input gradient (L > 1),
        # a built shape, and - with several graphs - graph-contiguous CSR positions (segments
        # numbered graph by graph, as NAG batches are).
        K_top = saved[2 + 5 * L + (L - 1)].shape[1]
        pooled_ok = (total is not None and L > 1
                     and _lib.lib.spt_fused_linear_pooled_supported_ex(int(K_top), int(N), ctx.meta[5]) == 1
                     and runs.sorted_batch
                     and (B == 1 or graph_ranges(ctx.seg_graph, B, ctx.csr.num_seg) is not None))
        if pooled_ok:
            gx0, grads = _fmlp_backward(saved, ctx.meta, None, top_total=total,
                                        pooled=(gout, arg, ctx.csr))
        else:
            gy = _seg_reduce_bwd(gout, arg, ctx.csr, 3, R)
            gx0, grads = _fmlp_backward(saved, ctx.meta, gy, top_total=total)
        return (gx0, None, None, None, None, None, None, *grads)


def fused_mlp(x, batch, num_graphs, layers):
    """``layers``: list of (linear_weight [N,K], gn_weight, gn_bias, gn_mean_scale, eps,
    act_slope) - ``act_slope = 1`` for a layer without activation.  ``batch`` may be sorted
    (clouds of a batch) or piecewise sorted (<= 16 runs of constant id: the edge MLP's norm
    index): every layer is ONE launch per direction over all graphs.  Returns None when
    the fused path does not apply (more runs / graphs than that, unbuilt shape): the caller
    then runs the layer-by-layer HIP path."""
    if x.dim() != 2 or not x.is_cuda:
        return None
    dims = [layers[0][0].shape[1]] + [l[0].shape[0] for l in layers]
    if not fused_mlp_supported(dims):
        return None
    runs = graph_runs(batch, num_graphs if batch is not None else 1, x.shape[0])
    if runs is None:
        return None
    params = [t for l in layers for t in l[:4]]
    return _FusedMLP.apply(x, batch, runs, [l[4] for l in layers], [l[5] for l in layers], *params)


def fused_mlp_maxpool(x, batch, num_graphs, layers, index, num_seg, seg_graph=None):
    """``max-pool(MLP(x), index)`` with the MLP's last GraphNorm + activation applied inside
    the pool's read (``_FusedMLPMaxPool``).  ``seg_graph`` [num_seg]: graph of every segment
    (needed when the batch holds several graphs).  Returns None when the fused path does not
    apply; the caller then runs MLP and pool separately."""
    if x.dim() != 2 or not x.is_cuda:
        return None
    dims = [layers[0][0].shape[1]] + [l[0].shape[0] for l in layers]
    if not fused_mlp_supported(dims) or dims[-1] % 4 != 0:
        return None
    runs = graph_runs(batch, num_graphs if batch is not None else 1, x.shape[0])
    if runs is None or (runs.B > 1 and seg_graph is None):
        return None
    csr = index if isinstance(index, SegmentCSR) else csr_of(index, num_seg)
    sg = None if runs.B <= 1 else seg_graph.long().contiguous()
    params = [t for l in layers for t in l[:4]]
    return _FusedMLPMaxPool.apply(x, batch, runs, [l[4] for l in layers],
                                  [l[5] for l in layers], csr, sg, *params)
