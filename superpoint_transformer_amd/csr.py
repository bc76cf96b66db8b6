"""CSR views of unsorted segment indices, built once and reused.

The reference regroups rows by index inside every torch_scatter call
(src/nn/pool.py:61-62, src/nn/norm.py:118-126, src/nn/attention.py:307-315).
Here ``csr_of(index, num_seg)`` runs the device radix sort
(`spt_csr_build`) once and memoises the result ON the index tensor object, so
the pool, UnitSphereNorm, unpool-backward and softmax-group uses of one
``super_index`` share it.  The memo is invalidated by in-place edits
(``Tensor._version``) and dies with the tensor object - it is never keyed on
``data_ptr`` alone (the caching allocator recycles addresses).
"""
import weakref

import torch

from . import _lib

_ATTR = "_spt_csr_memo"
_ATTR_BAD = "_spt_csr_rejected"     # a (sub, super_index) pair that failed adopt_csr's check


class SegmentCSR:
    """(perm, rowptr) view of ``idx``: rows of segment ``s`` are
    ``perm[rowptr[s]:rowptr[s+1]]`` in ascending original order."""

    __slots__ = ("idx", "perm", "rowptr", "n", "num_seg", "_pos_seg")

    def __init__(self, idx, perm, rowptr, n, num_seg):
        self.idx = idx
        self.perm = perm
        self.rowptr = rowptr
        self.n = n
        self.num_seg = num_seg
        self._pos_seg = None

    def pos_seg(self):
        """int32 [n]: segment of every CSR position (``idx[perm]``), built on first use."""
        if self._pos_seg is None:
            out = torch.empty(max(self.n, 1), dtype=torch.int32, device=self.perm.device)
            with torch.cuda.device(out.device):
                st = _lib.lib.spt_csr_pos_seg(_lib.ptr(self.rowptr), self.num_seg, self.n,
                                              _lib.ptr(out), _lib.stream_ptr(out.device))
            _lib.check(st, "spt_csr_pos_seg")
            self._pos_seg = out[:self.n]
        return self._pos_seg

    def counts(self):
        return (self.rowptr[1:] - self.rowptr[:-1])


def build_csr(idx, num_seg):
    """Run the device sort. ``idx``: int64 [n] on the GPU, values in [0, num_seg)."""
    _lib.require_cuda(idx)
    if idx.dim() != 1:
        raise ValueError("segment index must be 1-D")
    if idx.dtype != torch.int64:
        idx = idx.long()
    idx = idx.contiguous()
    n = idx.numel()
    num_seg = int(num_seg)
    if num_seg < 1:
        num_seg = 1
    dev = idx.device
    perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    rowptr = torch.empty(num_seg + 1, dtype=torch.int32, device=dev)
    ws_bytes = _lib.lib.spt_csr_build_workspace_bytes(n, num_seg)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_csr_build(
            _lib.ptr(idx), n, num_seg, _lib.ptr(perm), _lib.ptr(rowptr),
            _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(st, "spt_csr_build")
    # keep an ALIAS of the index (same storage, new tensor object): the memo dict hangs on the
    # caller's tensor object, so holding that object here would close a reference cycle
    # (tensor -> memo -> csr -> tensor) and leave every batch's views to the cyclic GC
    return SegmentCSR(idx.detach(), perm[:n], rowptr, n, num_seg)


def csr_of(idx, num_seg=None):
    """Memoised CSR of ``idx``.  ``num_seg=None`` costs a host sync
    (``idx.max()+1``), exactly like torch_scatter's ``dim_size=None``."""
    if isinstance(idx, SegmentCSR):
        return idx
    if num_seg is None:
        num_seg = int(idx.max().item()) + 1 if idx.numel() > 0 else 1
    num_seg = max(int(num_seg), 1)
    memo = getattr(idx, _ATTR, None)
    key = (idx._version, num_seg, idx.data_ptr(), idx.numel())
    if memo is not None and key in memo:
        return memo[key]
    csr = build_csr(idx, num_seg)
    if memo is None or any(k[0] != idx._version for k in memo):
        memo = {}
        try:
            setattr(idx, _ATTR, memo)
        except Exception:  # tensors that refuse attributes: just don't cache
            return csr
    memo[key] = csr
    return csr


_USE_SUB_VIEWS = True
_USE_MIRROR = True


def use_mirror_views(on=True):
    """Whether the attention backward takes its by-target edge stream from the mirror structure
    an edge list's maker declared (``EdgeCSR``; default) or always sorts the targets.  Returns the
    old setting."""
    global _USE_MIRROR
    old, _USE_MIRROR = _USE_MIRROR, bool(on)
    return old


def use_sub_views(on=True):
    """Whether ``adopt_csr`` installs the CSR a NAG level already carries (``nag[i+1].sub``)
    as the view of ``nag[i].super_index`` (default) or every view is rebuilt by the device
    sort (the worst case: a batch whose levels carry no ``sub``).  Returns the old setting."""
    global _USE_SUB_VIEWS
    old, _USE_SUB_VIEWS = _USE_SUB_VIEWS, bool(on)
    return old


class StaleCSRError(RuntimeError):
    """A stored ``sub`` adopted as the CSR view of a ``super_index`` turned out not to describe it."""


_PENDING = []          # deferred verdicts: (event | None, host flag | device flag, description, idx weakref, key, vkey)
_CHECK_BITS = {1: "pointers do not run 0 .. n monotonically", 2: "a point id outside [0, n)",
               4: "membership: idx[points[j]] is not the cluster holding position j",
               8: "the points of a cluster are not in ascending order"}
_ORDER_ONLY = 8        # a consistent partition whose clusters do not ascend: legitimate, just not the sort's view
_PINNED = {}           # ring of pinned int32 verdict slots (one hipHostMalloc per process, not per batch)


def _flag_text(v):
    return "; ".join(t for b, t in _CHECK_BITS.items() if v & b) or "ok"


def _pinned_slot():
    ring = _PINNED.get("ring")
    if ring is None:
        ring = _PINNED["ring"] = torch.zeros(256, dtype=torch.int32).pin_memory()
        _PINNED["next"] = 0
    busy = {h.data_ptr() for ev, h, *_ in _PENDING if ev is not None}
    for _ in range(ring.numel()):
        i = _PINNED["next"]
        _PINNED["next"] = (i + 1) % ring.numel()
        slot = ring[i:i + 1]
        if slot.data_ptr() not in busy:
            return slot
    return torch.zeros(1, dtype=torch.int32).pin_memory()     # > 256 verdicts in flight


_MIRROR_BITS = {1: "a pair (i, i + M) that is not (s, t) / (t, s)", 2: "a self loop with s != t"}
MIRROR_ATTR = "_spt_mirror_pairs"   # host knowledge left on an edge_index by its maker: M (see EdgeCSR)


def _drop_view(ref, key, vkey, remember):
    """Take a failed pair's view off its index tensor; ``remember``: the next ``adopt_csr`` of the
    same pair answers None at once (the level goes to the sort)."""
    idx = ref() if ref is not None else None
    if idx is None:
        return
    memo = getattr(idx, _ATTR, None)
    if memo is not None:
        memo.pop(key, None)
    if remember:
        try:
            setattr(idx, _ATTR_BAD, vkey)
        except Exception:
            pass


def verify_adopted(block=False):
    """Read the verdicts of the deferred ``adopt_csr`` checks that have COMPLETED (``block``: wait
    for all of them).  A view that failed on pointers, point range or membership is taken off its
    index tensor and ``StaleCSRError`` is raised; a consistent partition whose clusters merely do
    not ascend (a legitimate ``sub``: the reference builds Cluster with a non-stable sort,
    src/data/cluster.py:19-77) is dropped silently - the level goes to the device sort from the
    next batch on, the batches it served grouped the right rows (arg-max ties aside).  Never waits
    on the device unless asked to: the flag travels to pinned host memory behind its check kernel,
    an event tells when it has arrived.  Call with ``block=True`` where results leave the device
    (``SPT.forward`` does so outside training) and after the last batch of a training run."""
    keep, failed = [], None
    for item in _PENDING:
        ev, host, what, ref, key, vkey = item
        if ev is None:                  # recorded under stream capture: a device flag, read only on request
            if not block:
                keep.append(item)
                continue
            v = int(host.item())
        else:
            if block:
                ev.synchronize()
            if not ev.query():
                keep.append(item)
                continue
            v = int(host[0])
        if v and key == "mirror":
            # the edge list does not have the mirror structure its maker declared: the tile records
            # of that batch paired wrong edges.  Forget the hint (later batches sort), then raise.
            ei = ref() if ref is not None else None
            if ei is not None:
                for a in (MIRROR_ATTR, _ATTR):
                    if hasattr(ei, a):
                        try:
                            delattr(ei, a)
                        except Exception:
                            pass
            if failed is None:
                failed = (what, v, "; ".join(t for b, t in _MIRROR_BITS.items() if v & b))
            continue
        if v:
            _drop_view(ref, key, vkey, remember=True)
        if v & ~_ORDER_ONLY and failed is None:
            failed = (what, v, None)
    _PENDING[:] = keep
    if failed is not None and failed[2] is not None:
        raise StaleCSRError(
            f"{failed[0]} was declared [i<j | j>i | loops] with mirrored halves ({MIRROR_ATTR}) but is not "
            f"({failed[2]}): the attention backward of that batch paired wrong edges - drop the attribute "
            "or rebuild the edge list (the hint has been removed: later batches sort by target)")
    if failed is not None:
        raise StaleCSRError(
            f"the stored CSR adopted as the view of {failed[0]} does not describe it "
            f"({_flag_text(failed[1])}): the segment kernels grouped wrong rows since that batch (inside "
            "their buffers: the adopted view is clamped) - rebuild the NAG's `sub` or call "
            "csr.use_sub_views(False)")


def adopt_csr(idx, num_seg, pointers, points, ascending=None, verify="now"):
    """Install ``(pointers, points)`` as the memoised CSR view of ``idx`` - no sort.

    The reference's NAG stores, next to every ``super_index``, the same partition as a CSR:
    ``nag[i+1].sub`` with ``sub.points[sub.pointers[c]:sub.pointers[c+1]]`` the children of
    cluster ``c`` (src/data/cluster.py:19-77; kept consistent by ``NAG.select``,
    src/data/nag.py:306-399).  With the children of a cluster in ascending order - what the
    stable sort of ``build_csr`` produces - the two views are the same arrays, so the per-batch
    sort of the level is skipped: ONE kernel (``spt_csr_adopt_i64``) writes the int32 view the
    segment kernels read and checks it on the way.

    Nothing is taken on trust.  The view is adopted only for a contiguous int64 ``idx`` (what
    ``build_csr`` normalises to and the kernels read), and the kernel checks that ``pointers``
    runs 0 .. n without decreasing, that every point id is in range, that ``idx[points[j]]`` is
    the cluster holding position ``j`` (membership) and that the points of every cluster ascend
    STRICTLY - membership + strict ascent + ``points.numel() == n`` make ``points`` a permutation
    (a duplicate inside a cluster fails the ascent, one across clusters fails membership).  The
    ascent is compared whatever ``ascending`` says (one neighbouring load); a ``sub`` known NOT to
    ascend (``ascending=False``) is not adopted at all.  The int32 view is CLAMPED into the
    buffers, so a pair that fails cannot send a segment kernel out of bounds before its verdict
    has been read.

    ``verify``: ``"now"`` (default) reads the verdict before answering - a failing pair returns
    ``None`` (the level falls back to the sort) and is remembered; ``"deferred"`` (the model's
    per-batch hook) adopts at once and lets the verdict travel to pinned host memory behind the
    kernel: it is read - without ever waiting for the device - by a later ``adopt_csr`` /
    ``verify_adopted()`` call (``verify_adopted`` says what happens to a view that failed).  A
    training step so never pays a host round trip for the check."""
    if not _USE_SUB_VIEWS or idx is None or ascending is False:
        return None
    n = idx.numel()
    num_seg = max(int(num_seg), 1)
    if (points.numel() != n or pointers.numel() != num_seg + 1
            or points.device != idx.device or pointers.device != idx.device):
        return None                     # not the same partition / not resident: leave it to the sort
    if idx.dtype != torch.int64 or not idx.is_contiguous() or idx.dim() != 1:
        return None                     # build_csr normalises these; an adopted view cannot
    _lib.require_cuda(idx)
    capturing = torch.cuda.is_current_stream_capturing()
    if _PENDING and not capturing:
        verify_adopted()
    memo = getattr(idx, _ATTR, None)
    key = (idx._version, num_seg, idx.data_ptr(), n)
    if memo is not None and key in memo:
        return memo[key]
    bad = getattr(idx, _ATTR_BAD, None)
    vkey = key + (points.data_ptr(), points._version, pointers.data_ptr(), pointers._version)
    if bad == vkey:
        return None                     # this very pair failed the check before
    dev = idx.device
    pl = points if points.dtype == torch.int64 else points.long()
    ql = pointers if pointers.dtype == torch.int64 else pointers.long()
    pl, ql = pl.contiguous(), ql.contiguous()
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    rowptr = torch.empty(num_seg + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib.spt_csr_adopt_i64(_lib.ptr(idx), _lib.ptr(pl), _lib.ptr(ql), n, num_seg,
                                        _lib.ptr(perm), _lib.ptr(rowptr), _lib.ptr(flag),
                                        _lib.stream_ptr(dev))
    _lib.check(st, "spt_csr_adopt_i64")
    what = f"an index of {n} rows / {num_seg} segments"
    try:
        ref = weakref.ref(idx)
    except TypeError:
        ref = None
    if verify == "deferred" and capturing:
        # inside a captured graph no event can be queried: the device flag stays with the graph's
        # memory and is read by verify_adopted(block=True) whenever the caller asks
        _PENDING.append((None, flag, what, ref, key, vkey))
    elif verify == "deferred":
        host = _pinned_slot()
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        _PENDING.append((ev, host, what, ref, key, vkey))
    elif int(flag.item()):
        try:
            setattr(idx, _ATTR_BAD, vkey)
        except Exception:
            pass
        return None
    csr = SegmentCSR(idx.detach(), perm[:n], rowptr, n, num_seg)
    if memo is None or any(k[0] != idx._version for k in memo):
        memo = {}
        try:
            setattr(idx, _ATTR, memo)
        except Exception:
            return csr
    memo[key] = csr
    return csr


class EdgeCSR:
    """Edges of an attention graph grouped by SOURCE node (edge_index[0] is the
    softmax group and the output row, src/nn/attention.py:207-208,307,315):
    ``erowptr`` [n+1], ``eperm`` [e] (stable), ``tgt_sorted`` [e] int32 =
    edge_index[1] in CSR order."""

    __slots__ = ("erowptr", "eperm", "tgt_sorted", "n", "e", "_view", "_src_sorted", "_tview",
                 "_tile_ids", "_ei", "_mirror", "_ei_ref")

    def __init__(self, erowptr, eperm, tgt_sorted, n, e, view=None, edge_index=None):
        self.erowptr, self.eperm, self.tgt_sorted, self.n, self.e = \
            erowptr, eperm, tgt_sorted, n, e
        self._view = view
        self._src_sorted = None
        self._tview = None
        self._tile_ids = None
        # Host knowledge of the list's layout, left on the tensor by whoever built it
        # (transforms.horizontal_edge_features, the synthetic batches): M = the number of (i < j)
        # pairs of [i<j | j>i | loops], edge i mirrored at i + M.  The target-order tile records
        # then come from the by-source view alone (no second sort); the structure is checked on
        # the device (spt_attn_mirror_prepare) and a wrong hint raises StaleCSRError.
        self._ei = self._mirror = self._ei_ref = None
        m = getattr(edge_index, MIRROR_ATTR, None) if edge_index is not None else None
        if (_USE_MIRROR and m is not None and edge_index._version == 0 and edge_index.dtype == torch.int64
                and edge_index.is_contiguous() and edge_index.dim() == 2 and 0 <= 2 * int(m) <= e):
            self._ei, self._mirror = edge_index.detach(), int(m)
            try:
                self._ei_ref = weakref.ref(edge_index)
            except TypeError:
                pass

    def mirrored(self, mode=-1):
        """True when the target-order tile records of this graph come from the mirror structure
        (no sorted target view is needed by the attention backward in that order)."""
        return self._mirror is not None and int(_lib.lib.spt_attn_tile_record_ints_m(int(mode))) == 64

    def tile_ids(self, mode=-1):
        """int32 [ceil(e / 16), 48 | 64]: the edge-lane attention backward's tile records, built on
        first use in the format the call's ``mode`` word selects (bits 6-7; the process setting
        when it carries no choice): 16 consecutive positions of the edge stream in TARGET order
        (edge rows | targets | sources | source-order positions), or in source (CSR) order
        (edge rows | targets | sources).  One set per format is kept."""
        ints = int(_lib.lib.spt_attn_tile_record_ints_m(int(mode)))   # 64: target order, 48: source order
        if self._tile_ids is None:
            self._tile_ids = {}
        if ints not in self._tile_ids:
            dev = self.tgt_sorted.device
            nt = (self.e + 15) // 16
            out = torch.empty((max(nt, 1), ints), dtype=torch.int32, device=dev)
            src = self.src_sorted()
            if ints == 64 and self._mirror is not None:
                self._pack_mirror(out, src, dev)
                self._tile_ids[ints] = out
                return out
            tperm = self.target_view().perm if ints == 64 else None
            order = (1 << 6) if ints == 64 else (2 << 6)
            with torch.cuda.device(dev):
                st = _lib.lib.spt_attn_pack_tile_ids_m(
                    _lib.ptr(self.eperm), _lib.ptr(self.tgt_sorted), _lib.ptr(src), _lib.ptr(tperm),
                    self.e, order, _lib.ptr(out), _lib.stream_ptr(dev))
            _lib.check(st, "spt_attn_pack_tile_ids_m")
            self._tile_ids[ints] = out
        return self._tile_ids[ints]

    def _pack_mirror(self, out, src, dev):
        """Target-order tile records from the mirror structure: the inverse of ``eperm`` + the
        structure check in one kernel, the records in a second; the verdict is deferred like an
        adopted level view's (``verify_adopted``)."""
        inv = torch.empty(max(self.e, 1), dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        sp = _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            st = _lib.lib.spt_attn_mirror_prepare(_lib.ptr(self._ei), _lib.ptr(self.eperm), self.e,
                                                  self._mirror, _lib.ptr(inv), _lib.ptr(flag), sp)
            _lib.check(st, "spt_attn_mirror_prepare")
            st = _lib.lib.spt_attn_pack_tile_ids_mirror(
                _lib.ptr(self.eperm), _lib.ptr(self.tgt_sorted), _lib.ptr(src), _lib.ptr(inv), self.e,
                self._mirror, _lib.ptr(out), sp)
            _lib.check(st, "spt_attn_pack_tile_ids_mirror")
        what = f"an edge list of {self.e} edges ({self._mirror} pairs)"
        if torch.cuda.is_current_stream_capturing():
            _PENDING.append((None, flag, what, self._ei_ref, "mirror", None))
        else:
            host = _pinned_slot()
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            _PENDING.append((ev, host, what, self._ei_ref, "mirror", None))

    def target_view(self):
        """CSR view of ``tgt_sorted`` over the CSR positions (which positions point INTO node t,
        ascending), built on first use: the attention backward sums dk / dv per target through it
        instead of scattering them with atomics."""
        if self._tview is None:
            self._tview = build_csr(self.tgt_sorted.long(), self.n)
        return self._tview

    def src_sorted(self):
        """int32 [e]: edge_index[0] in CSR order (the source node of every CSR position), built on
        first use - the edge-lane attention backward fetches per-edge node rows through it."""
        if self._src_sorted is None:
            if self._view is not None:
                self._src_sorted = self._view.pos_seg()
            else:
                cnt = (self.erowptr[1:] - self.erowptr[:-1]).long()
                self._src_sorted = torch.repeat_interleave(
                    torch.arange(self.n, device=cnt.device, dtype=torch.int32), cnt)
        return self._src_sorted


def edge_csr_of(edge_index, num_nodes):
    """Memoised :class:`EdgeCSR` of a [2,E] ``edge_index`` (one device sort per
    batch and level, shared by every transformer block of the stage)."""
    if isinstance(edge_index, EdgeCSR):
        return edge_index
    memo = getattr(edge_index, _ATTR, None)
    key = (edge_index._version, int(num_nodes), edge_index.data_ptr(), edge_index.shape[1])
    if memo is not None and key in memo:
        return memo[key]
    view = build_csr(edge_index[0], num_nodes)
    # plumbing: the targets in CSR order (one gather per batch and level)
    e = edge_index.shape[1]
    tgt_sorted = torch.empty(max(e, 1), dtype=torch.int32, device=edge_index.device)[:e]
    tgt64 = edge_index[1].contiguous()
    if tgt64.dtype != torch.int64:
        tgt64 = tgt64.long()
    with torch.cuda.device(edge_index.device):
        st = _lib.lib.spt_csr_gather_i64_i32(_lib.ptr(tgt64), _lib.ptr(view.perm), e,
                                             _lib.ptr(tgt_sorted), _lib.stream_ptr(edge_index.device))
    _lib.check(st, "spt_csr_gather_i64_i32")
    if _PENDING and not torch.cuda.is_current_stream_capturing():
        verify_adopted()              # (a level without `sub` never reaches adopt_csr's own reading)
    ecsr = EdgeCSR(view.rowptr, view.perm, tgt_sorted, int(num_nodes), edge_index.shape[1], view,
                   edge_index=edge_index)
    if memo is None or any(k[0] != edge_index._version for k in memo):
        memo = {}
        try:
            setattr(edge_index, _ATTR, memo)
        except Exception:
            return ecsr
    memo[key] = ecsr
    return ecsr


def forget(*tensors):
    """Drop the memoised CSR views of these index tensors (a new batch would
    carry new tensors; benchmarks reusing one batch call this every step so
    that the per-batch sort stays inside the timed region)."""
    for t in tensors:
        if t is None:
            continue
        # the CSR views AND the run tables / row ranges ops.graph_runs(_via) / graph_ranges
        # memoise on batch vectors and edge indices: a fresh batch pays their read-backs too
        for a in (_ATTR, _ATTR_BAD, "_spt_graph_runs", "_spt_graph_runs_via", "_spt_graph_ranges",
                  "_spt_batch_checked"):
            if hasattr(t, a):
                try:
                    delattr(t, a)
                except Exception:
                    pass
