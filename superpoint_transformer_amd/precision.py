"""Matrix-pipe precision of the hot path's GEMMs (the reference's ``precision: 32 | bf16``
switch, configs/trainer/gpu.yaml:7-10).

``"f32"`` (default)  the reference's SHIPPED arithmetic class: configs/train.yaml:60-61 sets
                     ``float32_matmul_precision: high`` (src/train.py:94 hands it to
                     ``torch.set_float32_matmul_precision``), i.e. every f32 matmul of the
                     reference's run may use TF32 or the "bf16x3" scheme - 3 bf16 products per f32
                     product - next to ``precision: 32`` (configs/trainer/gpu.yaml:7-10).  Here:
                     attention split-bf16 = that bf16x3 scheme (hi*hi + lo*hi + hi*lo, ~2^-17
                     relative per product: 17 of f32's 24 bits, 7 more than TF32's 10); fused MLP
                     layers and (round 6) the tall-skinny Linears of the attention blocks: forward
                     as the f32-EXACT 3-way split (6 bf16 products: tighter than "high" asks for),
                     backward split-bf16.  Every f32 parity bar of tests/ holds
                     in this mode (the bars are set on tolerances, not on this knob).
``"bf16"``           what ``torch.autocast(bfloat16)`` does to the reference's Linear layers:
                     operands rounded to bf16, f32 accumulate, and - for the point MLP, whose
                     [N0, 32..128] layer outputs are most of a step's activation bytes - the raw
                     layer outputs STORED as bf16 (read back by the next layer, the segment
                     max-pool and the backward); parameters, norm statistics, gradients, softmax
                     and segment reductions stay f32 (master weights in f32, like Lightning's
                     bf16-mixed).  Tested at rtol 2e-2.
``"f32-exact"``      f32 matrix pipe everywhere (1/16 of the bf16 pipe's rate): the reference under
                     ``float32_matmul_precision: highest``, which it does not ship.

The mode travels PER CALL: ``matrix_precision(mode)`` (a context manager on a ``contextvars``
variable: per thread / task) and ``SPT(..., matrix_precision=...)`` make the autograd wrappers of
``ops`` pass the mode word to the ``*_ex`` entry points of the C ABI, which read no process-wide
state - two models at different precisions, on different streams or threads, never change each
other's arithmetic (tests/test_modes_gpu.py).  The backward of an op runs in the mode its forward
ran in.  ``set_matrix_precision`` is the older process-wide default (the library's own setters):
it applies wherever no per-call mode is active.
"""
import contextlib
import contextvars
import os

from . import _lib

_MODES = {"f32": (2, 1, 1), "bf16": (3, 3, 3), "f32-exact": (1, 0, 0)}   # (attention, fused MLP, skinny Linears)
_default = "f32"
_active = contextvars.ContextVar("spt_matrix_precision", default=None)


def _check(mode):
    if mode not in _MODES:
        raise ValueError(f"precision must be one of {sorted(_MODES)}")


def set_matrix_precision(mode):
    """Process-wide default (the library's setters); returns the previous mode name."""
    global _default
    _check(mode)
    a, m, k = _MODES[mode]
    _lib.lib.spt_attn_use_mfma(a)
    _lib.lib.spt_fused_linear_use_split_bf16(m)
    _lib.lib.spt_skinny_use_split_bf16(k)
    prev, _default = _default, mode
    return prev


def get_matrix_precision():
    return _active.get() or _default


_ORDERS = {"target": 1 << 6, "source": 2 << 6}       # SPT_ATTN_BWD_TARGET_ORDER / _SOURCE_ORDER
_order = contextvars.ContextVar("spt_attn_bwd_order", default=None)


def attention_mode():
    """Mode word for ``spt_edge_attn_*_ex_f32``: -1 (library defaults) unless a per-call precision
    or a per-call edge order of the attention backward is active (bits 0-1 precision, bits 6-7
    edge order; include/spt_hip.h)."""
    mode, order = _active.get(), _order.get()
    if mode is None and order is None:
        return -1
    prec = _MODES[mode][0] if mode is not None else int(_lib.lib.spt_attn_use_mfma(-2))
    return prec | (_ORDERS[order] if order is not None else 0)


@contextlib.contextmanager
def attention_backward_order(order):
    """Per-call edge order of the edge-lane attention backward for everything launched inside the
    block (this thread / task only; the backward of an op runs in the order its forward saw):
    ``"target"`` - dk / dv reduced per target inside the tile, a few float atomics per node (the
    default of the library); ``"source"`` - no atomics, every gradient bitwise reproducible run to
    run (the deterministic option; +25 % on the level-1 backward).  ``None``: the process default."""
    if order is not None and order not in _ORDERS:
        raise ValueError(f"order must be one of {sorted(_ORDERS)} or None")
    token = _order.set(order)
    try:
        yield
    finally:
        _order.reset(token)


def skinny_mode():
    """Mode word for the ``spt_skinny_*_m_f32`` entries (the attention block's qkv / out_proj Linears,
    the FFN, the heads' input gradients): -1 (the library's default) unless a per-call precision is
    active."""
    mode = _active.get()
    return -1 if mode is None else _MODES[mode][2]


def fused_mode():
    """Mode word for ``spt_fused_linear_*_ex_f32``."""
    mode = _active.get()
    return -1 if mode is None else _MODES[mode][1]


_bf16_storage = os.environ.get("SPT_BF16_STORAGE", "1") != "0"


def set_bf16_activation_storage(on):
    """In the ``"bf16"`` mode the raw outputs of the point MLP's fused layers are STORED as bf16
    (what ``torch.autocast(bfloat16)`` makes of the reference's Linear outputs,
    configs/trainer/gpu.yaml:7-10): on by default (``SPT_BF16_STORAGE=0`` in the environment or
    this switch turn it off, leaving operand rounding only).  Returns the previous setting."""
    global _bf16_storage
    prev, _bf16_storage = _bf16_storage, bool(on)
    return prev


def bf16_activation_storage():
    """True when fused layers launched now should store their activations as bf16."""
    return _bf16_storage and get_matrix_precision() == "bf16"


@contextlib.contextmanager
def matrix_precision(mode):
    """Per-call mode for everything launched inside the block (this thread / task only)."""
    if mode is not None:
        _check(mode)
    token = _active.set(mode)
    try:
        yield
    finally:
        _active.reset(token)
