"""Matrix-pipe precision of the hot path's GEMMs (the reference's ``precision: 32 | bf16``
switch, configs/trainer/gpu.yaml:7-10).

``"f32"`` (default)  attention: split-bf16 (3 bf16 products per f32 product, ~10 ulp of f32);
                     fused MLP layers: forward on the f32 matrix pipe (bitwise an fmaf chain),
                     backward split-bf16.  Every f32 parity bar of tests/ holds in this mode.
``"bf16"``           what ``torch.autocast(bfloat16)`` does to the reference's Linear layers:
                     operands rounded to bf16, f32 accumulate; parameters, activations, norm
                     statistics, softmax and segment reductions stay f32 (master weights in f32,
                     like Lightning's bf16-mixed).  Tested at rtol 2e-2.
``"f32-exact"``      f32 matrix pipe everywhere (1/16 of the bf16 pipe's rate).
"""
import contextlib

from . import _lib

_MODES = {"f32": (2, 1), "bf16": (3, 3), "f32-exact": (1, 0)}   # (attention, fused MLP)
_current = "f32"


def set_matrix_precision(mode):
    """Process-wide; returns the previous mode name."""
    global _current
    if mode not in _MODES:
        raise ValueError(f"precision must be one of {sorted(_MODES)}")
    a, m = _MODES[mode]
    _lib.lib.spt_attn_use_mfma(a)
    _lib.lib.spt_fused_linear_use_split_bf16(m)
    prev, _current = _current, mode
    return prev


def get_matrix_precision():
    return _current


@contextlib.contextmanager
def matrix_precision(mode):
    prev = set_matrix_precision(mode)
    try:
        yield
    finally:
        set_matrix_precision(prev)
