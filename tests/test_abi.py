"""CPU suite: the C-ABI shared object loads and exports exactly the symbols
include/spt_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(spt_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    from superpoint_transformer_amd import _lib
    syms = header_symbols()
    assert syms, "no declarations parsed from include/*.h"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in sorted(syms):
        assert hasattr(raw, s), f"{s} declared in include/ but not exported"
    # python binding table and header agree both ways
    assert set(_lib.SIGNATURES) == syms


def test_version_and_error_channel():
    from superpoint_transformer_amd import _lib
    assert _lib.lib.spt_version() >= 1000
    # argument validation happens before any device access
    st = _lib.lib.spt_segcsr_reduce_f32(99, None, None, None, 0, 0, 1, None, None, None)
    assert st != 0
    assert "op" in _lib.last_error()


def test_product_never_imports_oracle():
    """No module of the product imports (statically or by name) anything under ``oracle/``.
    The word itself may appear: ``InstanceData.oracle`` mirrors the reference's method of
    that name (src/data/instance.py:690), which has nothing to do with the test oracle."""
    import re
    pkg = os.path.join(ROOT, "superpoint_transformer_amd")
    imp = re.compile(r"^\s*(from\s+[\w.]*oracle[\w.]*\s+import|import\s+[\w., ]*\boracle\b)", re.M)
    dyn = re.compile(r"(import_module|__import__)\(\s*[\"'][\w.]*oracle")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                txt = open(os.path.join(dp, fn)).read()
                assert not imp.search(txt) and not dyn.search(txt), f"{fn} imports the oracle"
                assert "spt_oracle" not in txt and "spt_model" not in txt, \
                    f"{fn} references the oracle's modules"


def test_host_side_dispatch_queries():
    """Host-only entry points the autograd wrappers consult (no GPU needed): which (K, N) have a
    fused layer / a pooled backward per matrix mode, which (H, D, Dv, F) take the edge-lane
    attention backward, and the decomposition ops.edge_attention picks for wider head layouts."""
    import torch
    from superpoint_transformer_amd import _lib, ops
    L = _lib.lib
    assert L.spt_fused_linear_supported(64, 128) and L.spt_fused_linear_supported(12, 32)
    assert not L.spt_fused_linear_supported(64, 100)
    assert L.spt_fused_linear_pooled_supported_ex(64, 128, 1) and L.spt_fused_linear_pooled_supported_ex(32, 64, 3)
    assert not L.spt_fused_linear_pooled_supported_ex(64, 128, 0)          # f32-exact: dense route
    assert L.spt_skinny_dw_supported(128, 256) and not L.spt_skinny_dw_supported(48, 64)
    assert L.spt_edge_attn_bwd_el_supported(16, 4, 4, 32, 2) and L.spt_edge_attn_bwd_el_supported(16, 4, 4, 32, -1)
    assert not L.spt_edge_attn_bwd_el_supported(16, 4, 8, 32, 2)
    assert not L.spt_edge_attn_bwd_el_supported(16, 4, 4, 32, 0)           # VALU precision
    prev = L.spt_fused_linear_bwd_use_dma(-1)
    assert L.spt_fused_linear_bwd_use_dma(0) == prev and L.spt_fused_linear_bwd_use_dma(prev) == 0

    def split(H, Dv, F=32, enc=True):
        qkv = torch.zeros(3, 2 * H * 4 + H * Dv)
        ea = torch.zeros(5, F)
        W = torch.zeros(H * 4, F) if enc else None
        return ops._matrix_pipe_split(qkv, ea, W, W, W, H, 4)

    assert split(16, 4) is None                                            # the built shape itself
    assert split(16, 8) == (1, 2, 8) and split(32, 4) == (2, 1, 4) and split(32, 8) == (2, 2, 8)
    assert split(16, 8, F=18) is None and split(16, 8, enc=False) is None and split(16, 6) is None
    assert split(24, 4) is None


def test_round4_dispatch_queries_and_argument_checks():
    """The host-only queries of round 4's entries and their argument validation (everything here
    returns before any device access): bf16 activation storage shapes, the pre-norm fold of the
    skinny Linears, the pool's raw output, the tile-record format switch of the attention backward,
    the split UnitSphereNorm statistics' workspace, the fused layers' run-table bound."""
    from superpoint_transformer_amd import _lib, ops
    L = _lib.lib
    # point MLP chain 12 -> 32 -> 64 -> 128: every layer has a bf16-storage kernel; odd widths do not
    assert all(L.spt_fused_linear_storage_supported(k, n) for k, n in ((12, 32), (32, 64), (64, 128)))
    assert not L.spt_fused_linear_storage_supported(64, 100)
    # pre-norm folded into the qkv Linear: K in {32, 64, 128}, N in slabs of 64, the tables of all
    # graphs in LDS
    assert L.spt_skinny_pre_supported(64, 192, 4) and L.spt_skinny_pre_supported(128, 384, 1)
    assert not L.spt_skinny_pre_supported(48, 192, 1) and not L.spt_skinny_pre_supported(64, 100, 1)
    assert not L.spt_skinny_pre_supported(64, 192, 64)
    # streaming pool with the raw output / bf16 rows: 128 channels, >= 65 536 rows
    assert L.spt_segcsr_max_affine_raw_supported(128, 70_000) == 1
    assert L.spt_segcsr_max_affine_raw_supported(64, 70_000) == 0
    assert L.spt_segcsr_max_affine_raw_supported(128, 1_000) == 0
    assert L.spt_segcsr_max_affine_bf16_supported(128, 70_000) == 1
    st = L.spt_segcsr_max_affine_raw_f32(None, 0, None, None, 1_000, 10, 128, None, None, None, 0.01,
                                         None, None, None, None, None)
    assert st != 0 and "raw output" in _lib.last_error()
    # tile records of the attention backward: 64 ints per 16 edges in target order, 48 in source order
    prev = L.spt_attn_bwd_el_target_order(1)
    try:
        assert L.spt_attn_tile_record_ints() == 64
        st = L.spt_attn_pack_tile_ids(None, 8, 8, 16, 8, None)   # (never dereferenced) the 48-int packer refuses
        assert st != 0 and "spt_attn_pack_tile_ids_ex" in _lib.last_error()
        st = L.spt_attn_pack_tile_ids_ex(None, None, None, None, 16, None, None)
        assert st != 0 and "null pointer" in _lib.last_error()
        L.spt_attn_bwd_el_target_order(0)
        assert L.spt_attn_tile_record_ints() == 48
    finally:
        L.spt_attn_bwd_el_target_order(prev)
    # one huge segment (idx = None at the top level) is split over many workgroups: the workspace
    # grows with the slices, many small segments need none beyond the fixed part
    one = L.spt_unit_sphere_workspace_bytes(178_571, 1)
    many = L.spt_unit_sphere_workspace_bytes(15_000_000, 428_571)
    assert one > 256 + 64 and one < (1 << 20)
    assert many >= 428_571 * 8
    # the run table of one fused-layer launch
    assert ops.MAX_FUSED_RUNS == 16
    prev = L.spt_fused_linear_fwd_use_x3(-1)
    assert L.spt_fused_linear_fwd_use_x3(0) == prev and L.spt_fused_linear_fwd_use_x3(prev) == 0
    prev = L.spt_attn_bwd_el_full_line(-1)
    assert L.spt_attn_bwd_el_full_line(prev) == prev
