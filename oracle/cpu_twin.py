"""TEST INFRASTRUCTURE: ctypes binding of oracle/cpu/libspt_cpu.so (OpenMP CPU twins of
``spt_grid_knn_f32`` and ``spt_point_geof_dense_f32``, see oracle/cpu/spt_cpu.cpp).  Imported by
tests/ and by bench.py's cpu_baseline leg only."""
import ctypes
import os

import torch

from .cpu import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB
        if not os.path.exists(path):
            _build.build()
        L = ctypes.CDLL(path)
        p, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.spt_cpu_knn_cell_size.restype = f32
        L.spt_cpu_knn_cell_size.argtypes = [p, i64, i32, f32]
        L.spt_cpu_grid_knn_f32.restype = i32
        L.spt_cpu_grid_knn_f32.argtypes = [p, i64, p, i64, i32, f32, f32, i32, i32, p, p, i32]
        L.spt_cpu_point_geof_f32.restype = i32
        L.spt_cpu_point_geof_f32.argtypes = [p, i64, p, i32, i32, i32, i32, p, i32]
        L.spt_cpu_num_threads.restype = i32
        _lib = L
    return _lib


def num_threads():
    return lib().spt_cpu_num_threads()


def _f32(t):
    return t.detach().cpu().to(torch.float32).contiguous()


def grid_knn(query, search, k, r, inclusive=False, squared=True, cell_size=None, threads=0):
    """(idx int64 [nq, k], dist f32 [nq, k]): the contract of spt_grid_knn_f32 on host cores."""
    q, s = _f32(query), _f32(search)
    L = lib()
    if cell_size is None:
        cell_size = L.spt_cpu_knn_cell_size(s.data_ptr(), s.shape[0], int(k), float(r))
    idx = torch.empty((q.shape[0], k), dtype=torch.int64)
    dist = torch.empty((q.shape[0], k), dtype=torch.float32)
    st = L.spt_cpu_grid_knn_f32(q.data_ptr(), q.shape[0], s.data_ptr(), s.shape[0], int(k),
                                float(r), float(cell_size), int(bool(inclusive)),
                                int(bool(squared)), idx.data_ptr(), dist.data_ptr(), int(threads))
    if st != 0:
        raise RuntimeError("spt_cpu_grid_knn_f32: bad arguments")
    return idx, dist


def knn_1(xyz, k, r_max=1.0, threads=0):
    """src/utils/neighbors.py:51-123 without batch / oversample: search k + 1, drop the point itself."""
    idx, dist = grid_knn(xyz, xyz, k + 1, r_max, threads=threads)
    return idx[:, 1:].contiguous(), dist[:, 1:].contiguous()


def point_geof(xyz, nn, k_min=1, add_self=True, post=True, threads=0):
    """f32 [n, 11]: geometry.py's eigenfeatures (k_step = -1) of dense neighbour lists."""
    x = _f32(xyz)
    nb = nn.detach().cpu().to(torch.int64).contiguous()
    out = torch.empty((x.shape[0], 11), dtype=torch.float32)
    st = lib().spt_cpu_point_geof_f32(x.data_ptr(), x.shape[0], nb.data_ptr(), nb.shape[1],
                                      int(bool(add_self)), int(k_min), int(bool(post)),
                                      out.data_ptr(), int(threads))
    if st != 0:
        raise RuntimeError("spt_cpu_point_geof_f32: bad arguments")
    return out
