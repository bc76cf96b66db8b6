"""Reader of the reference's on-disk NAG format (HDF5) without h5py.

Layout (``NAG.save`` src/data/nag.py:401-432 -> ``Data.save`` src/data/data.py:663-733,
``CSRData.save`` src/data/csr.py, ``save_dense_to_csr`` src/utils/io.py:180-202):

    (prefix: NAG._data_serialization_prefix = 'level_', nag.py:43)
    /level_<i>/<key>                      dense tensor of level i (ints stored in the
                                              smallest dtype that holds them, io.py:17-44)
    /level_<i>/_cluster_/sub/pointers     Cluster CSR over the level below
    /level_<i>/_cluster_/sub/value_0      (its points)
    /level_<i>/_csr_/y/{pointers,columns,values,shape}   label histograms, CSR-compressed
    /level_<i>/_not_indexable_            names of the non-node attributes

``load_nag(path)`` mirrors ``NAG.load`` (nag.py:434-461) for the keys of the hot path and
returns a :class:`superpoint_transformer_amd.data.NAG`; integers come back as int64 (the
reference's ``non_fp_to_long`` / ``NAGCast``), ``rgb`` as float in [0, 1] (data.py:830-833).

The HDF5 C library is bound with ctypes (``libhdf5.so`` ships in this image under
/opt/conda/lib; ``SPT_LIBHDF5`` overrides the path).  This is I/O plumbing on the host: nothing
here is a kernel, and nothing falls back silently - a missing library raises.
"""
import ctypes
import os

import numpy as np
import torch

_LIB = None
_HID = ctypes.c_int64          # hid_t of HDF5 >= 1.10
_CANDIDATES = ("libhdf5.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so")


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    paths = ([os.environ["SPT_LIBHDF5"]] if "SPT_LIBHDF5" in os.environ else []) + list(_CANDIDATES)
    err = None
    for p in paths:
        try:
            lib = ctypes.CDLL(p)
            break
        except OSError as e:
            err = e
    else:
        raise ImportError(f"libhdf5 not found (tried {paths}): {err}")
    maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise ImportError(f"libhdf5 {maj.value}.{mnr.value} is older than 1.10 (32-bit hid_t)")
    sig = {
        "H5open": (ctypes.c_int, []),
        "H5Fopen": (_HID, [ctypes.c_char_p, ctypes.c_uint, _HID]),
        "H5Fclose": (ctypes.c_int, [_HID]),
        "H5Gopen2": (_HID, [_HID, ctypes.c_char_p, _HID]),
        "H5Gclose": (ctypes.c_int, [_HID]),
        "H5Gget_num_objs": (ctypes.c_int, [_HID, ctypes.POINTER(ctypes.c_uint64)]),
        "H5Gget_objname_by_idx": (ctypes.c_ssize_t, [_HID, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]),
        "H5Gget_objtype_by_idx": (ctypes.c_int, [_HID, ctypes.c_uint64]),
        "H5Dopen2": (_HID, [_HID, ctypes.c_char_p, _HID]),
        "H5Dclose": (ctypes.c_int, [_HID]),
        "H5Dget_space": (_HID, [_HID]),
        "H5Dget_type": (_HID, [_HID]),
        "H5Dread": (ctypes.c_int, [_HID, _HID, _HID, _HID, _HID, ctypes.c_void_p]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [_HID]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [_HID, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
        "H5Sclose": (ctypes.c_int, [_HID]),
        "H5Tget_class": (ctypes.c_int, [_HID]),
        "H5Tget_size": (ctypes.c_size_t, [_HID]),
        "H5Tget_sign": (ctypes.c_int, [_HID]),
        "H5Tclose": (ctypes.c_int, [_HID]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise ImportError("H5open failed")
    _LIB = lib
    return lib


_H5G_GROUP, _H5G_DATASET = 0, 1
_H5T_INTEGER, _H5T_FLOAT = 0, 1


def _native(lib, name):
    return _HID.in_dll(lib, name).value


def _read_dataset(lib, loc, name):
    d = lib.H5Dopen2(loc, name.encode(), 0)
    if d < 0:
        raise OSError(f"cannot open dataset {name}")
    try:
        sp, tp = lib.H5Dget_space(d), lib.H5Dget_type(d)
        try:
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (ctypes.c_uint64 * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            cls, size, sign = lib.H5Tget_class(tp), lib.H5Tget_size(tp), lib.H5Tget_sign(tp)
            if cls == _H5T_FLOAT:
                np_dt = {2: np.float16, 4: np.float32, 8: np.float64}[size]
                mem = {4: "H5T_NATIVE_FLOAT_g", 8: "H5T_NATIVE_DOUBLE_g"}.get(size)
                if mem is None:                       # half floats: read as f32
                    np_dt, mem = np.float32, "H5T_NATIVE_FLOAT_g"
            elif cls == _H5T_INTEGER:
                np_dt = np.dtype(f"{'i' if sign else 'u'}{size}")
                mem = f"H5T_NATIVE_{'' if sign else 'U'}INT{8 * size}_g"
            else:
                return None                           # strings / compounds: metadata, skipped
            out = np.empty(shape, dtype=np_dt)
            if out.size:
                st = lib.H5Dread(d, _native(lib, mem), 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p))
                if st < 0:
                    raise OSError(f"H5Dread failed on {name}")
            return out
        finally:
            lib.H5Sclose(sp)
            lib.H5Tclose(tp)
    finally:
        lib.H5Dclose(d)


def _walk(lib, loc, prefix, out):
    n = ctypes.c_uint64()
    lib.H5Gget_num_objs(loc, ctypes.byref(n))
    buf = ctypes.create_string_buffer(1024)
    for i in range(n.value):
        lib.H5Gget_objname_by_idx(loc, i, buf, 1024)
        name = buf.value.decode()
        kind = lib.H5Gget_objtype_by_idx(loc, i)
        if kind == _H5G_GROUP:
            g = lib.H5Gopen2(loc, name.encode(), 0)
            try:
                _walk(lib, g, prefix + name + "/", out)
            finally:
                lib.H5Gclose(g)
        elif kind == _H5G_DATASET:
            a = _read_dataset(lib, loc, name)
            if a is not None:
                out[prefix + name] = a


def read_h5(path):
    """Every numeric dataset of the file as ``{'group/sub/name': ndarray}``."""
    lib = _lib()
    f = lib.H5Fopen(os.fsencode(path), 0, 0)          # H5F_ACC_RDONLY, H5P_DEFAULT
    if f < 0:
        raise OSError(f"cannot open {path} as HDF5")
    out = {}
    try:
        _walk(lib, f, "", out)
    finally:
        lib.H5Fclose(f)
    return out


def _tensor(a, device):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if not t.is_floating_point():
        t = t.long()                                   # io.py load_tensor(non_fp_to_long=True)
    return t.to(device)


def load_nag(path, device="cpu", low=0, high=-1, keys=None):
    """``NAG.load`` (src/data/nag.py:434-461): levels ``low..high`` of the file.  Per level:
    dense keys as tensors, ``sub`` as :class:`Cluster` (from ``_cluster_/sub``), ``y`` densified
    from its CSR form (io.py:205-260), ``rgb`` rescaled to [0, 1], instance annotations
    (``_instance_data_/obj``) as :class:`InstanceData`."""
    return nag_from_datasets(read_h5(path), device=device, low=low, high=high, keys=keys,
                             source=path)


def nag_from_datasets(flat, device="cpu", low=0, high=-1, keys=None, source="<datasets>"):
    """The same from the ``{"level_i/...": ndarray}`` table ``read_h5`` returns."""
    from .data import NAG, Cluster, Data
    from .instance import InstanceData
    path = source
    names = sorted({k.split("/")[0] for k in flat if k.startswith("level_")},
                   key=lambda s: int(s.split("_")[1]))
    if not names:
        raise ValueError(f"{path}: no /level_<i> groups (not a NAG file)")
    nlev = len(names)
    high = nlev - 1 if high < 0 else min(high, nlev - 1)
    levels = []
    for i in range(low, high + 1):
        pre = f"level_{i}/"
        attrs = {}
        for k, a in flat.items():
            if not k.startswith(pre):
                continue
            rest = k[len(pre):]
            if "/" in rest or rest == "_not_indexable_":
                continue
            if keys is not None and rest not in keys:
                continue
            t = _tensor(a, device)
            if rest in ("rgb", "mean_rgb") and a.dtype == np.uint8:
                t = t.float() / 255                    # data.py:830-833
            attrs[rest] = t
        cp = pre + "_cluster_/sub/"
        if cp + "pointers" in flat and i > low:        # nag.py:452-458: `sub` of the lowest
            attrs["sub"] = Cluster(_tensor(flat[cp + "pointers"], device),    # loaded level is dropped
                                   _tensor(flat[cp + "value_0"], device))
        yp = pre + "_csr_/y/"
        if yp + "pointers" in flat and (keys is None or "y" in keys):
            ptr = flat[yp + "pointers"].astype(np.int64)
            shape = tuple(int(v) for v in flat[yp + "shape"])
            y = np.zeros(shape, dtype=np.int64)
            rows = np.repeat(np.arange(shape[0]), ptr[1:] - ptr[:-1])
            y[rows, flat[yp + "columns"].astype(np.int64)] = flat[yp + "values"].astype(np.int64)
            attrs["y"] = torch.from_numpy(y).to(device)
        ip = pre + "_instance_data_/"                  # data.py:716-718, csr.py:456-490:
        for name in sorted({k[len(ip):].split("/")[0] for k in flat if k.startswith(ip)}):
            if keys is not None and name not in keys:  # pointers + value_0..2 = obj, count, y
                continue
            grp = ip + name + "/"
            attrs[name] = InstanceData(*(_tensor(flat[grp + d], device) for d in
                                         ("pointers", "value_0", "value_1", "value_2")))
        levels.append(Data(**attrs))
    return NAG(levels)
