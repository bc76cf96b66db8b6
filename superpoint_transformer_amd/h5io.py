"""Reader and writer of the reference's on-disk NAG format (HDF5) without h5py.

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
        "H5Tis_variable_str": (ctypes.c_int, [_HID]),
        "H5Tcopy": (_HID, [_HID]),
        "H5Tset_size": (ctypes.c_int, [_HID, ctypes.c_size_t]),
        "H5Tset_cset": (ctypes.c_int, [_HID, ctypes.c_int]),
        "H5Tset_fields": (ctypes.c_int, [_HID] + [ctypes.c_size_t] * 5),
        "H5Tset_ebias": (ctypes.c_int, [_HID, ctypes.c_size_t]),
        "H5Dvlen_reclaim": (ctypes.c_int, [_HID, _HID, _HID, ctypes.c_void_p]),
        "H5Fcreate": (_HID, [ctypes.c_char_p, ctypes.c_uint, _HID, _HID]),
        "H5Gcreate2": (_HID, [_HID, ctypes.c_char_p, _HID, _HID, _HID]),
        "H5Lexists": (ctypes.c_int, [_HID, ctypes.c_char_p, _HID]),
        "H5Screate": (_HID, [ctypes.c_int]),
        "H5Screate_simple": (_HID, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.POINTER(ctypes.c_uint64)]),
        "H5Dcreate2": (_HID, [_HID, ctypes.c_char_p, _HID, _HID, _HID, _HID, _HID]),
        "H5Dwrite": (ctypes.c_int, [_HID, _HID, _HID, _HID, _HID, ctypes.c_void_p]),
        "H5Acreate2": (_HID, [_HID, ctypes.c_char_p, _HID, _HID, _HID, _HID]),
        "H5Awrite": (ctypes.c_int, [_HID, _HID, ctypes.c_void_p]),
        "H5Aopen": (_HID, [_HID, ctypes.c_char_p, _HID]),
        "H5Aexists": (ctypes.c_int, [_HID, ctypes.c_char_p]),
        "H5Aread": (ctypes.c_int, [_HID, _HID, ctypes.c_void_p]),
        "H5Aclose": (ctypes.c_int, [_HID]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise ImportError("H5open failed")
    _LIB = lib
    return lib


_H5G_GROUP, _H5G_DATASET = 0, 1
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3


def _native(lib, name):
    return _HID.in_dll(lib, name).value


def _read_dataset(lib, loc, name, strings=False):
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
            elif cls == _H5T_STRING and strings and lib.H5Tis_variable_str(tp) > 0 and nd == 1:
                # variable-length strings (h5py writes a list of str this way: data.py:732)
                ptrs = (ctypes.c_char_p * max(shape[0], 1))()
                if shape[0]:
                    if lib.H5Dread(d, tp, 0, 0, 0, ctypes.cast(ptrs, ctypes.c_void_p)) < 0:
                        raise OSError(f"H5Dread failed on {name}")
                out = np.array([ptrs[i].decode("utf-8") for i in range(shape[0])], dtype=object)
                if shape[0]:
                    lib.H5Dvlen_reclaim(tp, sp, 0, ctypes.cast(ptrs, ctypes.c_void_p))
                return out
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


def _walk(lib, loc, prefix, out, strings=False, select=None):
    n = ctypes.c_uint64()
    lib.H5Gget_num_objs(loc, ctypes.byref(n))
    buf = ctypes.create_string_buffer(1024)
    for i in range(n.value):
        lib.H5Gget_objname_by_idx(loc, i, buf, 1024)
        name = buf.value.decode()
        kind = lib.H5Gget_objtype_by_idx(loc, i)
        if kind == _H5G_GROUP:
            if select is not None and not select(prefix + name + "/"):
                continue                               # a whole group nobody asked for
            g = lib.H5Gopen2(loc, name.encode(), 0)
            try:
                _walk(lib, g, prefix + name + "/", out, strings, select)
            finally:
                lib.H5Gclose(g)
        elif kind == _H5G_DATASET:
            if select is not None and not select(prefix + name):
                continue
            a = _read_dataset(lib, loc, name, strings)
            if a is not None:
                out[prefix + name] = a


def read_h5(path, strings=False, select=None):
    """Every numeric dataset of the file as ``{'group/sub/name': ndarray}``; ``strings``: also
    the 1-D variable-length string datasets (object arrays of str).  ``select(name) -> bool``
    (names of groups end with '/') skips datasets and whole groups without reading them."""
    lib = _lib()
    f = lib.H5Fopen(os.fsencode(path), 0, 0)          # H5F_ACC_RDONLY, H5P_DEFAULT
    if f < 0:
        raise OSError(f"cannot open {path} as HDF5")
    out = {}
    try:
        _walk(lib, f, "", out, strings, select)
    finally:
        lib.H5Fclose(f)
    return out


def read_root_attr(path, name):
    """An integer attribute of the root group (``start_i_level``, nag.py:427), or None."""
    lib = _lib()
    f = lib.H5Fopen(os.fsencode(path), 0, 0)
    if f < 0:
        raise OSError(f"cannot open {path} as HDF5")
    try:
        if lib.H5Aexists(f, name.encode()) <= 0:
            return None
        a = lib.H5Aopen(f, name.encode(), 0)
        try:
            v = ctypes.c_int64()
            if lib.H5Aread(a, _native(lib, "H5T_NATIVE_INT64_g"), ctypes.byref(v)) < 0:
                raise OSError(f"H5Aread failed on {name}")
            return int(v.value)
        finally:
            lib.H5Aclose(a)
    finally:
        lib.H5Fclose(f)


def _tensor(a, device, to_long=True):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if to_long and not t.is_floating_point():
        t = t.long()                                   # io.py load_tensor(non_fp_to_long=True)
    return t.to(device)


def load_nag(path, device="cpu", low=0, high=-1, keys=None, non_fp_to_long=True,
             rgb_to_float=True, keys_low=None):
    """``NAG.load`` (src/data/nag.py:434-461): levels ``low..high`` of the file.  Per level:
    dense keys as tensors, ``sub`` as :class:`Cluster` (from ``_cluster_/sub``), ``y`` densified
    from its CSR form (io.py:205-260), ``rgb`` rescaled to [0, 1], instance annotations
    (``_instance_data_/obj``) as :class:`InstanceData`.  ``non_fp_to_long`` / ``rgb_to_float``
    are the reference's switches of the same name (io.py:83-92, data.py:925-929) with the values
    the device pipeline wants as defaults (the reference defaults to the compressed integer types
    and byte colours and casts later with ``NAGCast``); index tensors of ``sub`` / ``obj`` / the
    CSR histogram are always int64.  ``keys``: the attributes to read (``sub``, ``y``, ``obj``
    count like any other name); ``keys_low``: the same for level ``low`` only (the datasets read
    point features at the lowest level and segment features above: datasets/base.py:1098-1104)."""
    k_low = keys if keys_low is None and keys is not None else keys_low

    def wanted(name):
        """Only the levels and attributes asked for are read off the disk (a 15 M-point level
        holds a dozen feature arrays; a training sample wants a few of them)."""
        parts = name.split("/")
        if not parts[0].startswith("level_"):
            return True
        lvl = int(parts[0].split("_")[1])
        if lvl < low or (high >= 0 and lvl > high):
            return False
        ks = k_low if lvl == low else keys
        if ks is None or len(parts) < 2 or parts[1] == "":
            return True
        if parts[1] in ("_csr_", "_cluster_", "_instance_data_"):
            return len(parts) < 3 or parts[2] == "" or parts[2] in ks
        return parts[1] in ks

    return nag_from_datasets(read_h5(path, select=wanted), device=device, low=low, high=high,
                             keys=keys, source=path, non_fp_to_long=non_fp_to_long,
                             rgb_to_float=rgb_to_float, keys_low=keys_low)


def nag_from_datasets(flat, device="cpu", low=0, high=-1, keys=None, source="<datasets>",
                      non_fp_to_long=True, rgb_to_float=True, keys_low=None):
    """The same from the ``{"level_i/...": ndarray}`` table ``read_h5`` returns."""
    from .data import NAG, Cluster, Data
    from .instance import InstanceData
    path = source
    names = sorted({k.split("/")[0] for k in flat if k.startswith("level_")},
                   key=lambda s: int(s.split("_")[1]))
    if not names:
        raise ValueError(f"{path}: no /level_<i> groups (not a NAG file)")
    top = int(names[-1].split("_")[1])                 # absolute index of the highest level present
    high = top if high < 0 else min(high, top)
    levels = []
    if keys_low is None and keys is not None:          # nag.py:489
        keys_low = keys
    keys_above = keys
    for i in range(low, high + 1):
        pre = f"level_{i}/"
        attrs = {}
        keys = keys_low if i == low else keys_above
        for k, a in flat.items():
            if not k.startswith(pre):
                continue
            rest = k[len(pre):]
            if "/" in rest or rest == "_not_indexable_":
                continue
            if keys is not None and rest not in keys:
                continue
            t = _tensor(a, device, non_fp_to_long)
            if rest in ("rgb", "mean_rgb"):            # data.py:925-929, utils/color.py:17-29
                if rgb_to_float:
                    t = t.float()
                    t = (t / 255 if t.numel() and float(t.max()) > 1 else t).clamp(min=0, max=1)
                else:
                    t = (t * 255 if t.is_floating_point() and (not t.numel() or float(t.max()) <= 1)
                         else t).clamp(min=0, max=255).byte()
            attrs[rest] = t
        cp = pre + "_cluster_/sub/"
        if cp + "pointers" in flat and (keys is None or "sub" in keys):
            # kept at the lowest loaded level too: NAG.get_sub_size counts the points of a
            # level that was not loaded through it (nag.py:73-89)
            attrs["sub"] = Cluster(_tensor(flat[cp + "pointers"], device),
                                   _tensor(flat[cp + "value_0"], device))
        yp = pre + "_csr_/y/"
        if yp + "pointers" in flat and (keys is None or "y" in keys):
            ptr = flat[yp + "pointers"].astype(np.int64)
            shape = tuple(int(v) for v in flat[yp + "shape"])
            vals = flat[yp + "values"]
            if non_fp_to_long and vals.dtype.kind in "iub":
                vals = vals.astype(np.int64)           # else: the stored type (sparse.py:63-90)
            y = np.zeros(shape, dtype=vals.dtype)
            rows = np.repeat(np.arange(shape[0]), ptr[1:] - ptr[:-1])
            y[rows, flat[yp + "columns"].astype(np.int64)] = vals
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


# ---------------------------------------------------------------------------------------------
# Writing: NAG.save (src/data/nag.py:401-432) -> Data.save (src/data/data.py:663-733),
# CSRData.save (src/data/csr.py:456-490), save_tensor / save_dense_to_csr (src/utils/io.py:47-202)
# ---------------------------------------------------------------------------------------------
_FILE_TYPES = {"uint8": "H5T_STD_U8LE_g", "int8": "H5T_STD_I8LE_g", "int16": "H5T_STD_I16LE_g",
               "int32": "H5T_STD_I32LE_g", "int64": "H5T_STD_I64LE_g", "float32": "H5T_IEEE_F32LE_g",
               "float64": "H5T_IEEE_F64LE_g"}


def _half_type(lib):
    """IEEE binary16 the way h5py declares it: a copy of F32LE with 1 + 5 + 10 bit fields."""
    t = lib.H5Tcopy(_native(lib, "H5T_IEEE_F32LE_g"))
    lib.H5Tset_fields(t, 15, 10, 5, 0, 10)
    lib.H5Tset_size(t, 2)
    lib.H5Tset_ebias(t, 15)
    return t


def _ensure_group(lib, f, path, opened):
    """Create the groups along ``a/b/c`` (once) and return the handle of the last one."""
    loc, sofar = f, ""
    for part in path.split("/"):
        sofar = part if not sofar else sofar + "/" + part
        if sofar not in opened:
            g = lib.H5Gcreate2(loc, part.encode(), 0, 0, 0)
            if g < 0:
                raise OSError(f"cannot create group {sofar}")
            opened[sofar] = g
        loc = opened[sofar]
    return loc


def _write_dataset(lib, loc, name, a):
    if isinstance(a, (list, tuple)) or (isinstance(a, np.ndarray) and a.dtype == object):
        items = [str(x) for x in a]
        if not items:                                  # h5py turns [] into an empty f64 dataset
            a = np.zeros(0, dtype=np.float64)
        else:
            tp = lib.H5Tcopy(_native(lib, "H5T_C_S1_g"))
            lib.H5Tset_size(tp, ctypes.c_size_t(-1).value)          # H5T_VARIABLE
            lib.H5Tset_cset(tp, 1)                                  # H5T_CSET_UTF8
            dims = (ctypes.c_uint64 * 1)(len(items))
            sp = lib.H5Screate_simple(1, dims, None)
            d = lib.H5Dcreate2(loc, name.encode(), tp, sp, 0, 0, 0)
            try:
                if d < 0:
                    raise OSError(f"cannot create dataset {name}")
                ptrs = (ctypes.c_char_p * len(items))(*[x.encode("utf-8") for x in items])
                if lib.H5Dwrite(d, tp, 0, 0, 0, ctypes.cast(ptrs, ctypes.c_void_p)) < 0:
                    raise OSError(f"H5Dwrite failed on {name}")
            finally:
                if d >= 0:
                    lib.H5Dclose(d)
                lib.H5Sclose(sp)
                lib.H5Tclose(tp)
            return
    a = np.asarray(a)
    if not a.flags.c_contiguous:                       # (ascontiguousarray would make 0-d 1-d)
        a = a.copy(order="C")
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    own = None
    if a.dtype == np.float16:
        own = tp = _half_type(lib)
    elif a.dtype.name in _FILE_TYPES:
        tp = _native(lib, _FILE_TYPES[a.dtype.name])
    else:
        raise TypeError(f"{name}: dtype {a.dtype} has no HDF5 mapping here")
    dims = (ctypes.c_uint64 * max(a.ndim, 1))(*a.shape)
    sp = lib.H5Screate_simple(a.ndim, dims, None) if a.ndim else lib.H5Screate(0)
    d = lib.H5Dcreate2(loc, name.encode(), tp, sp, 0, 0, 0)
    try:
        if d < 0:
            raise OSError(f"cannot create dataset {name}")
        if a.size and lib.H5Dwrite(d, tp, 0, 0, 0, a.ctypes.data_as(ctypes.c_void_p)) < 0:
            raise OSError(f"H5Dwrite failed on {name}")
    finally:
        if d >= 0:
            lib.H5Dclose(d)
        lib.H5Sclose(sp)
        if own is not None:
            lib.H5Tclose(own)


def write_h5(path, datasets, root_attrs=None, groups=()):
    """``{'group/sub/name': ndarray | list of str}`` -> an HDF5 file (little-endian standard
    types, contiguous layout - what h5py's ``create_dataset(data=...)`` produces); ``root_attrs``:
    integer attributes of the root group; ``groups``: groups to create even if they stay empty."""
    lib = _lib()
    f = lib.H5Fcreate(os.fsencode(path), 2, 0, 0)      # H5F_ACC_TRUNC
    if f < 0:
        raise OSError(f"cannot create {path}")
    opened = {}
    try:
        for name, val in (root_attrs or {}).items():
            sp = lib.H5Screate(0)                      # scalar
            a = lib.H5Acreate2(f, name.encode(), _native(lib, "H5T_STD_I64LE_g"), sp, 0, 0)
            v = ctypes.c_int64(int(val))
            lib.H5Awrite(a, _native(lib, "H5T_NATIVE_INT64_g"), ctypes.byref(v))
            lib.H5Aclose(a)
            lib.H5Sclose(sp)
        for grp in groups:
            _ensure_group(lib, f, grp.strip("/"), opened)
        for key, a in datasets.items():
            grp, _, name = key.rpartition("/")
            loc = _ensure_group(lib, f, grp, opened) if grp else f
            _write_dataset(lib, loc, name, a)
    finally:
        for g in reversed(list(opened.values())):
            lib.H5Gclose(g)
        lib.H5Fclose(f)


def _optimal_int(t):
    """cast_to_optimal_integer_type (src/utils/tensor.py:223-241): the first of uint8, int16,
    int32, int64 that holds every value; empty -> uint8."""
    a = t.detach().cpu().numpy()
    if a.dtype == np.bool_:
        return a.astype(np.uint8)
    if a.size == 0:
        return a.astype(np.uint8)
    lo, hi = int(a.min()), int(a.max())
    for dt in (np.uint8, np.int16, np.int32, np.int64):
        info = np.iinfo(dt)
        if info.min <= lo and hi <= info.max:
            return a.astype(dt)
    raise ValueError(f"Could not cast dtype={a.dtype} to integer.")


def _np_dtype(torch_dtype):
    return {torch.float16: np.float16, torch.float32: np.float32, torch.float64: np.float64,
            torch.half: np.float16, torch.float: np.float32, torch.double: np.float64}[torch_dtype]


def _numpyfy(t, fp_dtype):
    """cast_numpyfy (src/utils/tensor.py:268-285)."""
    if not t.is_floating_point():
        return _optimal_int(t)
    return t.detach().cpu().numpy().astype(_np_dtype(fp_dtype))


def nag_to_datasets(nag, y_to_csr=True, pos_dtype=torch.float, fp_dtype=torch.float,
                    rgb_to_byte=True, start_i_level=0):
    """The dataset table ``NAG.save`` writes (nag.py:401-432, data.py:663-733)."""
    from .data import Cluster
    from .instance import InstanceData
    out = {}
    for i in range(nag.num_levels):
        data = nag[i]
        pre = f"level_{i + start_i_level}/"
        n = data.num_nodes
        n_e = data.num_edges
        not_indexable = []
        for k, val in data:
            if k == "pos_offset":
                out[pre + k] = _numpyfy(val, torch.double)
            elif k == "pos":
                out[pre + k] = _numpyfy(val, pos_dtype)
            elif k == "y" and val.dim() > 1 and y_to_csr:                    # io.py:180-202
                rows, cols = val.nonzero(as_tuple=True)
                ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=val.device),
                                 torch.bincount(rows, minlength=val.shape[0]).cumsum(0)])
                g = pre + "_csr_/y/"
                out[g + "pointers"] = _numpyfy(ptr, fp_dtype)
                out[g + "columns"] = _numpyfy(cols, fp_dtype)
                out[g + "values"] = _numpyfy(val[rows, cols], fp_dtype)
                out[g + "shape"] = np.array(val.shape)
            elif k in ("rgb", "mean_rgb") and rgb_to_byte:
                v = (val * 255).byte() if val.is_floating_point() else val.byte()
                out[pre + k] = v.cpu().numpy()
            elif isinstance(val, Cluster):
                g = pre + f"_cluster_/{k}/"
                out[g + "pointers"] = _numpyfy(val.pointers, fp_dtype)
                out[g + "is_index_value"] = np.array([1], dtype=np.uint8)   # cluster.py: [True]
                out[g + "value_0"] = _numpyfy(val.points, fp_dtype)
            elif isinstance(val, InstanceData):
                g = pre + f"_instance_data_/{k}/"
                out[g + "pointers"] = _numpyfy(val.pointers, fp_dtype)
                out[g + "is_index_value"] = np.array([1, 0, 0], dtype=np.uint8)
                for j, v in enumerate(val.values):
                    out[g + f"value_{j}"] = _numpyfy(v, fp_dtype)
            elif torch.is_tensor(val):
                out[pre + k] = _numpyfy(val, fp_dtype)
            else:
                raise NotImplementedError(
                    f"Cannot save attribute {k} with unsupported type {type(val)}")
            node_sized = torch.is_tensor(val) and val.dim() > 0 and val.shape[0] == n
            if not node_sized or ("edge" in k and not k.startswith("v_edge") and n_e == n):
                not_indexable.append(k)                                      # data.py:730-732
        out[pre + "_not_indexable_"] = not_indexable
    return out


def save_nag(nag, path, y_to_csr=True, pos_dtype=torch.float, fp_dtype=torch.float,
             rgb_to_byte=True, start_i_level=0):
    """``NAG.save`` (src/data/nag.py:401-432): one ``level_<i>`` group per level, tensors in the
    smallest integer type that holds them / ``fp_dtype`` (``pos`` in ``pos_dtype``), label
    histograms CSR-compressed, colours as bytes, ``sub`` / ``obj`` as CSR groups, the names of the
    non-node attributes, and the root attribute ``start_i_level``."""
    write_h5(path, nag_to_datasets(nag, y_to_csr, pos_dtype, fp_dtype, rgb_to_byte, start_i_level),
             root_attrs={"start_i_level": start_i_level})
