"""``h5py`` as the reference uses it (src/utils/io.py, src/data/{csr,cluster,instance,data,nag}.py:
``File`` as a context manager, ``Group.__getitem__ / keys / create_group / create_dataset /
__setitem__ / name``, ``Dataset[...] / shape / dtype`` and the root ``attrs``) on the ctypes
binding of libhdf5 in ``h5io`` - so that the reference's own ``NAG.load`` / ``NAG.save`` chain
runs on a box without h5py.

A file opened for reading is parsed once into memory (``h5io.read_h5``); a file opened for
writing collects its datasets and is written when it is closed (``h5io.write_h5``).  Opt-in:
``shims.install(with_h5py=True)`` - a real h5py is never shadowed silently.
"""
import numpy as np

from .. import h5io

__all__ = ["File", "Group", "Dataset"]


class Dataset:
    def __init__(self, name, value):
        self.name = name
        if isinstance(value, np.ndarray) and value.dtype == object:      # variable-length strings
            value = np.array([s.encode("utf-8") for s in value], dtype=object)   # h5py: bytes
        self._a = value

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)
    ndim = property(lambda self: self._a.ndim)
    size = property(lambda self: self._a.size)

    def __getitem__(self, idx):
        if isinstance(idx, tuple) and idx == ():
            return self._a[()]
        return self._a[idx]

    def __len__(self):
        return self._a.shape[0]

    def __iter__(self):
        return iter(self._a)

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)


class _Attrs:
    """Integer attributes of the root group (``start_i_level``, nag.py:427,448)."""

    def __init__(self, file):
        self._f, self._new = file, {}

    def __setitem__(self, key, value):
        self._f._need_write()
        self._new[key] = int(value)

    def _read(self, key):
        if key in self._new:
            return self._new[key]
        if self._f.mode == "r":
            return h5io.read_root_attr(self._f.filename, key)
        return None

    def __getitem__(self, key):
        v = self._read(key)
        if v is None:
            raise KeyError(key)
        return v

    def get(self, key, default=None):
        v = self._read(key)
        return default if v is None else v

    def __contains__(self, key):
        return self._read(key) is not None

    def keys(self):
        return list(self._new)


class Group:
    def __init__(self, file, name):
        self.file, self.name = file, name                    # name: '/', '/level_0', ...

    def _abs(self, key):
        if key.startswith("/"):
            return key.strip("/")
        base = self.name.strip("/")
        return (base + "/" + key.strip("/")) if base else key.strip("/")

    def keys(self):
        pre = self.name.strip("/")
        pre = pre + "/" if pre else ""
        seen = []
        for k in list(self.file._data) + list(self.file._groups):
            if k.startswith(pre) and k != pre.rstrip("/"):
                head = k[len(pre):].split("/")[0]
                if head and head not in seen:
                    seen.append(head)
        return seen

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, key):
        p = self._abs(key)
        return p in self.file._data or p in self.file._groups or any(
            k.startswith(p + "/") for k in self.file._data)

    def __getitem__(self, key):
        p = self._abs(key)
        if p in self.file._data:
            return Dataset("/" + p, self.file._data[p])
        if p in self.file._groups or any(k.startswith(p + "/") for k in self.file._data):
            return Group(self.file, "/" + p)
        raise KeyError(f"Unable to open object (object '{key}' doesn't exist)")

    def create_group(self, name):
        self.file._need_write()
        p = self._abs(name)
        if p in self.file._groups or p in self.file._data:
            raise ValueError(f"Unable to create group (name already exists): {name}")
        parts = p.split("/")
        for i in range(1, len(parts) + 1):
            self.file._groups.setdefault("/".join(parts[:i]), True)
        return Group(self.file, "/" + p)

    def create_dataset(self, name, shape=None, dtype=None, data=None, **unused):
        self.file._need_write()
        p = self._abs(name)
        if p in self.file._data:
            raise ValueError(f"Unable to create dataset (name already exists): {name}")
        if data is None:
            data = np.zeros(shape if shape is not None else (), dtype=dtype or np.float32)
        if isinstance(data, (list, tuple)) and (len(data) == 0 or isinstance(data[0], str)):
            value = list(data)
        else:
            value = np.asarray(data) if dtype is None else np.asarray(data).astype(dtype)
        parent = p.rpartition("/")[0]
        if parent:
            parts = parent.split("/")
            for i in range(1, len(parts) + 1):
                self.file._groups.setdefault("/".join(parts[:i]), True)
        self.file._data[p] = value
        return Dataset("/" + p, value if isinstance(value, np.ndarray) else np.array(value, dtype=object))

    def __setitem__(self, name, value):                      # f['_not_indexable_'] = [...]
        self.create_dataset(name, data=value)


class File(Group):
    def __init__(self, path, mode="r", **unused):
        if mode not in ("r", "w"):
            raise ValueError(f"h5py shim: mode '{mode}' (only 'r' and 'w' are used by the reference)")
        self.filename, self.mode = str(path), mode
        self._groups = {}
        self._data = h5io.read_h5(self.filename, strings=True) if mode == "r" else {}
        self.attrs = _Attrs(self)
        self._open = True
        Group.__init__(self, self, "/")

    def _need_write(self):
        if self.mode != "w":
            raise ValueError("h5py shim: file is open read-only")

    def close(self):
        if self._open and self.mode == "w":
            h5io.write_h5(self.filename, self._data, root_attrs=dict(self.attrs._new),
                          groups=list(self._groups))
        self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
