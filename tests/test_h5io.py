"""f4 (I/O half): the reference's HDF5 NAG layout (src/data/nag.py:401-461,
src/data/data.py:663-733) read without h5py, against the reference's only real data file -
notebooks/demo_nag_v3.h5 - whose h5py dump is the committed tests/golden/demo_nag_v3.npz."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

H5 = "/root/reference/notebooks/demo_nag_v3.h5"
pytestmark = pytest.mark.skipif(not os.path.exists(H5), reason="reference tree only exists in the build container")


def test_raw_datasets_match_the_h5py_dump():
    from superpoint_transformer_amd import h5io
    got = h5io.read_h5(H5)
    ref = load_golden("demo_nag_v3.npz")
    assert len(got) == len(ref)
    for k, a in got.items():
        r = ref[k.replace("/", "__")]
        assert a.dtype == r.dtype and a.shape == r.shape, k
        assert np.array_equal(a, r), k


def test_load_nag_builds_the_hierarchy():
    from superpoint_transformer_amd import h5io
    nag = h5io.load_nag(H5)
    assert nag.num_levels == 4
    assert nag.num_points == [41568, 1192, 501, 166]
    ref = load_golden("demo_nag_v3.npz")
    assert nag[0].pos.dtype == torch.float32 and nag[0].super_index.dtype == torch.int64
    assert torch.equal(nag[0].super_index, torch.from_numpy(ref["level_0__super_index"]).long())
    assert float(nag[0].rgb.max()) <= 1.0 and nag[0].rgb.dtype == torch.float32
    # the Cluster CSR of level 1 describes exactly the inverse of level 0's super_index
    sub = nag[1].sub
    owner = torch.repeat_interleave(torch.arange(1192), sub.pointers[1:] - sub.pointers[:-1])
    assert torch.equal(nag[0].super_index[sub.points], owner)
    # label histograms come back dense (they count the raw points behind every voxel): one row
    # per node, every stored (row, column, value) triple in place, nothing else
    y = nag[1].y
    assert tuple(y.shape) == tuple(int(v) for v in ref["level_1___csr___y__shape"])
    ptr = ref["level_1___csr___y__pointers"].astype(np.int64)
    rows = np.repeat(np.arange(1192), ptr[1:] - ptr[:-1])
    cols = ref["level_1___csr___y__columns"].astype(np.int64)
    vals = torch.from_numpy(ref["level_1___csr___y__values"].astype(np.int64))
    assert torch.equal(y[rows, cols], vals) and int(y.sum()) == int(vals.sum())
    assert nag[1].edge_index.shape == (2, 9158) and nag[1].edge_attr.shape == (9158, 7)
    # partial read
    part = h5io.load_nag(H5, low=1, high=2, keys=["pos", "super_index"])
    assert part.num_levels == 2 and set(part[0].keys) == set(part[1].keys) == {"pos", "super_index"}
    # `sub` is a key like any other (data.py:887-903) and stays on the lowest loaded level
    part = h5io.load_nag(H5, low=1, high=2, keys=["pos", "sub"])
    assert "sub" in part[0] and "sub" in part[1] and part[0].sub.num_points == 41568


def test_instance_annotations_come_back_as_instance_data():
    """``Data.save`` writes an InstanceData under ``_instance_data_/<key>`` as pointers +
    value_0..2 = obj, count, y in the smallest integer types (data.py:716-718,
    csr.py:456-490); the demo file has none, so the group is added to its dataset table."""
    from superpoint_transformer_amd import h5io
    from superpoint_transformer_amd.instance import InstanceData
    flat = dict(h5io.read_h5(H5))
    n1 = 1192
    rng = np.random.default_rng(0)
    sizes = rng.integers(1, 4, n1)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    m = int(ptr[-1])
    grp = "level_1/_instance_data_/obj/"
    flat[grp + "pointers"] = ptr.astype(np.int16)
    flat[grp + "is_index_value"] = np.array([True, False, False])
    flat[grp + "value_0"] = rng.integers(0, 50, m).astype(np.uint8)
    flat[grp + "value_1"] = rng.integers(1, 3000, m).astype(np.int16)
    flat[grp + "value_2"] = rng.integers(0, 13, m).astype(np.uint8)
    nag = h5io.nag_from_datasets(flat)
    obj = nag[1].obj
    assert isinstance(obj, InstanceData) and obj.num_clusters == n1 and obj.num_overlaps == m
    assert obj.obj.dtype == obj.count.dtype == obj.y.dtype == obj.pointers.dtype == torch.int64
    assert np.array_equal(obj.count.numpy(), flat[grp + "value_1"].astype(np.int64))
    assert "obj" not in nag[0] and "obj" not in nag[2]
    assert "obj" not in h5io.nag_from_datasets(flat, keys=["pos", "super_index"])[1]


def test_write_h5_round_trips_every_dataset_of_the_demo_file(tmp_path):
    from superpoint_transformer_amd import h5io
    ref = h5io.read_h5(H5, strings=True)
    out = str(tmp_path / "copy.h5")
    h5io.write_h5(out, ref, root_attrs={"start_i_level": 0})
    got = h5io.read_h5(out, strings=True)
    assert list(got) == list(ref) or set(got) == set(ref)
    for k, a in ref.items():
        b = got[k]
        assert a.dtype == b.dtype and a.shape == b.shape, k
        assert (list(a) == list(b)) if a.dtype == object else np.array_equal(a, b), k
    assert h5io.read_root_attr(out, "start_i_level") == 0
    # half floats (fp_dtype=torch.half is the datasets' default for features) and scalars
    h = {"g/half": np.linspace(-3, 3, 77, dtype=np.float16).reshape(7, 11),
         "g/sub/scalar": np.array(5, dtype=np.int32), "empty": np.zeros((0, 3), np.float32),
         "names": ["a", "bc"], "none": []}
    h5io.write_h5(out, h)
    back = h5io.read_h5(out, strings=True)
    assert np.array_equal(back["g/half"], h["g/half"].astype(np.float32))
    assert back["g/sub/scalar"].shape == () and int(back["g/sub/scalar"]) == 5
    assert back["empty"].shape == (0, 3) and list(back["names"]) == ["a", "bc"]
    assert back["none"].dtype == np.float64 and back["none"].shape == (0,)


def test_saving_the_loaded_demo_nag_reproduces_the_reference_file(tmp_path):
    """``load_nag`` then ``save_nag`` must write what the reference's own ``NAG.save`` wrote:
    same dataset names, the same (smallest) integer types, byte colours, CSR label histograms,
    cluster groups, non-indexable names (a set in the reference: compared as sets)."""
    from superpoint_transformer_amd import h5io
    ref = h5io.read_h5(H5, strings=True)
    out = str(tmp_path / "resaved.h5")
    h5io.save_nag(h5io.load_nag(H5), out)
    got = h5io.read_h5(out, strings=True)
    assert set(got) == set(ref)
    for k, a in ref.items():
        b = got[k]
        if k.endswith("_not_indexable_"):
            assert a.dtype == b.dtype and set(a) == set(b), k
            continue
        assert a.dtype == b.dtype and a.shape == b.shape, (k, a.dtype, b.dtype)
        if k.endswith("/rgb"):
            assert np.abs(a.astype(int) - b.astype(int)).max() <= 1, k       # x/255*255 in f32
        else:
            assert np.array_equal(a, b), k
    assert h5io.read_root_attr(out, "start_i_level") == h5io.read_root_attr(H5, "start_i_level")
    again = h5io.load_nag(out)
    first = h5io.load_nag(H5)
    assert again.num_points == first.num_points
    assert torch.equal(again[1].sub.points, first[1].sub.points)


def test_instance_annotations_are_saved_like_the_reference_saves_them(tmp_path):
    from superpoint_transformer_amd import h5io
    from superpoint_transformer_amd.data import NAG, Data
    from superpoint_transformer_amd.instance import InstanceData
    ptr = torch.tensor([0, 2, 3, 6])
    obj = InstanceData(ptr, torch.tensor([7, 300, 7, 1, 2, 70000]), torch.tensor([5, 1, 9, 2, 2, 2]),
                       torch.tensor([0, 3, 0, 1, 1, -1]))
    nag = NAG([Data(pos=torch.rand(3, 3), obj=obj, rgb=torch.rand(3, 3))])
    out = str(tmp_path / "inst.h5")
    h5io.save_nag(nag, out, fp_dtype=torch.half)
    flat = h5io.read_h5(out, strings=True)
    g = "level_0/_instance_data_/obj/"
    assert flat[g + "pointers"].dtype == np.uint8 and flat[g + "value_0"].dtype == np.int32
    assert flat[g + "value_1"].dtype == np.uint8 and flat[g + "value_2"].dtype == np.int16
    assert list(flat[g + "is_index_value"]) == [1, 0, 0]
    assert flat["level_0/pos"].dtype == np.float32 and flat["level_0/rgb"].dtype == np.uint8
    assert list(flat["level_0/_not_indexable_"]) == ["obj"]
    back = h5io.load_nag(out)[0].obj
    for a, b in zip(back.values + [back.pointers], obj.values + [obj.pointers]):
        assert torch.equal(a, b)


def _reference_io_on_the_h5py_shim():
    """src/utils/io.py, src/data/csr.py, cluster.py, instance.py imported verbatim with ``h5py``
    = the shim over libhdf5 (h5py itself is not installed in this image)."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden_select as mgs
    from superpoint_transformer_amd.shims import h5py_shim
    U, csr, cluster = mgs.load_reference()               # installs an inert h5py stand-in ...
    sys.modules["h5py"] = h5py_shim                      # ... replaced by the working one
    io = importlib.reload(importlib.import_module("src.utils.io"))
    for name in ("save_tensor", "load_tensor", "save_dense_to_csr", "load_csr_to_dense"):
        setattr(U, name, getattr(io, name))
    csr = importlib.reload(csr)
    cluster = importlib.reload(cluster)
    from oracle import spt_oracle as O
    graph_mod = sys.modules.get("src.utils.graph")
    if graph_mod is None:
        sys.modules["torch_geometric.utils"].coalesce = O.coalesce
        sys.modules["torch_geometric.utils"].remove_self_loops = O.remove_self_loops
        edge = importlib.import_module("src.utils.edge")
        U.edge_wise_points = edge.edge_wise_points
        sys.modules["src.utils.scatter"].edge_wise_points = edge.edge_wise_points
        graph_mod = importlib.import_module("src.utils.graph")
    U.to_trimmed = graph_mod.to_trimmed
    inst = importlib.reload(importlib.import_module("src.data.instance"))
    return h5py_shim, io, csr, cluster, inst


@pytest.fixture
def clean_modules():
    """The golden-script hooks register stand-ins under the reference's import names
    (torch_scatter, torch_geometric, src.*): leave ``sys.modules`` as it was found."""
    import sys
    before = dict(sys.modules)
    path = list(sys.path)
    yield
    for k in list(sys.modules):
        if k not in before:
            del sys.modules[k]
    for k, v in before.items():
        sys.modules[k] = v
    sys.path[:] = path


def test_reference_readers_and_writers_run_on_the_h5py_shim(tmp_path, clean_modules):
    """The reference's own load_tensor / load_csr_to_dense / Cluster.load read the demo file
    through the shim and agree with ``load_nag``; its own save_tensor / save_dense_to_csr /
    CSRData.save / InstanceData.save write through the shim exactly what ``save_nag`` writes."""
    from superpoint_transformer_amd import h5io
    from superpoint_transformer_amd.data import NAG, Cluster, Data
    from superpoint_transformer_amd.instance import InstanceData
    h5py, io, csr, cluster, inst = _reference_io_on_the_h5py_shim()
    mine = h5io.load_nag(H5)
    with h5py.File(H5, "r") as f:
        assert f.attrs["start_i_level"] == 0 and "level_3" in f and "level_4" not in f
        assert sorted(f.keys()) == ["level_0", "level_1", "level_2", "level_3"]
        g = f["level_1"]
        assert isinstance(g, h5py.Group) and isinstance(g["pos"], h5py.Dataset)
        assert [s.decode("utf-8") for s in g["_not_indexable_"]] == ["sub", "edge_attr", "edge_index"]
        pos = io.load_tensor(g, key="pos")
        assert torch.equal(pos, mine[1].pos)
        si = io.load_tensor(f["level_0"]["super_index"], non_fp_to_long=True)
        assert torch.equal(si, mine[0].super_index)
        rows = torch.tensor([5, 0, 77])
        assert torch.equal(io.load_tensor(g["pos"], idx=rows), mine[1].pos[rows])
        y = io.load_csr_to_dense(g["_csr_"]["y"], non_fp_to_long=True)
        assert torch.equal(y, mine[1].y)
        sub, _ = cluster.Cluster.load(g["_cluster_"]["sub"], non_fp_to_long=True)
        assert torch.equal(sub.pointers, mine[1].sub.pointers)
        assert torch.equal(sub.points, mine[1].sub.points)

    # writing: the reference's code through the shim vs save_nag, same content
    gen = torch.Generator().manual_seed(0)
    n = 300
    hist = torch.randint(0, 400, (n, 14), generator=gen) * (torch.rand(n, 14, generator=gen) < 0.2)
    hist[:, 3] = hist[:, 3].clamp(min=1)         # the reference's dense_to_csr needs no empty row
    si = torch.randint(0, 70000, (n,), generator=gen)
    feat = torch.randn(n, 4, generator=gen)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long),
                     torch.randint(1, 4, (n,), generator=gen).cumsum(0)])
    m = int(ptr[-1])
    obj = (torch.randint(0, 90000, (m,), generator=gen), torch.randint(1, 200, (m,), generator=gen),
           torch.randint(-1, 13, (m,), generator=gen))
    pts = torch.randperm(m, generator=gen)
    ref_path, my_path = str(tmp_path / "ref.h5"), str(tmp_path / "mine.h5")
    with h5py.File(ref_path, "w") as f:
        f.attrs["start_i_level"] = 0
        g = f.create_group("level_0")
        io.save_tensor(feat, g, "x", fp_dtype=torch.half)
        io.save_tensor(feat[:, :3].double() * 1000, g, "pos", fp_dtype=torch.float)
        io.save_tensor(si, g, "super_index", fp_dtype=torch.half)
        io.save_dense_to_csr(hist, f.create_group(f"{g.name}/_csr_/y"), fp_dtype=torch.half)
        cluster.Cluster(ptr, pts).save(f.create_group(f"{g.name}/_cluster_/sub"), fp_dtype=torch.half)
        inst.InstanceData(ptr, *obj).save(f.create_group(f"{g.name}/_instance_data_/obj"),
                                          fp_dtype=torch.half)
        g["_not_indexable_"] = ["sub", "obj"]
    data = Data(x=feat, pos=feat[:, :3].double() * 1000, super_index=si, y=hist,
                sub=Cluster(ptr, pts), obj=InstanceData(ptr, *obj))
    h5io.save_nag(NAG([data]), my_path, fp_dtype=torch.half)
    a, b = h5io.read_h5(ref_path, strings=True), h5io.read_h5(my_path, strings=True)
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (k, a[k].dtype, b[k].dtype)
        assert (set(a[k]) == set(b[k])) if a[k].dtype == object else np.array_equal(a[k], b[k]), k
    assert h5io.read_root_attr(ref_path, "start_i_level") == 0
    back = inst.InstanceData.load(h5py.File(my_path, "r")["level_0/_instance_data_/obj"],
                                  non_fp_to_long=True)
    assert torch.equal(back.obj, obj[0]) and torch.equal(back.count, obj[1])


def test_reference_data_load_on_the_shim_agrees_with_load_nag(clean_modules):
    """``Data.load`` (src/data/data.py:735-935) cut out of the reference with ``ast`` -
    unmodified - and run level by level on the demo file through the h5py shim, against
    ``h5io.load_nag`` (integers to int64, colours to float, label histograms dense)."""
    import ast
    import time as _time
    from superpoint_transformer_amd import h5io
    h5py, io, csr, cluster, inst = _reference_io_on_the_h5py_shim()
    import sys
    U = sys.modules["src.utils"]
    tree = ast.parse(open("/root/reference/src/data/data.py").read())
    cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Data")
    fn = next(n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name == "load")
    fn.returns, fn.decorator_list = None, []
    for a in fn.args.args:
        a.annotation = None
    color = {"torch": torch}                       # color.py imports colorhash: cut the two helpers
    ctree = ast.parse(open("/root/reference/src/utils/color.py").read())
    helpers = [n for n in ctree.body if isinstance(n, ast.FunctionDef)
               and n.name in ("to_float_rgb", "to_byte_rgb")]
    exec(compile(ast.Module(body=helpers, type_ignores=[]), "color.py", "exec"), color)

    class Duck:
        _NOT_INDEXABLE = ["_csr_", "_cluster_", "_instance_data_", "edge_index", "edge_attr",
                          "_slice_dict", "_inc_dict", "_num_graphs"]

        def __init__(self, **kw):
            self.items = kw

    ns = {"torch": torch, "np": np, "h5py": h5py, "time": _time.time, "tensor_idx": U.tensor_idx,
          "is_arange": U.is_arange, "load_tensor": io.load_tensor,
          "load_tensor_dict": io.load_tensor_dict, "load_csr_to_dense": io.load_csr_to_dense,
          "Cluster": cluster.Cluster, "InstanceData": inst.InstanceData, "Data": Duck, "Batch": None,
          "to_float_rgb": color["to_float_rgb"], "to_byte_rgb": color["to_byte_rgb"]}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "data.py", "exec"), ns)
    mine = h5io.load_nag(H5)
    with h5py.File(H5, "r") as f:
        for i in range(4):
            ref = ns["load"](Duck, f[f"level_{i}"], non_fp_to_long=True, rgb_to_float=True).items
            got = mine[i]
            keys = set(got.keys)
            if i == 0:
                assert "sub" not in ref
            else:
                keys |= {"sub"} if "sub" not in keys else set()
            assert keys == set(ref), (i, keys ^ set(ref))
            for k, v in ref.items():
                if k == "sub":
                    if k in got:
                        assert torch.equal(v.pointers, got.sub.pointers)
                        assert torch.equal(v.points, got.sub.points)
                    continue
                assert v.dtype == got[k].dtype and torch.equal(v, got[k]), (i, k)
        # the reference's defaults: compressed integer types, byte colours
        raw = h5io.load_nag(H5, non_fp_to_long=False, rgb_to_float=False)
        for i in range(4):
            ref = ns["load"](Duck, f[f"level_{i}"]).items
            for k, v in ref.items():
                if k != "sub":
                    assert v.dtype == raw[i][k].dtype and torch.equal(v, raw[i][k]), (i, k, v.dtype)
        assert raw[0].rgb.dtype == torch.uint8 and raw[0].super_index.dtype == torch.int16
        # a subset of the keys, like NAG.load(keys=...)
        ref = ns["load"](Duck, f["level_1"], keys=["pos", "y"], non_fp_to_long=True).items
        sel = h5io.load_nag(H5, low=1, high=1, keys=["pos", "y"])[0]
        assert set(ref) == set(sel.keys) == {"pos", "y"}
        assert torch.equal(ref["y"], sel.y)
        # the dataset's call (datasets/base.py:1098-1104): one key set for the lowest level,
        # another above; `sub` only where asked for - and kept at the lowest loaded level
        low_keys, up_keys = ["pos", "sub", "super_index", "normal"], ["pos", "edge_index", "sub"]
        part = h5io.load_nag(H5, low=1, keys_low=low_keys, keys=up_keys)
        for i, ks in ((1, low_keys), (2, up_keys), (3, up_keys)):
            ref = ns["load"](Duck, f[f"level_{i}"], keys=ks, non_fp_to_long=True).items
            assert set(ref) == set(part[i - 1].keys), (i, set(ref) ^ set(part[i - 1].keys))
            assert torch.equal(ref["sub"].points, part[i - 1].sub.points)
            assert torch.equal(ref["pos"], part[i - 1].pos)


def test_only_the_requested_levels_and_keys_are_read_off_the_disk():
    from superpoint_transformer_amd import h5io
    seen = []
    real = h5io._read_dataset

    def spy(lib, loc, name, strings=False):
        seen.append(name)
        return real(lib, loc, name, strings)

    h5io._read_dataset = spy
    try:
        part = h5io.load_nag(H5, low=1, high=2, keys_low=["pos", "sub"], keys=["pos", "y"])
    finally:
        h5io._read_dataset = real
    assert part.num_levels == 2 and set(part[0].keys) == {"pos", "sub"} and set(part[1].keys) == {"pos", "y"}
    # level 1: pos + the two datasets of sub (+ is_index_value); level 2: pos + the four of y
    assert sorted(seen) == sorted(["pos", "pointers", "value_0", "is_index_value",
                                   "pos", "pointers", "columns", "values", "shape"])
    full = h5io.load_nag(H5)
    assert torch.equal(part[0].pos, full[1].pos) and torch.equal(part[1].y, full[2].y)


@pytest.mark.parametrize("flags", [dict(), dict(y_to_csr=False), dict(rgb_to_byte=False),
                                   dict(pos_dtype=torch.double, fp_dtype=torch.half)])
def test_reference_data_save_on_the_shim_writes_what_save_nag_writes(flags, tmp_path, clean_modules):
    """``Data.save`` (src/data/data.py:663-733) cut out of the reference with ``ast`` and run on
    a duck-typed store holding the reference's own Cluster / InstanceData objects, through the
    h5py shim - against ``h5io.save_nag`` on the mirror objects with the same tensors, for the
    switches of ``NAG.save``."""
    import ast
    from superpoint_transformer_amd import h5io
    from superpoint_transformer_amd.data import NAG, Cluster, Data
    from superpoint_transformer_amd.instance import InstanceData
    h5py, io, csr, cluster, inst = _reference_io_on_the_h5py_shim()
    tree = ast.parse(open("/root/reference/src/data/data.py").read())
    cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Data")
    fn = next(n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name == "save")
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    ns = {"torch": torch, "np": np, "h5py": h5py, "save_tensor": io.save_tensor,
          "save_dense_to_csr": io.save_dense_to_csr, "Cluster": cluster.Cluster,
          "InstanceData": inst.InstanceData, "CSRData": csr.CSRData}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "data.py", "exec"), ns)

    gen = torch.Generator().manual_seed(4)
    n, e = 200, 700
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, 5, (n,), generator=gen).cumsum(0)])
    m = int(ptr[-1])
    pts = torch.randperm(m, generator=gen)
    ov = (torch.randint(0, 500, (m,), generator=gen), torch.randint(1, 90, (m,), generator=gen),
          torch.randint(0, 13, (m,), generator=gen))
    hist = torch.randint(0, 70000, (n, 14), generator=gen) * (torch.rand(n, 14, generator=gen) < 0.3)
    hist[:, 0] = hist[:, 0].clamp(min=1)
    tensors = dict(pos=torch.randn(n, 3, generator=gen).double() * 1e4, rgb=torch.rand(n, 3, generator=gen),
                   x=torch.randn(n, 5, generator=gen), y=hist,
                   super_index=torch.randint(0, 40, (n,), generator=gen),
                   edge_index=torch.randint(0, n, (2, e), generator=gen),
                   edge_attr=torch.randn(e, 7, generator=gen),
                   pos_offset=torch.tensor([1234567.125, 7654321.5, 12.0], dtype=torch.double))

    class Duck:
        def __init__(self, **kw):
            self.store = kw

        def items(self):
            return self.store.items()

        keys = property(lambda self: list(self.store))

        def node_attrs(self):          # PyG: tensors whose first dimension is the node count
            return [k for k, v in self.store.items()
                    if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n and "edge" not in k]

    ref_path, my_path = str(tmp_path / "ref.h5"), str(tmp_path / "mine.h5")
    theirs = Duck(**tensors, sub=cluster.Cluster(ptr, pts), obj=inst.InstanceData(ptr, *ov))
    with h5py.File(ref_path, "w") as f:
        f.attrs["start_i_level"] = 0
        ns["save"](theirs, f.create_group("level_0"), **flags)
    ours = Data(**tensors, sub=Cluster(ptr, pts), obj=InstanceData(ptr, *ov))
    ours.num_nodes = n
    h5io.save_nag(NAG([ours]), my_path, **flags)
    a, b = h5io.read_h5(ref_path, strings=True), h5io.read_h5(my_path, strings=True)
    assert set(a) == set(b), set(a) ^ set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (k, a[k].dtype, b[k].dtype)
        assert (set(a[k]) == set(b[k])) if a[k].dtype == object else np.array_equal(a[k], b[k]), k
