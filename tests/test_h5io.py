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
    assert part.num_levels == 2 and not hasattr(part[0], "sub") and hasattr(part[1], "sub")
