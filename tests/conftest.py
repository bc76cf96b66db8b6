import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a CPU box: skip them there,
    and make them FAIL (not skip) if the HIP library is missing on a GPU box."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_global_rng():
    """Module constructors draw their initial weights from torch's GLOBAL generator:
    seed it per test so that every run (and every box) sees the same parameters."""
    torch.manual_seed(20240607)
    yield


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def t64(a):
    return torch.from_numpy(np.asarray(a)).double()


def tl(a):
    return torch.from_numpy(np.asarray(a)).long()


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")


def demo_nag():
    """The reference's real room NAG (notebooks/demo_nag_v3.h5 -> npz)."""
    d = load_golden("demo_nag_v3.npz")
    levels = []
    for i in range(4):
        pre = f"level_{i}__"
        lv = {k[len(pre):]: v for k, v in d.items()
              if k.startswith(pre) and not k.startswith(pre + "_")}
        sub_p = d.get(f"level_{i}___cluster___sub__pointers")
        if sub_p is not None:
            lv["sub_pointers"] = sub_p
            lv["sub_points"] = d[f"level_{i}___cluster___sub__value_0"]
        levels.append(lv)
    return levels


@pytest.fixture
def materialised_pool_route():
    """Tests that pin the ROUND-4 route of MLP -> max-pool (the top layer's output written, the
    streaming segment-max with the folded norm, the pooled LDS-DMA backward) bit for bit on
    norm-then-pool: switch the round-5 pool-fused top layer off for their duration.  The fused
    route has its own file (tests/test_fused_pool_gpu.py)."""
    from superpoint_transformer_amd import ops
    prev = ops.pool_in_forward(False)
    yield
    ops.pool_in_forward(prev)
