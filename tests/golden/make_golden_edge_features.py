"""Golden fixture for the on-the-fly horizontal edge features, produced by
EXECUTING THE REFERENCE'S OWN FUNCTION SOURCE.

``src/transforms/graph.py`` cannot be imported here (it pulls torch_geometric,
torch_scatter, h5py ... at module level), so the FunctionDef of
``_on_the_fly_horizontal_edge_features`` (graph.py:1135-1277) is cut out of the
file with ``ast`` - unmodified - and executed against a duck-typed Data object;
``sanitize_keys`` / ``ON_THE_FLY_HORIZONTAL_FEATURES`` come from the reference's
own ``src/utils/keys.py`` (loaded by path).  Inputs: level 1 of the reference's
demo room (real trimmed edges, real 7-D edge attributes, real node attributes).

Usage (build container only): python tests/golden/make_golden_edge_features.py
"""
import ast
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class DuckData(dict):
    """Just enough of src.data.Data for the function body."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v

    def raise_if_edge_keys(self):
        pass

    @property
    def edge_keys(self):
        return []


def reference_function(name="_on_the_fly_horizontal_edge_features"):
    src = open(os.path.join(REF, "src", "transforms", "graph.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src))
              if isinstance(n, ast.FunctionDef) and n.name == name)
    spec = importlib.util.spec_from_file_location("ref_keys", os.path.join(REF, "src", "utils", "keys.py"))
    keys = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(keys)
    ns = {"torch": torch, "sanitize_keys": keys.sanitize_keys,
          "ON_THE_FLY_HORIZONTAL_FEATURES": keys.ON_THE_FLY_HORIZONTAL_FEATURES,
          "ON_THE_FLY_VERTICAL_FEATURES": keys.ON_THE_FLY_VERTICAL_FEATURES,
          "is_trimmed": lambda ei: bool((ei[0] < ei[1]).all())}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "graph.py", "exec"), ns)
    return ns[name]


def main():
    d = np.load(os.path.join(HERE, "demo_nag_v3.npz"))
    lv = {k[len("level_1__"):]: d[k] for k in d.files if k.startswith("level_1__") and "___" not in k}
    f = reference_function()
    data = DuckData(
        edge_index=torch.from_numpy(lv["edge_index"].astype(np.int64)),
        edge_attr=torch.from_numpy(lv["edge_attr"]),                 # float16 on disk
        pos=torch.from_numpy(lv["pos"]).float(),
        normal=torch.from_numpy(lv["normal"]).float(),
        log_length=torch.from_numpy(lv["log_length"]).float(),
        log_surface=torch.from_numpy(lv["log_surface"]).float(),
        log_volume=torch.from_numpy(lv["log_volume"]).float(),
        log_size=torch.from_numpy(lv["log_size"]).float())
    inputs = {k: v.clone() for k, v in data.items()}
    out = f(data)
    np.savez_compressed(
        os.path.join(HERE, "horizontal_edge_features.npz"),
        **{"in__" + k: v.numpy() for k, v in inputs.items()},
        edge_index=out.edge_index.numpy(), edge_attr=out.edge_attr.numpy())
    print("wrote horizontal_edge_features.npz", tuple(out.edge_index.shape), tuple(out.edge_attr.shape))

    # vertical graph child (level 1) -> parent (level 2): graph.py:1335-1416, all default keys
    fv = reference_function("_on_the_fly_vertical_edge_features")

    def level(i):
        lv = {k[len(f"level_{i}__"):]: d[k] for k in d.files
              if k.startswith(f"level_{i}__") and "___" not in k}
        out = DuckData(pos=torch.from_numpy(lv["pos"]).float(),
                       normal=torch.from_numpy(lv["normal"]).float(),
                       log_length=torch.from_numpy(lv["log_length"]).float(),
                       log_surface=torch.from_numpy(lv["log_surface"]).float(),
                       log_volume=torch.from_numpy(lv["log_volume"]).float(),
                       log_size=torch.from_numpy(lv["log_size"]).float())
        if "super_index" in lv:
            out["super_index"] = torch.from_numpy(lv["super_index"].astype(np.int64))
        return out
    child, parent = level(1), level(2)
    # a child sitting exactly on its parent's centroid: 0/0 direction -> 0 (graph.py:1380-1381)
    child["pos"][3] = parent["pos"][child["super_index"][3]]
    cin = {k: v.clone() for k, v in child.items()}
    pin = {k: v.clone() for k, v in parent.items()}
    res = fv(child, parent)
    np.savez_compressed(
        os.path.join(HERE, "vertical_edge_features.npz"),
        **{"child__" + k: v.numpy() for k, v in cin.items()},
        **{"parent__" + k: v.numpy() for k, v in pin.items()},
        v_edge_attr=res.v_edge_attr.numpy())
    print("wrote vertical_edge_features.npz", tuple(res.v_edge_attr.shape))


if __name__ == "__main__":
    main()
