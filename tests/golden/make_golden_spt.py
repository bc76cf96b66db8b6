"""Golden fixtures of the WHOLE backbone, produced by the REFERENCE'S OWN
``SPT.forward`` (src/models/components/spt.py:760-944) in float64.

``spt.py`` is imported verbatim by path on top of the hooks of make_golden.py
(reference src/nn/*.py + src/utils/*.py verbatim, torch_scatter / PyG symbols
bound to the oracle restatement).  ``src.data`` cannot be imported (it
subclasses torch_geometric's Data and pulls h5py at import), so ``Data`` /
``NAG`` are duck-typed attribute stores whose ``norm_index`` and
``add_keys_to`` are the reference's FunctionDefs, cut out of
src/data/data.py and src/data/nag.py with ``ast`` - unmodified.

Two configurations, each derived from the reference's config tree:
  * spt64  - configs/experiment/semantic/s3dis.yaml -> model/semantic/spt-2.yaml
             (down_dim [64,64], 16 heads, qk_dim 4, no_ffn, 3 blocks down / 1 up)
  * spt128 - configs/experiment/semantic/kitti360.yaml (same spt-2 tree with
             _down_dim/_up_dim 128, no_ffn False, down_ffn_ratio 1)
both on a 2-cloud batch: 3 levels, 8 point features, 18-D edge features.
  * nano2  - configs/experiment/semantic/s3dis_nano.yaml -> model/semantic/nano-2.yaml
             (nano=True: no level-0 stage, the NAG starts at level 1 with 8 handcrafted segment
             features; dims 16, qk_dim 2, node / edge MLPs to 16) on levels 1-2 of the same
             kind of batch.

Outputs: every input tensor, the state dict (names = the reference's), the
stage-wise outputs, and d(loss)/d(parameter) for loss = sum_i <out_i, gw_i>.

Usage (build container only): python tests/golden/make_golden_spt.py
"""
import ast
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF = mg.REF


def cut(path, cls, name):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.returns = None
    for a in fn.args.args:
        a.annotation = None
    return ast.Module(body=[fn], type_ignores=[])


def load_reference_spt():
    U, N = mg.install_reference_import_hooks()
    src = sys.modules["src"]
    src.__version__ = "3.0.0"                                  # src/__init__.py:13
    oc = types.ModuleType("omegaconf")                         # isinstance checks only
    oc.ListConfig = type("ListConfig", (), {})
    sys.modules.setdefault("omegaconf", oc)
    lst = importlib.import_module("src.utils.list")
    for k in lst.__all__:
        setattr(U, k, getattr(lst, k))

    env = {"torch": torch, "List": list,
           "fill_list_with_string_indexing": U.fill_list_with_string_indexing}

    class Data:
        """Attribute store; a missing attribute reads as None (the reference
        accesses optional keys through ``getattr(data, key, None)``)."""

        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __getattr__(self, k):          # only called when the attribute is missing
            if k.startswith("__"):
                raise AttributeError(k)
            return None

        @property
        def num_nodes(self):
            return self.pos.shape[0]

        @property
        def device(self):
            return self.pos.device

    for name in ("norm_index", "add_keys_to"):                 # data.py:103-130, 1097-1141
        exec(compile(cut("src/data/data.py", "Data", name), "data.py", "exec"), env)
        setattr(Data, name, env[name])

    class NAG:
        def __init__(self, levels, start_i_level=0):
            self._list = levels
            self.start_i_level = start_i_level

        def __getitem__(self, i):                  # absolute level index (nag.py:133-160)
            return self._list[i - self.start_i_level]

        num_levels = property(lambda self: len(self._list))
        absolute_num_levels = property(lambda self: len(self._list) + self.start_i_level)
        end_i_level = property(lambda self: len(self._list) + self.start_i_level - 1)

    exec(compile(cut("src/data/nag.py", "NAG", "add_keys_to"), "nag.py", "exec"), env)
    NAG.add_keys_to = env["add_keys_to"]                       # nag.py:834-868

    data_pkg = types.ModuleType("src.data")
    data_pkg.Data, data_pkg.NAG = Data, NAG
    sys.modules["src.data"] = data_pkg
    for name in ("src.models", "src.models.components"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = m
    spt = importlib.import_module("src.models.components.spt")
    return spt.SPT, Data, NAG


def config(which, point_in=8, edge_in=18):
    """Widths from the config tree (model/semantic/default.yaml, _point/_down/_up/_attention,
    spt-2.yaml; segment_hf = [] in every shipped datamodule => node_mlp None,
    _node_injection_dim = 3 + 1)."""
    GN = sys.modules["torch_geometric.nn.norm"].GraphNorm
    d = 64 if which == "spt64" else 128
    inj = 3 + 1
    cfg = dict(
        point_hf=[], segment_hf=[], post_cnn_point_hf=[],
        point_mlp=[point_in + inj, 32, 64, 128], point_drop=None, nano=False,
        down_dim=[d, d], down_pool_dim=[128, d],
        down_in_mlp=[[inj + 128, d, d], [inj + d, d, d]], down_out_mlp=None,
        down_mlp_drop=None, down_num_heads=16, down_num_blocks=3, down_ffn_ratio=1,
        down_residual_drop=None, down_attn_drop=None, down_drop_path=None,
        up_dim=[d], up_in_mlp=[[inj + d + d, d, d]], up_out_mlp=None, up_mlp_drop=None,
        up_num_heads=16, up_num_blocks=1, up_ffn_ratio=1, up_residual_drop=None,
        up_attn_drop=None, up_drop_path=None,
        node_mlp=None, h_edge_mlp=[edge_in, 32, 32], v_edge_mlp=None,
        mlp_activation=torch.nn.LeakyReLU(), mlp_norm=GN, qk_dim=4, qkv_bias=True,
        qk_scale=None, in_rpe_dim=32, activation=torch.nn.LeakyReLU(), norm=GN,
        pre_norm=True, no_sa=False, no_ffn=(which == "spt64"), k_rpe=True, q_rpe=True,
        v_rpe=True, k_delta_rpe=False, q_delta_rpe=False, qk_share_rpe=False,
        q_on_minus_rpe=False, share_hf_mlps=False, stages_share_rpe=False,
        blocks_share_rpe=False, heads_share_rpe=False, use_pos=True, use_node_hf=True,
        use_diameter=False, use_diameter_parent=True, pool="max", unpool="index",
        fusion="cat", norm_mode="graph", output_stage_wise=True)
    return cfg


def config_nano(seg_in=8, edge_in=18):
    """model/semantic/nano-2.yaml on datamodule/semantic/s3dis_nano.yaml: 8 segment features
    (linearity, planarity, scattering, verticality, elevation, rgb), _node_mlp_out =
    _h_edge_mlp_out = 16, _node_injection_dim = 3 + 1 + 16, qk_dim 2."""
    cfg = config("spt64")
    d, inj = 16, 3 + 1 + 16
    cfg.update(
        nano=True, point_mlp=None, down_dim=[d, d], down_pool_dim=[128, d],
        down_in_mlp=[[inj, d, d], [inj + d, d, d]], up_dim=[d],
        up_in_mlp=[[inj + d + d, d, d]], node_mlp=[seg_in, d, d], h_edge_mlp=[edge_in, d, d],
        qk_dim=2, in_rpe_dim=d)
    return cfg


def synth_levels(gen, n0=2600, n1=140, n2=45, clouds=2):
    """Small 3-level hierarchy, 2 clouds, unsorted super_index, edges in both
    directions + self loops (the layout the per-batch transforms hand to the model)."""
    rnd = mg.rnd
    b2 = (torch.arange(n2) >= n2 // 2).long() if clouds == 2 else torch.zeros(n2, dtype=torch.long)
    sup1 = torch.randint(0, n2, (n1,), generator=gen)
    sup1[:n2] = torch.arange(n2)                   # no empty level-2 node
    b1 = b2[sup1]
    sup0 = torch.randint(0, n1, (n0,), generator=gen)
    sup0[:n1] = torch.arange(n1)
    b0 = b1[sup0]
    # GraphNorm's `batch` need not be sorted; keep the hierarchy's natural (unsorted) order
    size1 = torch.bincount(sup0, minlength=n1)
    size2 = torch.zeros(n2, dtype=torch.long).index_add_(0, sup1, size1)
    lv = []
    lv.append(dict(pos=rnd(gen, n0, 3, scale=2.0), x=rnd(gen, n0, 8), super_index=sup0,
                   batch=b0, node_size=None))
    for n, sup, b, size, deg in ((n1, sup1, b1, size1, 9.0), (n2, None, b2, size2, 7.0)):
        ei = mg.synth_graph(gen, n, deg)
        # edges must stay inside a cloud for a realistic batch; drop the others
        keep = b[ei[0]] == b[ei[1]]
        ei = ei[:, keep]
        lv.append(dict(pos=rnd(gen, n, 3, scale=2.0), super_index=sup, batch=b,
                       node_size=size, edge_index=ei,
                       edge_attr=rnd(gen, ei.shape[1], 18, scale=0.5)))
    return lv


def run(which, SPT, Data, NAG, gen):
    torch.manual_seed(77 if which == "spt64" else 78)
    model = SPT(**config(which)).double()
    with torch.no_grad():                          # f32-representable, away from the all-ones init
        for p in model.parameters():
            p.copy_((p + 0.05 * torch.randn(p.shape, generator=gen).double()).float().double())
    levels = synth_levels(gen)
    nag = NAG([Data(**{k: v for k, v in lv.items() if v is not None}) for lv in levels])
    outs = model(nag)
    gws = [mg.rnd(gen, *o.shape) for o in outs]
    sum((o * g).sum() for o, g in zip(outs, gws)).backward()
    arrays = {}
    for i, lv in enumerate(levels):
        for k, v in lv.items():
            if v is not None:
                arrays[f"l{i}__{k}"] = v
    for i, (o, g) in enumerate(zip(outs, gws)):
        arrays[f"out{i}"] = o
        arrays[f"gw{i}"] = g
    for k, p in model.named_parameters():
        arrays["p__" + k] = p.detach().float()     # f32-representable by construction: exact
        assert p.grad is not None, k
        arrays["g__" + k] = p.grad.float()         # f64 result rounded once (fixture size)
    arrays["num_clouds"] = 2
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)
           for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, f"spt_forward_{which}.npz"), **out)
    print(f"wrote spt_forward_{which}.npz: {len(out)} arrays, "
          f"{sum(p.numel() for p in model.parameters())} parameters")


def run_nano(SPT, Data, NAG, gen):
    torch.manual_seed(79)
    model = SPT(**config_nano()).double()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_((p + 0.05 * torch.randn(p.shape, generator=gen).double()).float().double())
    levels = synth_levels(gen, n0=2000, n1=900, n2=130)[1:]    # levels 1 and 2 of a hierarchy
    for lv in levels:
        lv["x"] = mg.rnd(gen, lv["pos"].shape[0], 8)
    nag = NAG([Data(**{k: v for k, v in lv.items() if v is not None}) for lv in levels],
              start_i_level=1)
    outs = model(nag)
    gws = [mg.rnd(gen, *o.shape) for o in outs]
    sum((o * g).sum() for o, g in zip(outs, gws)).backward()
    arrays = {}
    for i, lv in enumerate(levels):
        for k, v in lv.items():
            if v is not None:
                arrays[f"l{i + 1}__{k}"] = v
    for i, (o, g) in enumerate(zip(outs, gws)):
        arrays[f"out{i}"] = o
        arrays[f"gw{i}"] = g
    for k, p in model.named_parameters():
        arrays["p__" + k] = p.detach().float()
        assert p.grad is not None, k
        arrays["g__" + k] = p.grad.float()
    arrays["num_clouds"] = 2
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)
           for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, "spt_forward_nano2.npz"), **out)
    print(f"wrote spt_forward_nano2.npz: {len(out)} arrays, "
          f"{sum(p.numel() for p in model.parameters())} parameters")


def main():
    SPT, Data, NAG = load_reference_spt()
    if "--nano-only" not in sys.argv:
        gen = torch.Generator().manual_seed(4242)
        for which in ("spt64", "spt128"):
            run(which, SPT, Data, NAG, gen)
    run_nano(SPT, Data, NAG, torch.Generator().manual_seed(4343))


if __name__ == "__main__":
    main()
