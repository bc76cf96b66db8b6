"""Generate the golden fixtures of tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the produced .npz
files are committed and are what travels to the GPU box.

How the reference is run here.  The reference's hot-path modules are pure
PyTorch except for imports of un-installable native deps.  We import them
VERBATIM by file path (no copy, no edit):

  * ``src``, ``src.nn``, ``src.utils`` ... are registered as bare package
    objects whose ``__path__`` points into /root/reference/src, so the
    packages' heavy ``__init__`` files (hydra, h5py, numba ...) never run;
  * ``torch_scatter`` and the handful of ``torch_geometric`` symbols the
    modules import are bound to the pure-CPU restatements of
    ``oracle/spt_oracle.py`` (the third-party binaries are absent - the
    restatement is the only available stand-in; that boundary is "parity
    unpinned");
  * ``numba``, ``pgeof``, ``git``, ``src.dependencies.FRNN`` are inert stubs.

Everything the reference computes ITSELF (SelfAttentionBlock.forward,
UnitSphereNorm, scatter_mean_weighted, scatter_pca, the eigenfeature
formulas, knn_brute_force, neighbors_dense_to_csr, TransformerBlock / Stage /
DownNFuseStage / UpNFuseStage orchestration, build_qk_scale_func) therefore
executes from the reference's own source.

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
        /opt/conda/bin/python3.9 tests/golden/make_golden.py --demo-nag
            (h5py lives only in the conda interpreter: dumps demo_nag_v3.h5)
"""
import importlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# --------------------------------------------------------------------------
def dump_demo_nag():
    """HDF5 -> npz of notebooks/demo_nag_v3.h5 (structure: SURVEY.md 8c)."""
    import h5py
    import numpy as np
    out = {}
    with h5py.File(os.path.join(REF, "notebooks", "demo_nag_v3.h5"), "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                a = np.asarray(obj[()])
                if a.dtype.kind in "biuf":      # skip string/object metadata
                    out[name.replace("/", "__")] = a
        f.visititems(visit)
    np.savez_compressed(os.path.join(HERE, "demo_nag_v3.npz"), **out)
    print("wrote demo_nag_v3.npz with", len(out), "arrays")


if "--demo-nag" in sys.argv:
    dump_demo_nag()
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from oracle import spt_oracle as O  # noqa: E402


def install_reference_import_hooks():
    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        sys.modules[name] = m
        return m

    src = pkg("src", os.path.join(REF, "src"))
    src.is_debug_enabled = lambda: False
    pkg("src.nn", os.path.join(REF, "src", "nn"))
    pkg("src.utils", os.path.join(REF, "src", "utils"))
    pkg("src.dependencies")
    frnn_pkg = pkg("src.dependencies.FRNN")
    frnn_pkg.frnn = types.SimpleNamespace(frnn_grid_points=None)
    sys.modules["src.dependencies.FRNN.frnn"] = frnn_pkg.frnn

    # inert stubs
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules["numba"] = numba
    sys.modules["pgeof"] = types.ModuleType("pgeof")
    sys.modules["git"] = types.ModuleType("git")

    # torch_scatter -> oracle restatement
    ts = types.ModuleType("torch_scatter")
    for n in ("scatter", "scatter_sum", "scatter_mean", "scatter_min",
              "scatter_max", "scatter_std"):
        setattr(ts, n, getattr(O, n))
    ts.scatter_add = O.scatter_sum
    sys.modules["torch_scatter"] = ts

    # torch_geometric subset -> oracle restatement
    tg = pkg("torch_geometric")
    tgu = pkg("torch_geometric.utils")
    tgu.softmax = O.pyg_softmax
    tgu.degree = lambda index, num_nodes=None, dtype=None: torch.bincount(
        index, minlength=num_nodes or 0).to(dtype or torch.float)
    tgu.coalesce = None
    tgn = pkg("torch_geometric.nn")
    aggr = pkg("torch_geometric.nn.aggr")

    def make_aggr(reduce):
        class _A(torch.nn.Module):
            def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
                return O.scatter(x, index, 0, None, dim_size, reduce)
        return _A

    aggr.SumAggregation = make_aggr("sum")
    aggr.MeanAggregation = make_aggr("mean")
    aggr.MaxAggregation = make_aggr("max")
    aggr.MinAggregation = make_aggr("min")

    class StdAggregation(torch.nn.Module):
        def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
            mean = O.scatter_mean(x, index, 0, None, dim_size)
            mean2 = O.scatter_mean(x * x, index, 0, None, dim_size)
            return (mean2 - mean * mean).clamp(min=1e-5).sqrt()
    aggr.StdAggregation = StdAggregation

    norm = pkg("torch_geometric.nn.norm")

    class GraphNorm(torch.nn.Module):
        def __init__(self, in_channels, eps=1e-5):
            super().__init__()
            self.in_channels, self.eps = in_channels, eps
            self.weight = torch.nn.Parameter(torch.ones(in_channels))
            self.bias = torch.nn.Parameter(torch.zeros(in_channels))
            self.mean_scale = torch.nn.Parameter(torch.ones(in_channels))

        def forward(self, x, batch=None):
            return O.graph_norm(x, batch, self.weight, self.bias,
                                self.mean_scale, self.eps)

    class _Unused(torch.nn.Module):
        pass
    norm.GraphNorm = GraphNorm
    norm.LayerNorm = type("LayerNorm", (_Unused,), {})
    norm.InstanceNorm = type("InstanceNorm", (_Unused,), {})
    inits = pkg("torch_geometric.nn.inits")
    inits.ones = lambda t: t.data.fill_(1.0) if t is not None else None
    inits.zeros = lambda t: t.data.fill_(0.0) if t is not None else None
    pool = pkg("torch_geometric.nn.pool")
    cons = pkg("torch_geometric.nn.pool.consecutive")
    cons.consecutive_cluster = None
    tg.utils, tg.nn = tgu, tgn
    tgn.aggr, tgn.norm, tgn.inits, tgn.pool = aggr, norm, inits, pool

    # src.utils: import only the hot-path files, by path, then expose names
    U = sys.modules["src.utils"]
    for sub in ("dict", "parameter", "version", "nn", "tensor", "sparse", "edge",
                "scatter", "neighbors", "geometry"):
        m = importlib.import_module(f"src.utils.{sub}")
        for k in getattr(m, "__all__", []):
            setattr(U, k, getattr(m, k))
    # src.nn: same
    N = sys.modules["src.nn"]
    for sub in ("norm", "mlp", "pool", "unpool", "attention", "fusion",
                "dropout", "transformer", "stage"):
        m = importlib.import_module(f"src.nn.{sub}")
        for k in getattr(m, "__all__", []):
            setattr(N, k, getattr(m, k))
    return U, N


def rnd(gen, *shape, scale=1.0):
    """float64 tensor holding float32-representable values."""
    return (torch.randn(*shape, generator=gen) * scale).float().double()


def synth_graph(gen, n, deg_mean):
    """Directed edges (s,t) incl. both directions and self loops, UNSORTED."""
    m = int(n * deg_mean / 2)
    a = torch.randint(0, n, (m,), generator=gen)
    b = torch.randint(0, n, (m,), generator=gen)
    keep = a != b
    a, b = a[keep], b[keep]
    loops = torch.arange(n)
    s = torch.cat([a, b, loops])
    t = torch.cat([b, a, loops])
    return torch.stack([s, t])


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


def main():
    U, N = install_reference_import_hooks()
    gen = torch.Generator().manual_seed(1234)
    torch.manual_seed(1234)

    # ---------------- a6: SelfAttentionBlock fwd + bwd (SPT-64 shape) -------
    n, dim, H, D, rpe = 97, 64, 16, 4, 32
    ei = synth_graph(gen, n, 12.0)
    E = ei.shape[1]
    blk = N.SelfAttentionBlock(dim, num_heads=H, out_dim=dim, qk_dim=D,
                               in_rpe_dim=rpe, k_rpe=True, q_rpe=True,
                               v_rpe=True).double()
    with torch.no_grad():                      # float32-representable weights
        for p in blk.parameters():
            p.copy_(p.float().double())
    x = rnd(gen, n, dim).requires_grad_()
    ea = rnd(gen, E, rpe, scale=0.5).requires_grad_()
    gw = rnd(gen, n, dim)
    out = blk(x, ei, edge_attr=ea)
    (out * gw).sum().backward()
    arrays = dict(x=x, edge_index=ei, edge_attr=ea, gw=gw, out=out,
                  g_x=x.grad, g_edge_attr=ea.grad, num_heads=H, qk_dim=D)
    for k, p in blk.named_parameters():
        arrays["p__" + k] = p
        arrays["g__" + k] = p.grad
    save("attention_spt64.npz", **arrays)

    # SPT-128 shape (value dim 8, qk_dim 4)  -- config #4
    n, dim = 61, 128
    ei = synth_graph(gen, n, 9.0)
    E = ei.shape[1]
    blk = N.SelfAttentionBlock(dim, num_heads=H, out_dim=dim, qk_dim=D,
                               in_rpe_dim=rpe, k_rpe=True, q_rpe=True,
                               v_rpe=True).double()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(p.float().double())
    x = rnd(gen, n, dim).requires_grad_()
    ea = rnd(gen, E, rpe, scale=0.5).requires_grad_()
    gw = rnd(gen, n, dim)
    out = blk(x, ei, edge_attr=ea)
    (out * gw).sum().backward()
    arrays = dict(x=x, edge_index=ei, edge_attr=ea, gw=gw, out=out,
                  g_x=x.grad, g_edge_attr=ea.grad, num_heads=H, qk_dim=D)
    for k, p in blk.named_parameters():
        arrays["p__" + k] = p
        arrays["g__" + k] = p.grad
    save("attention_spt128.npz", **arrays)

    # ---------------- a4: UnitSphereNorm ------------------------------------
    npts, nseg = 700, 40
    pos = rnd(gen, npts, 3, scale=3.0)
    idx = torch.randint(0, nseg - 2, (npts,), generator=gen)   # 2 empty segs
    w = torch.randint(0, 50, (npts,), generator=gen)
    usn = N.UnitSphereNorm()
    o1, d1 = usn(pos, idx, w=None, num_super=nseg)
    o2, d2 = usn(pos, idx, w=w, num_super=nseg)
    o3, d3 = usn(pos, None, w=w)
    o4, d4 = usn(pos, None, w=None)
    save("unit_sphere_norm.npz", pos=pos, idx=idx, w=w, num_super=nseg,
         out_unw=o1, diam_unw=d1, out_w=o2, diam_w=d2,
         out_none_w=o3, diam_none_w=d3, out_none=o4, diam_none=d4)

    # ---------------- a12/a13: scatter_pca + eigenfeatures -------------------
    npts, k = 400, 12
    # planar / linear / volumetric patches so that all feature regimes occur
    base = rnd(gen, npts, 3)
    base[:150, 2] *= 0.01
    base[150:250, 1:] *= 0.01
    xyz = base.float()
    nn_idx, nn_d = U.knn_brute_force(xyz, xyz, k + 1, r_max=0.9)
    nn_idx = nn_idx[:, 1:]                         # drop self like knn_1
    save("knn_brute_force.npz", xyz=xyz, k=k, r_max=0.9,
         neighbors=nn_idx, distances=nn_d[:, 1:])
    nn_full = torch.cat((torch.arange(npts).view(-1, 1), nn_idx), dim=1)
    G = importlib.import_module("src.utils.geometry")
    # scatter_pca allocates `cov` with the default dtype (scatter.py:73):
    # run the reference in float64 by switching the default dtype
    torch.set_default_dtype(torch.float64)
    f = G.geometric_features_torch(xyz.double(), nn_full, k_min=1, k_step=-1,
                                   k_min_search=25, chunk_size=None)
    f["verticality"] = f["verticality"] * 2          # geometry.py:121
    nrm = f["normal"].clone()
    nrm[nrm[:, 2] < 0] *= -1                         # geometry.py:124
    feats = torch.cat([f["linearity"], f["planarity"], f["scattering"],
                       f["verticality"], nrm, f["length"], f["surface"],
                       f["volume"], f["curvature"]], dim=1)
    ptr, val, sizes = U.neighbors_dense_to_csr(nn_full)
    gidx = torch.repeat_interleave(torch.arange(npts), ptr[1:] - ptr[:-1])
    ev, evec = U.scatter_pca(xyz.double()[val], gidx)
    torch.set_default_dtype(torch.float32)
    save("geometric_features.npz", xyz=xyz, nn=nn_idx, k_min=1, feats=feats,
         eigenval=ev, nn_ptr=ptr, nn_val=val, sizes=sizes)

    # ---------------- a7: TransformerBlock / Stage orchestration -------------
    from functools import partial
    VH = U.VersionHolder
    vh = VH("3.0.0", commit_hash="golden")
    GN = sys.modules["torch_geometric.nn.norm"].GraphNorm
    n1, n2, dim = 83, 31, 64
    ei1 = synth_graph(gen, n1, 10.0)
    E1 = ei1.shape[1]
    stage = N.DownNFuseStage(
        dim, num_blocks=2, num_heads=16, in_mlp=[4 + 3 + 128, dim, dim],
        mlp_norm=GN, qk_dim=4, k_rpe=True, q_rpe=True, v_rpe=True,
        in_rpe_dim=32, norm=GN, no_ffn=True, pool="max", fusion="cat",
        use_pos=True, use_diameter_parent=False, version_holder=vh).double()
    with torch.no_grad():
        for p in stage.parameters():
            p.copy_((p + 0.05 * torch.randn(p.shape, generator=gen).double()).float().double())
    n0 = 900
    x_child = rnd(gen, n0, 128).requires_grad_()
    pool_index = torch.randint(0, n1, (n0,), generator=gen)
    x_parent = rnd(gen, n1, 4)
    pos1 = rnd(gen, n1, 3, scale=2.0)
    super1 = torch.randint(0, n2, (n1,), generator=gen)
    node_size = torch.randint(1, 200, (n1,), generator=gen)
    norm_index = (torch.arange(n1) >= n1 // 2).long()      # 2 clouds in batch
    ea1 = rnd(gen, E1, 32, scale=0.5)
    xo, diam = stage(x_parent, x_child, norm_index, pool_index, pos=pos1,
                     node_size=node_size, super_index=super1, edge_index=ei1,
                     edge_attr=ea1, num_super=n1)
    gw = rnd(gen, n1, dim)
    (xo * gw).sum().backward()
    arrays = dict(x_parent=x_parent, x_child=x_child, norm_index=norm_index,
                  pool_index=pool_index, pos=pos1, node_size=node_size,
                  super_index=super1, edge_index=ei1, edge_attr=ea1,
                  num_super=n1, out=xo, diameter=diam, gw=gw,
                  g_x_child=x_child.grad)
    for k_, p in stage.named_parameters():
        arrays["p__" + k_] = p
        arrays["g__" + k_] = p.grad
    save("down_stage.npz", **arrays)

    up = N.UpNFuseStage(
        dim, num_blocks=1, num_heads=16, in_mlp=[dim + 3 + dim, dim, dim],
        mlp_norm=GN, qk_dim=4, k_rpe=True, q_rpe=True, v_rpe=True,
        in_rpe_dim=32, norm=GN, no_ffn=False, ffn_ratio=1,
        unpool="index", fusion="cat", use_pos=True,
        version_holder=vh).double()
    with torch.no_grad():
        for p in up.parameters():
            p.copy_((p + 0.05 * torch.randn(p.shape, generator=gen).double()).float().double())
    x_par = rnd(gen, n2, dim).requires_grad_()
    x_ch = rnd(gen, n1, dim).requires_grad_()
    xo, _ = up(x_ch, x_par, norm_index, super1, pos=pos1, node_size=node_size,
               super_index=super1, edge_index=ei1, edge_attr=ea1)
    (xo * gw).sum().backward()
    arrays = dict(x_child=x_ch, x_parent=x_par, norm_index=norm_index,
                  unpool_index=super1, pos=pos1, node_size=node_size,
                  super_index=super1, edge_index=ei1, edge_attr=ea1, out=xo,
                  gw=gw, g_x_child=x_ch.grad, g_x_parent=x_par.grad)
    for k_, p in up.named_parameters():
        arrays["p__" + k_] = p
        arrays["g__" + k_] = p.grad
    save("up_stage.npz", **arrays)


if __name__ == "__main__":
    main()
