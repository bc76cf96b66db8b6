"""Import-name shims: the module names the reference imports for this path
(``torch_scatter``, a subset of ``torch_geometric``, ``src.dependencies.FRNN.frnn``,
``pgeof``) bound to the HIP kernels, so that the reference's own
``src/nn/*.py`` / ``src/utils/*.py`` import and run UNCHANGED on an MI355X.

    from superpoint_transformer_amd import shims
    shims.install()          # before `import src...`

Nothing is registered implicitly, and an already-imported real package is
never replaced unless ``force=True``."""
import sys
import types

from . import frnn_shim, pgeof_shim, pyg_shim, scatter_shim


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def install(force=False, with_h5py=False):
    """Register the shims in ``sys.modules`` under the reference's import names.
    ``with_h5py``: also ``h5py`` (the subset the reference's I/O uses, on libhdf5 through
    ctypes) - opt-in, for boxes without h5py."""
    table = {
        "torch_scatter": scatter_shim,
        "pgeof": pgeof_shim,
        "src.dependencies.FRNN.frnn": frnn_shim,
    }
    tg = _module("torch_geometric", __path__=[])
    tgu = _module("torch_geometric.utils", softmax=pyg_shim.softmax, degree=pyg_shim.degree,
                  scatter=pyg_shim.scatter, coalesce=pyg_shim.coalesce,
                  remove_self_loops=pyg_shim.remove_self_loops,
                  add_self_loops=pyg_shim.add_self_loops, to_undirected=pyg_shim.to_undirected)
    tgn = _module("torch_geometric.nn", __path__=[])
    aggr = _module("torch_geometric.nn.aggr", SumAggregation=pyg_shim.SumAggregation,
                   MeanAggregation=pyg_shim.MeanAggregation,
                   MaxAggregation=pyg_shim.MaxAggregation,
                   MinAggregation=pyg_shim.MinAggregation,
                   StdAggregation=pyg_shim.StdAggregation)
    norm = _module("torch_geometric.nn.norm", GraphNorm=pyg_shim.GraphNorm,
                   LayerNorm=pyg_shim.LayerNorm, InstanceNorm=pyg_shim.InstanceNorm)
    inits = _module("torch_geometric.nn.inits", ones=pyg_shim.ones, zeros=pyg_shim.zeros)
    pool = _module("torch_geometric.nn.pool", __path__=[])
    cons = _module("torch_geometric.nn.pool.consecutive",
                   consecutive_cluster=pyg_shim.consecutive_cluster)
    tg.utils, tg.nn = tgu, tgn
    tgn.aggr, tgn.norm, tgn.inits, tgn.pool = aggr, norm, inits, pool
    pool.consecutive = cons
    table.update({
        "torch_geometric": tg, "torch_geometric.utils": tgu, "torch_geometric.nn": tgn,
        "torch_geometric.nn.aggr": aggr, "torch_geometric.nn.norm": norm,
        "torch_geometric.nn.inits": inits, "torch_geometric.nn.pool": pool,
        "torch_geometric.nn.pool.consecutive": cons,
    })
    if with_h5py:
        from . import h5py_shim
        table["h5py"] = h5py_shim
    for name, mod in table.items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
    return sorted(table)
