"""Child -> parent pooling (src/nn/pool.py:24-82) on the segment-CSR kernels."""
from torch import nn

from .. import ops

__all__ = ["pool_factory", "SumPool", "MeanPool", "MaxPool", "MinPool"]


class _AggregationPool(nn.Module):
    reduce = None

    def forward(self, x_child, x_parent, index, edge_attr=None, num_pool=None):
        return ops.segment_reduce(x_child, index, num_pool, self.reduce)


class SumPool(_AggregationPool):
    reduce = "sum"


class MeanPool(_AggregationPool):
    reduce = "mean"


class MaxPool(_AggregationPool):
    reduce = "max"


class MinPool(_AggregationPool):
    reduce = "min"


def pool_factory(pool, *args, **kwargs):
    if isinstance(pool, _AggregationPool):
        return pool
    table = {"max": MaxPool, "min": MinPool, "mean": MeanPool, "sum": SumPool}
    if isinstance(pool, str):
        if pool not in table:
            raise NotImplementedError(f"pool='{pool}' is not built on the HIP path")
        return table[pool]()
    return pool(*args, **kwargs)
