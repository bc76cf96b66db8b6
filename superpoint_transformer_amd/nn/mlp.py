"""MLP / FFN / Classifier (src/nn/mlp.py).  Tall ``Linear -> GraphNorm -> LeakyReLU`` stacks
run as one fused HIP kernel per layer and direction (``ops.fused_mlp``); other Linears go through
``ops.linear`` (the skinny-GEMM kernels where the shape is built, the library below 4 096 rows
or for unbuilt shapes); a lone ``GraphNorm -> LeakyReLU`` pair is one fused pass."""
from torch import nn

from .. import ops
from .norm import INDEX_BASED_NORMS, GraphNorm

__all__ = ["MLP", "FFN", "Classifier"]


def _mlp(dims, activation, last_activation, norm, last_norm, drop):
    assert len(dims) >= 2
    bias = norm is None
    mods = []
    for i in range(1, len(dims)):
        mods.append(nn.Linear(dims[i - 1], dims[i], bias=bias))
        if norm is not None and (last_norm or i < len(dims) - 1):
            mods.append(norm(dims[i]))
        if activation is not None and (last_activation or i < len(dims) - 1):
            mods.append(activation)
    if drop is not None and drop > 0:
        mods.append(nn.Dropout(drop, inplace=True))
    return nn.ModuleList(mods)


class MLP(nn.Module):
    """Linear -> norm -> activation stacks on [N, D] features (mlp.py:60-94).
    Parameter paths are ``mlp.<i>.*`` like the reference."""

    def __init__(self, dims, activation=nn.LeakyReLU(), last_activation=True,
                 norm=GraphNorm, last_norm=True, drop=None):
        super().__init__()
        self.mlp = _mlp(dims, activation, last_activation, norm, last_norm, drop)
        self.out_dim = dims[-1]

    def _fused_layers(self):
        """[(W, gn_weight, gn_bias, gn_mean_scale, eps, slope)] when the stack is
        exactly [bias-free Linear -> GraphNorm (-> LeakyReLU)] x L, else None."""
        mods = list(self.mlp)
        out, i = [], 0
        while i < len(mods):
            lin = mods[i]
            gn = mods[i + 1] if i + 1 < len(mods) else None
            if not (isinstance(lin, nn.Linear) and lin.bias is None and isinstance(gn, GraphNorm)
                    and not gn.generic):
                return None
            act = mods[i + 2] if i + 2 < len(mods) else None
            if isinstance(act, nn.LeakyReLU):
                out.append((lin.weight, gn.weight, gn.bias, gn.mean_scale, gn.eps, act.negative_slope))
                i += 3
            else:
                out.append((lin.weight, gn.weight, gn.bias, gn.mean_scale, gn.eps, 1.0))
                i += 2
        return out or None

    FUSE_MIN_ROWS = 4096      # below that a layer is a few microseconds either way

    def forward_max_pooled(self, x, index, num_pool, batch=None, batch_size=None,
                           seg_graph=None):
        """``max-pool(self(x), index)`` with the last norm + activation folded into the pool's
        read of the raw activations (ops.fused_mlp_maxpool), or None when that route does not
        apply (the caller then runs ``self(x)`` and the pool separately)."""
        if not (x.is_cuda and x.shape[0] >= self.FUSE_MIN_ROWS and
                (batch is None or batch_size is not None)):
            return None
        layers = self._fused_layers()
        if layers is None:
            return None
        return ops.fused_mlp_maxpool(x, batch, batch_size, layers, index, num_pool, seg_graph)

    def forward(self, x, batch=None, batch_size=None):
        if x.is_cuda and x.shape[0] >= self.FUSE_MIN_ROWS and (batch is None or batch_size is not None):
            layers = self._fused_layers()
            if layers is not None:
                y = ops.fused_mlp(x, batch, batch_size, layers)
                if y is not None:
                    return y
        mods = list(self.mlp)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, GraphNorm):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, nn.LeakyReLU):
                    x = m(x, batch=batch, batch_size=batch_size,
                          act_slope=nxt.negative_slope)
                    i += 2
                    continue
                x = m(x, batch=batch, batch_size=batch_size)
            elif isinstance(m, INDEX_BASED_NORMS):
                x = m(x, batch=batch)
            elif isinstance(m, nn.Linear):
                x = ops.linear(x, m.weight, m.bias)
            else:
                x = m(x)
            i += 1
        return x


class FFN(MLP):
    """Two Linear layers, no norm, no last activation (mlp.py:97-125)."""

    def __init__(self, dim, hidden_dim=None, out_dim=None, activation=nn.LeakyReLU(),
                 drop=None):
        hidden_dim = hidden_dim or dim
        out_dim = out_dim or dim
        super().__init__([dim, hidden_dim, out_dim], activation=activation,
                         last_activation=False, norm=None, last_norm=False, drop=drop)

    def forward_prenorm_residual(self, x, norm, norm_index, num_graphs):
        """``x + self(norm(x))`` (the FFN branch of a pre-norm TransformerBlock,
        src/nn/transformer.py:246-249) with the GraphNorm applied inside the first Linear's read
        of x and the residual added in the second Linear's epilogue; None when the fused route
        does not apply."""
        mods = list(self.mlp)
        if not (len(mods) == 3 and isinstance(mods[0], nn.Linear) and isinstance(mods[2], nn.Linear)
                and not getattr(norm, "generic", False)
                and ops.norm_linear_ok(x, norm_index, num_graphs, mods[0].weight)
                and ops.linear_residual_ok(x, mods[2].weight)):
            return None
        h, x_res = ops.norm_linear(x, norm_index, num_graphs, norm.weight, norm.bias,
                                   norm.mean_scale, norm.eps, mods[0].weight, mods[0].bias)
        return ops.linear_residual(mods[1](h), mods[2].weight, mods[2].bias, x_res)


class Classifier(nn.Module):
    def __init__(self, in_dim, num_classes, bias=True):
        super().__init__()
        self.classifier = nn.Linear(in_dim, num_classes, bias=bias)

    def forward(self, x):
        # the head's Linear on the skinny kernels (narrow variant: 13 classes; backward = one pass)
        return ops.linear(x, self.classifier.weight, self.classifier.bias)
