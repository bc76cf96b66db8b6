"""Golden fixture for the attentive pools (src/nn/pool.py:84-360), produced by the REFERENCE'S
OWN modules imported verbatim on the hooks of make_golden.py (torch_scatter / PyG softmax =
oracle restatements), in float64 with float32-representable inputs and parameters: forward
output, input gradients and parameter gradients of

  * ``AttentivePool`` with in_proj / out_proj, k_rpe + q_rpe per head, default ``d.g`` scale;
  * ``AttentivePool`` with RPE encoders shared by the heads and ``qk_scale='d+g'``;
  * ``AttentivePoolWithLearntQueries`` with a scalar ``qk_scale``.

Usage (build container only): python tests/golden/make_golden_pool.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def run(tag, pool, gen, nc, n_parent, c_child, c_parent, f, out, empty_parents=0):
    pool = pool.double()
    with torch.no_grad():
        for p in pool.parameters():
            p.copy_((p + 0.05 * torch.randn(p.shape, generator=gen).double()).float().double())
    live = n_parent - empty_parents                 # the last parents have no child
    index = torch.randint(0, live, (nc,), generator=gen)
    index[:live] = torch.randperm(live, generator=gen)
    xc = mg.rnd(gen, nc, c_child).requires_grad_()
    xp = mg.rnd(gen, n_parent, c_parent).requires_grad_()
    ea = mg.rnd(gen, nc, f, scale=0.5).requires_grad_()
    y = pool(xc, xp, index, edge_attr=ea, num_pool=n_parent)
    gw = mg.rnd(gen, *y.shape)
    (y * gw).sum().backward()
    out.update({f"{tag}__x_child": xc, f"{tag}__x_parent": xp, f"{tag}__index": index,
                f"{tag}__edge_attr": ea, f"{tag}__gw": gw, f"{tag}__out": y,
                f"{tag}__g_x_child": xc.grad, f"{tag}__g_edge_attr": ea.grad})
    if xp.grad is not None:
        out[f"{tag}__g_x_parent"] = xp.grad
    for k, p in pool.named_parameters():
        out[f"{tag}__p__{k}"] = p
        out[f"{tag}__g__{k}"] = p.grad


def main():
    U, N = mg.install_reference_import_hooks()
    import importlib
    P = importlib.import_module("src.nn.pool")
    gen = torch.Generator().manual_seed(4242)
    torch.manual_seed(4242)
    out = {}
    run("a", P.AttentivePool(dim=64, q_in_dim=48, num_heads=16, in_dim=40, out_dim=96, qk_dim=4,
                             in_rpe_dim=9, k_rpe=True, q_rpe=True),
        gen, 700, 90, 40, 48, 9, out, empty_parents=3)
    run("b", P.AttentivePool(dim=32, q_in_dim=32, num_heads=4, qk_dim=8, qk_scale="d+g",
                             in_rpe_dim=5, k_rpe=True, q_rpe=True, heads_share_rpe=True),
        gen, 300, 41, 32, 32, 5, out)
    run("c", P.AttentivePoolWithLearntQueries(dim=32, num_heads=8, qk_dim=2, qk_scale=0.7,
                                              in_rpe_dim=6, k_rpe=True, qkv_bias=False),
        gen, 250, 33, 32, 7, 6, out)
    mg.save("attentive_pool.npz", **out)


if __name__ == "__main__":
    main()
