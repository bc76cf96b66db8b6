"""The transformer block's pre-norm folded into the qkv Linear's read of x, its residual into
the out_proj Linear's epilogue, and the two gradients of x summed inside the norm's backward
pass (src/nn/transformer.py:231-249): the same arithmetic in the same order as the op-by-op
route, so outputs and every gradient must be BITWISE equal to it."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(n, e, g):
    s = torch.randint(0, n, (e,), generator=g)
    t = torch.randint(0, n, (e,), generator=g)
    loops = torch.arange(n)
    return torch.stack([torch.cat([s, loops]), torch.cat([t, loops])])


def _run(block, x, ni, ei, ea, B, fused):
    from superpoint_transformer_amd import ops
    prev = ops.fuse_prenorm(fused)
    try:
        b = copy.deepcopy(block)
        xx = x.clone().requires_grad_()
        eaa = ea.clone().requires_grad_()
        out, _, _ = b(xx, ni, edge_index=ei, edge_attr=eaa, num_graphs=B)
        (out * torch.linspace(-1, 1, out.numel(), device=out.device).view_as(out)).sum().backward()
        return out.detach(), xx.grad, eaa.grad, {k: p.grad for k, p in b.named_parameters()}
    finally:
        ops.fuse_prenorm(prev)


@pytest.mark.parametrize("B", [1, 3])
@pytest.mark.parametrize("dim,no_ffn", [(64, True), (128, False)])
def test_fused_prenorm_block_is_bitwise_the_unfused_block(dev, B, dim, no_ffn):
    from superpoint_transformer_amd.nn import TransformerBlock
    g = torch.Generator().manual_seed(5)
    n, e = 9_000, 60_000
    x = torch.randn(n, dim, generator=g).to(dev)
    ei = _graph(n, e, g).to(dev)
    ea = (torch.randn(ei.shape[1], 32, generator=g) * 0.3).to(dev)
    ni = None if B == 1 else (torch.arange(n) * B // n).to(dev)          # sorted clouds
    block = TransformerBlock(dim, num_heads=16, qk_dim=4, in_rpe_dim=32, ffn_ratio=1, no_ffn=no_ffn,
                             k_rpe=True, q_rpe=True, v_rpe=True).to(dev)
    with torch.no_grad():
        for p in block.parameters():                   # norms away from their (1, 0, 1) init
            p.add_(torch.randn(p.shape, generator=g).to(dev) * 0.05)
    o1, gx1, ge1, gp1 = _run(block, x, ni, ei, ea, B, True)
    o0, gx0, ge0, gp0 = _run(block, x, ni, ei, ea, B, False)
    assert torch.equal(o1, o0)
    assert torch.equal(gx1, gx0)
    assert torch.equal(ge1, ge0)
    for k in gp0:
        assert torch.equal(gp1[k], gp0[k]), k


def test_fused_route_is_taken_and_skipped(dev):
    """The fused route runs where it is built and hands over elsewhere (few rows, a generic norm
    index, training-time DropPath)."""
    from superpoint_transformer_amd import ops
    from superpoint_transformer_amd.nn import TransformerBlock
    g = torch.Generator().manual_seed(1)
    block = TransformerBlock(64, num_heads=16, qk_dim=4, in_rpe_dim=32, no_ffn=True,
                             k_rpe=True, q_rpe=True, v_rpe=True).to(dev)
    n = 6_000
    x = torch.randn(n, 64, generator=g).to(dev)
    ei = _graph(n, 30_000, g).to(dev)
    ea = torch.randn(ei.shape[1], 32, generator=g).to(dev)
    assert block.sa.forward_prenorm_residual(x, block.sa_norm, None, None, ei, ea) is not None
    assert block.sa.forward_prenorm_residual(x[:100], block.sa_norm, None, None,
                                             _graph(100, 300, g).to(dev), None) is None
    block.sa_norm.generic = True
    assert block.sa.forward_prenorm_residual(x, block.sa_norm, None, None, ei, ea) is None
    block.sa_norm.generic = False
    prev = ops.fuse_prenorm(False)
    try:
        assert block.sa.forward_prenorm_residual(x, block.sa_norm, None, None, ei, ea) is None
    finally:
        ops.fuse_prenorm(prev)


def test_linear_residual_and_norm_linear_against_f64(dev):
    """The two fused entries against a float64 evaluation (the bitwise test above pins them on
    the unfused kernels; this one pins them on the math): 1e-5 relative to the row scale."""
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(9)
    rows, K, N, B = 20_011, 64, 192, 4                  # a ragged last tile
    x = torch.randn(rows, K, generator=g) * 2 + 0.5
    batch = (torch.arange(rows) * B // rows)
    W = torch.randn(N, K, generator=g) * 0.2
    b = torch.randn(N, generator=g) * 0.1
    gw, gb, ga = (torch.rand(K, generator=g) + 0.5 for _ in range(3))
    xd = x.to(dev).requires_grad_()
    Wd, bd = W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    pw, pb, pa = (t.to(dev).requires_grad_() for t in (gw, gb, ga))
    y, xres = ops.norm_linear(xd, batch.to(dev), B, pw, pb, pa, 1e-5, Wd, bd)
    Wo = (torch.randn(K, 64, generator=g) * 0.1).to(dev).requires_grad_()
    assert ops.linear_residual_ok(xres, Wo)
    out = ops.linear_residual(y[:, 64:128].contiguous(), Wo, None, xres)   # residual = x itself, like a block
    gout = torch.randn(rows, K, generator=g)
    (out * gout.to(dev)).sum().backward()

    x64 = x.double().requires_grad_()
    W64, b64 = W.double().requires_grad_(), b.double().requires_grad_()
    w64, bb64, a64 = (t.double().requires_grad_() for t in (gw, gb, ga))
    yn = torch.empty(rows, K, dtype=torch.float64)
    parts = []
    for gi in range(B):
        m = batch == gi
        xg = x64[m]
        o = xg - a64 * xg.mean(0)
        parts.append(w64 * o / (o.pow(2).mean(0) + 1e-5).sqrt() + bb64)
    yn = torch.cat(parts)
    y64 = yn @ W64.t() + b64
    out64 = x64 + y64[:, 64:128] @ Wo.detach().cpu().double().t()
    (out64 * gout.double()).sum().backward()
    tol = lambda r: 1e-5 * max(1.0, float(r.abs().max()))
    assert (y.detach().cpu().double() - y64.detach()).abs().max() < tol(y64)
    assert (out.detach().cpu().double() - out64.detach()).abs().max() < tol(out64)
    assert (xd.grad.cpu().double() - x64.grad).abs().max() < 2e-5 * float(x64.grad.abs().max())
    assert (Wd.grad.cpu().double() - W64.grad).abs().max() < 2e-5 * float(W64.grad.abs().max())
    assert (bd.grad.cpu().double() - b64.grad).abs().max() < 2e-5 * float(b64.grad.abs().max())
    for got, ref in ((pw, w64), (pb, bb64), (pa, a64)):
        assert (got.grad.cpu().double() - ref.grad).abs().max() < 5e-5 * float(ref.grad.abs().max())
