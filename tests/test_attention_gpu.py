"""GPU parity: fused edge attention (fwd + bwd) and the Stage-level operator
surface against fixtures produced by the REFERENCE's own modules
(tests/golden/make_golden.py: SelfAttentionBlock / DownNFuseStage /
UpNFuseStage of /root/reference run in float64) and against the float64
oracle on random graphs.

Stated tolerance (f32 kernel vs f64 reference): |err| <= 1e-5 + 1e-4 * |ref|
for the block output and input gradients; parameter gradients (sums over all
edges / nodes) are checked relative to the largest entry of the same tensor."""
import pytest
import torch

from conftest import load_golden, t64, tl
from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


def _check(a, ref, name, rel_to_max=False):
    a = a.detach().cpu().double()
    ref = ref.detach().double()
    assert a.shape == ref.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(ref.shape)}"
    if rel_to_max:
        # sums of O(E) terms of magnitude ~0.1 that may cancel exactly (e.g. the
        # k_rpe bias: softmax is shift invariant, its true gradient is 0)
        scale = ref.abs().max().clamp(min=5e-2)
        err = ((a - ref).abs() / scale).max().item()
        assert err <= 2e-5, f"{name}: max err / max|ref| = {err:.3e}"
    else:
        err = ((a - ref).abs() - RTOL * ref.abs()).max().item()
        assert err <= ATOL, f"{name}: |err| - rtol|ref| = {err:.3e}"


def _load_params(module, g, dev):
    sd = {k[3:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("p__")}
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return module.to(dev)


@pytest.mark.parametrize("name,dim", [("attention_spt64.npz", 64), ("attention_spt128.npz", 128)])
def test_self_attention_block_matches_reference_fixture(name, dim, dev):
    from superpoint_transformer_amd import nn as N
    g = load_golden(name)
    blk = N.SelfAttentionBlock(dim, num_heads=int(g["num_heads"]), out_dim=dim,
                               qk_dim=int(g["qk_dim"]), in_rpe_dim=g["edge_attr"].shape[1],
                               k_rpe=True, q_rpe=True, v_rpe=True)
    blk = _load_params(blk, g, dev)           # reference parameter names, strict
    x = torch.from_numpy(g["x"]).float().to(dev).requires_grad_()
    ea = torch.from_numpy(g["edge_attr"]).float().to(dev).requires_grad_()
    ei = tl(g["edge_index"]).to(dev)
    out = blk(x, ei, edge_attr=ea)
    (out * torch.from_numpy(g["gw"]).float().to(dev)).sum().backward()
    _check(out, t64(g["out"]), "out")
    _check(x.grad, t64(g["g_x"]), "g_x")
    _check(ea.grad, t64(g["g_edge_attr"]), "g_edge_attr")
    for k, p in blk.named_parameters():
        _check(p.grad, t64(g["g__" + k]), "g_" + k, rel_to_max=True)


def _rand_graph(gen, n, deg):
    m = int(n * deg / 2)
    a = torch.randint(0, n, (m,), generator=gen)
    b = torch.randint(0, n, (m,), generator=gen)
    keep = a != b
    a, b = a[keep], b[keep]
    loops = torch.arange(n)
    return torch.stack([torch.cat([a, b, loops]), torch.cat([b, a, loops])])


CASES = [
    # n, deg, H, D, Dv, F, which rpe, scale
    (300, 10.0, 16, 4, 4, 32, "kqv", None),
    (200, 25.0, 16, 4, 8, 32, "kqv", None),      # SPT-128: value dim 8
    (150, 8.0, 32, 4, 4, 32, "kqv", "d+g"),      # scannet: 32 heads
    (140, 9.0, 32, 4, 8, 32, "kqv", None),       # both: 2 head groups x 2 value slices (matrix pipe only)
    (120, 6.0, 16, 2, 1, 18, "kqv", "d"),        # nano: qk_dim 2, dim 16, raw 18-D edge features
    (130, 7.0, 16, 2, 1, 16, "kqv", None),       # nano-2: qk_dim 2, dim 16, edge MLP to 16
    (100, 5.0, 16, 4, 4, 32, "k", "g"),
    (100, 5.0, 16, 4, 4, 32, "v", 0.37),
    (100, 5.0, 8, 8, 8, 32, "", None),           # no RPE at all
    (50, 70.0, 16, 4, 4, 32, "kqv", None),       # degree > several tiles
]


@pytest.mark.parametrize("n,deg,H,D,Dv,F,rpe,scale", CASES)
def test_edge_attention_vs_oracle(n, deg, H, D, Dv, F, rpe, scale, dev):
    from superpoint_transformer_amd import nn as N
    gen = torch.Generator().manual_seed(n * 7 + H + F)
    dim = H * Dv
    ei = _rand_graph(gen, n, deg)
    # isolated node (no outgoing edge) + unsorted sources
    ei = ei[:, ei[0] != 3]
    ei = ei[:, torch.randperm(ei.shape[1], generator=gen)]
    E = ei.shape[1]
    blk = N.SelfAttentionBlock(dim, num_heads=H, out_dim=None, qk_dim=D, qk_scale=scale,
                               in_rpe_dim=F, k_rpe="k" in rpe, q_rpe="q" in rpe,
                               v_rpe="v" in rpe).to(dev)
    x = torch.randn(n, dim, generator=gen)
    ea = torch.randn(E, F, generator=gen) * 0.5
    gw = torch.randn(n, dim, generator=gen)
    xd = x.to(dev).requires_grad_()
    ead = ea.to(dev).requires_grad_()
    out = blk(xd, ei.to(dev), edge_attr=ead)
    (out * gw.to(dev)).sum().backward()

    p = {k: v.detach().cpu().double().requires_grad_() for k, v in blk.named_parameters()}
    x64 = x.double().requires_grad_()
    ea64 = ea.double().requires_grad_()

    # oracle with the requested qk scaling
    def scale_fn(s):
        dd = (dim // H) ** -0.5
        gg = (s.bincount(minlength=n).double() ** -0.5)[s].view(-1, 1, 1)
        if scale is None:
            return dd * gg
        if scale == "d+g":
            return dd + gg
        if scale == "d":
            return dd
        if scale == "g":
            return gg
        return scale
    old = O.qk_scale_dg
    O.qk_scale_dg = lambda s, d_, h_: torch.as_tensor(scale_fn(s), dtype=torch.float64)
    try:
        ref = O.self_attention(x64, ei, ea64, p, H, D)
    finally:
        O.qk_scale_dg = old
    (ref * gw.double()).sum().backward()
    _check(out, ref, "out")
    assert out[3].abs().sum().item() == 0          # node without edges -> zeros
    _check(xd.grad, x64.grad, "g_x")
    if rpe:
        _check(ead.grad, ea64.grad, "g_edge_attr")
    else:
        assert ead.grad is None
    for k, v in blk.named_parameters():
        _check(v.grad, p[k].grad, "g_" + k, rel_to_max=True)


def test_edge_attention_forward_is_deterministic_and_order_invariant(dev):
    from superpoint_transformer_amd import ops
    gen = torch.Generator().manual_seed(11)
    n, H, D = 500, 16, 4
    ei = _rand_graph(gen, n, 12.0)
    E = ei.shape[1]
    qkv = torch.randn(n, 192, generator=gen).to(dev)
    ea = (torch.randn(E, 32, generator=gen) * 0.5).to(dev)
    W = [(torch.randn(64, 32, generator=gen).to(dev) * 0.1, torch.randn(64, generator=gen).to(dev) * 0.1)
         for _ in range(3)]
    a = ops.edge_attention(qkv, ei.to(dev), ea, *W, num_heads=H, qk_dim=D, scale_a=0.5)
    b = ops.edge_attention(qkv, ei.to(dev).clone(), ea, *W, num_heads=H, qk_dim=D, scale_a=0.5)
    assert torch.equal(a, b)
    perm = torch.randperm(E, generator=gen)
    c = ops.edge_attention(qkv, ei[:, perm].to(dev), ea[perm.to(dev)], *W, num_heads=H,
                           qk_dim=D, scale_a=0.5)
    torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-6)


def test_edge_attention_rejects_unbuilt_shapes_loudly(dev):
    from superpoint_transformer_amd import ops
    qkv = torch.randn(10, 2 * 40 * 4 + 40 * 4, device=dev)      # H*D = 160 > 128
    ei = torch.stack([torch.arange(10), torch.arange(10)]).to(dev)
    with pytest.raises(RuntimeError, match="unsupported attention shape"):
        ops.edge_attention(qkv, ei, None, num_heads=40, qk_dim=4)


def _stage_case(g, stage, dev, kind):
    f = lambda k: torch.from_numpy(g[k]).float().to(dev)
    l = lambda k: tl(g[k]).to(dev)
    stage = _load_params(stage, g, dev)
    xc = f("x_child").requires_grad_()
    if kind == "down":
        out, diam = stage(f("x_parent"), xc, l("norm_index"), l("pool_index"), pos=f("pos"),
                          node_size=l("node_size"), super_index=l("super_index"),
                          edge_index=l("edge_index"), edge_attr=f("edge_attr"),
                          num_super=int(g["num_super"]))
        xp = None
    else:
        xp = f("x_parent").requires_grad_()
        out, diam = stage(xc, xp, l("norm_index"), l("unpool_index"), pos=f("pos"),
                          node_size=l("node_size"), super_index=l("super_index"),
                          edge_index=l("edge_index"), edge_attr=f("edge_attr"))
    (out * f("gw")).sum().backward()
    # a whole stage chains 2-3 GraphNorms, MLPs and attention blocks: 5x the single-op bar
    a = out.detach().cpu().double()
    ref = t64(g["out"])
    assert ((a - ref).abs() - 5e-4 * ref.abs()).max().item() <= 5e-5
    _check(xc.grad, t64(g["g_x_child"]), "g_x_child", rel_to_max=True)
    if xp is not None:
        _check(xp.grad, t64(g["g_x_parent"]), "g_x_parent", rel_to_max=True)
    for k, p in stage.named_parameters():
        ref = t64(g["g__" + k])
        scale = ref.abs().max().clamp(min=1e-3)
        err = ((p.grad.detach().cpu().double() - ref).abs() / scale).max().item()
        assert err <= 2e-4, f"g_{k}: {err:.3e}"
    return diam


def test_down_stage_matches_reference_fixture(dev):
    from superpoint_transformer_amd import nn as N
    g = load_golden("down_stage.npz")
    dim = 64
    stage = N.DownNFuseStage(
        dim, num_blocks=2, num_heads=16, in_mlp=[4 + 3 + 128, dim, dim],
        mlp_norm=N.GraphNorm, qk_dim=4, k_rpe=True, q_rpe=True, v_rpe=True, in_rpe_dim=32,
        norm=N.GraphNorm, no_ffn=True, pool="max", fusion="cat", use_pos=True,
        use_diameter_parent=False, version_holder=N.VersionHolder("3.0.0"))
    diam = _stage_case(g, stage, dev, "down")
    assert torch.equal(diam.cpu(), t64(g["diameter"]).float())


def test_up_stage_matches_reference_fixture(dev):
    from superpoint_transformer_amd import nn as N
    g = load_golden("up_stage.npz")
    dim = 64
    stage = N.UpNFuseStage(
        dim, num_blocks=1, num_heads=16, in_mlp=[dim + 3 + dim, dim, dim],
        mlp_norm=N.GraphNorm, qk_dim=4, k_rpe=True, q_rpe=True, v_rpe=True, in_rpe_dim=32,
        norm=N.GraphNorm, no_ffn=False, ffn_ratio=1, unpool="index", fusion="cat",
        use_pos=True, version_holder=N.VersionHolder("3.0.0"))
    _stage_case(g, stage, dev, "up")


def _degree_graph(gen, degrees):
    """Directed edges with prescribed out-degrees (targets random), shuffled."""
    n = len(degrees)
    src = torch.repeat_interleave(torch.arange(n), torch.tensor(degrees))
    tgt = torch.randint(0, n, (src.numel(),), generator=gen)
    p = torch.randperm(src.numel(), generator=gen)
    return torch.stack([src[p], tgt[p]])


@pytest.mark.parametrize("degrees", [
    [1] * 40 + [2] * 30 + [3] * 20,                       # up to 16 nodes in one 16-edge tile
    [0, 5, 0, 0, 7, 16, 0, 17, 1, 0, 31, 32, 33, 0, 2],   # edge-less nodes between the others
    [16] * 12,                                            # every tile exactly one node
    [15, 17] * 8 + [1, 100, 1, 3],                        # boundaries drifting through the tiles
    [0] * 5 + [9] + [0] * 5,                              # a single node with edges
])
@pytest.mark.parametrize("packed,target_order", [(2, 1), (2, 0), (1, None), (0, None)],
                         ids=["edge-lane-target-order", "edge-lane-source-order", "packed", "per-node"])
def test_attention_backward_tilings_match_oracle(degrees, packed, target_order, dev):
    """The edge-lane backward over the edge stream in target order (the default) and in source
    order, the packed backward (16-edge tiles over the edge stream, two node contexts per pass,
    extra passes when a tile holds more nodes) and the per-node tiling, against the f64 oracle
    on graphs built to hit the tile / node boundary cases (short last tiles included: the
    edge counts are 160, 144, 192, 361 and 9)."""
    from superpoint_transformer_amd import _lib, nn as N
    gen = torch.Generator().manual_seed(sum(degrees) + len(degrees))
    n, H, D, dim, F = len(degrees), 16, 4, 64, 32
    ei = _degree_graph(gen, degrees)
    E = ei.shape[1]
    blk = N.SelfAttentionBlock(dim, num_heads=H, out_dim=None, qk_dim=D, in_rpe_dim=F,
                               k_rpe=True, q_rpe=True, v_rpe=True).to(dev)
    x = torch.randn(n, dim, generator=gen)
    ea = torch.randn(E, F, generator=gen) * 0.5
    gw = torch.randn(n, dim, generator=gen)
    prev = _lib.lib.spt_attn_bwd_packed(packed)
    prev_to = _lib.lib.spt_attn_bwd_el_target_order(target_order) if target_order is not None else None
    try:
        xd = x.to(dev).requires_grad_()
        ead = ea.to(dev).requires_grad_()
        out = blk(xd, ei.to(dev), edge_attr=ead)
        (out * gw.to(dev)).sum().backward()
    finally:
        _lib.lib.spt_attn_bwd_packed(prev)
        if prev_to is not None:
            _lib.lib.spt_attn_bwd_el_target_order(prev_to)
    p = {k: v.detach().cpu().double().requires_grad_() for k, v in blk.named_parameters()}
    x64, ea64 = x.double().requires_grad_(), ea.double().requires_grad_()
    deg = torch.tensor(degrees).double().clamp(min=1)
    old = O.qk_scale_dg
    O.qk_scale_dg = lambda s, d_, h_: ((dim // H) ** -0.5 * deg[s] ** -0.5).view(-1, 1, 1)
    try:
        ref = O.self_attention(x64, ei, ea64, p, H, D)
    finally:
        O.qk_scale_dg = old
    (ref * gw.double()).sum().backward()
    _check(out, ref, "out")
    _check(xd.grad, x64.grad, "g_x")
    _check(ead.grad, ea64.grad, "g_edge_attr")
    for k, v in blk.named_parameters():
        _check(v.grad, p[k].grad, "g_" + k, rel_to_max=True)


@pytest.mark.parametrize("mode", [2, 1, 0])
def test_shared_edge_attr_gradient_of_a_stage(mode, dev):
    """Three chained blocks reading one edge_attr: with the stage's shared gradient buffer
    (the blocks' d edge_attr accumulate in the kernel, the first block hands the sum to
    autograd) every gradient equals the one autograd sums from three separate tensors - on the
    packed / per-node MFMA kernels (mode 2), the f32 pipe (1) and the VALU kernels (0); a second
    backward through the same graph (retain_graph) restarts the buffer."""
    from superpoint_transformer_amd import _lib, ops, nn as N
    gen = torch.Generator().manual_seed(31)
    n, H, D, dim, F = 500, 16, 4, 64, 32
    ei = _rand_graph(gen, n, 12.0).to(dev)
    blocks = [N.SelfAttentionBlock(dim, num_heads=H, out_dim=None, qk_dim=D, in_rpe_dim=F,
                                   k_rpe=True, q_rpe=True, v_rpe=True).to(dev) for _ in range(3)]
    x = torch.randn(n, dim, generator=gen).to(dev)
    ea = (torch.randn(ei.shape[1], F, generator=gen) * 0.5).to(dev)
    gw = torch.randn(n, dim, generator=gen).to(dev)

    def run(share, twice=False, partial_first=False):
        xd, ead = x.clone().requires_grad_(), ea.clone().requires_grad_()
        for b in blocks:
            b.zero_grad()
        hs = [xd]
        for b in blocks:
            hs.append(hs[-1] + b(hs[-1], ei, edge_attr=ead, ea_grad=share))
        loss = (hs[-1] * gw).sum()
        if partial_first:
            # a walk that stops short of the first block (gradient wrt the LAST block's input
            # only) leaves the share half-built; the full walk afterwards must start afresh
            torch.autograd.grad(loss, hs[-2], retain_graph=True)
        if twice:
            loss.backward(retain_graph=True)
            xd.grad = ead.grad = None
            for b in blocks:
                b.zero_grad()
        loss.backward()
        return [xd.grad, ead.grad] + [p.grad.clone() for b in blocks for p in b.parameters()]

    prev = _lib.lib.spt_attn_use_mfma(mode)
    try:
        ref = run(None)
        got = run(ops.EdgeAttrGradShare())
        again = run(ops.EdgeAttrGradShare(), twice=True)
        stale = run(ops.EdgeAttrGradShare(), partial_first=True)
    finally:
        _lib.lib.spt_attn_use_mfma(prev)
    for a, b, c, e in zip(ref, got, again, stale):
        tol = 1e-5 * float(a.abs().max())                  # atomics: summation order only
        assert float((a - b).abs().max()) <= tol
        assert float((a - c).abs().max()) <= tol
        assert float((a - e).abs().max()) <= tol
    with pytest.raises(ValueError):
        sh = ops.EdgeAttrGradShare()
        xd = x.clone().requires_grad_()
        blocks[0](xd, ei, edge_attr=ea.clone().requires_grad_(), ea_grad=sh)
        blocks[1](xd, ei, edge_attr=ea.clone().requires_grad_(), ea_grad=sh)


# ---------------------------------------------------------------------------
# The options outside the fused kernel (delta RPE, attention dropout): the op-by-op route
# ---------------------------------------------------------------------------
class _FixedDrop(torch.nn.Module):
    """``a -> a * mask / (1 - p)`` in place of nn.Dropout: the same stand-in the fixture's
    script gave the reference's module."""

    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, a):
        return a * self.mask.to(a.dtype) / (1 - self.p)


def _option_block(tag):
    from superpoint_transformer_amd import nn as N
    if tag == "d":
        return N.SelfAttentionBlock(64, num_heads=16, in_dim=48, out_dim=64, qk_dim=4,
                                    in_rpe_dim=18, k_rpe=True, q_rpe=True, v_rpe=True,
                                    k_delta_rpe=True, q_delta_rpe=True)
    if tag == "s":
        return N.SelfAttentionBlock(32, num_heads=4, qk_dim=8, qk_scale="d+g", in_rpe_dim=7,
                                    k_rpe=True, q_rpe=True, k_delta_rpe=True, q_delta_rpe=True,
                                    qk_share_rpe=True, q_on_minus_rpe=True, heads_share_rpe=True)
    return N.SelfAttentionBlock(64, num_heads=16, out_dim=64, qk_dim=4, in_rpe_dim=32,
                                k_rpe=True, q_rpe=True, v_rpe=True)


@pytest.mark.parametrize("tag", ["d", "s", "m"])
@pytest.mark.parametrize("as_csr", [False, True])
def test_delta_rpe_and_attention_dropout_match_the_reference(tag, as_csr, dev):
    """src/nn/attention.py:255-290 (RPE from x[target] - x[source], shared / negated variants)
    and :309-311 (dropout on the attention weights; fixed keep-mask) against the reference's
    own module (tests/golden/make_golden_attention_options.py)."""
    from superpoint_transformer_amd.csr import edge_csr_of
    G = load_golden("attention_options.npz")
    g = {k[3:]: v for k, v in G.items() if k.startswith(tag + "__")}
    blk = _option_block(tag)
    sd = {k[3:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("p__")}
    blk.load_state_dict(sd, strict=True)                      # the reference's parameter names
    blk = blk.to(dev)
    ei = tl(g["edge_index"]).to(dev)
    if tag == "m":
        if as_csr:
            pytest.skip("the fixture's mask is given in edge order")
        blk.attn_drop = _FixedDrop(torch.from_numpy(g["mask"]).to(dev), float(g["p"]))
    x = torch.from_numpy(g["x"]).float().to(dev).requires_grad_()
    ea = torch.from_numpy(g["edge_attr"]).float().to(dev).requires_grad_()
    out = blk(x, edge_csr_of(ei, x.shape[0]) if as_csr else ei, edge_attr=ea)
    (out * torch.from_numpy(g["gw"]).float().to(dev)).sum().backward()
    _check(out, t64(g["out"]), "out")
    _check(x.grad, t64(g["g_x"]), "g_x")
    _check(ea.grad, t64(g["g_edge_attr"]), "g_edge_attr")
    for k, p in blk.named_parameters():
        _check(p.grad, t64(g["g__" + k]), "g_" + k, rel_to_max=True)


def test_attention_dropout_is_the_fused_kernel_in_eval_and_unbiased_in_training(dev):
    from superpoint_transformer_amd import nn as N
    gen = torch.Generator().manual_seed(17)
    n = 400
    ei = _rand_graph(gen, n, 12.0).to(dev)
    x = torch.randn(n, 64, generator=gen).to(dev)
    ea = (torch.randn(ei.shape[1], 32, generator=gen) * 0.5).to(dev)
    torch.manual_seed(1)
    plain = N.SelfAttentionBlock(64, num_heads=16, qk_dim=4, in_rpe_dim=32, k_rpe=True,
                                 q_rpe=True, v_rpe=True).to(dev)
    drop = N.SelfAttentionBlock(64, num_heads=16, qk_dim=4, in_rpe_dim=32, k_rpe=True,
                                q_rpe=True, v_rpe=True, attn_drop=0.25).to(dev)
    drop.load_state_dict(plain.state_dict())
    with torch.no_grad():
        ref = plain(x, ei, edge_attr=ea)
        assert torch.equal(drop.eval()(x, ei, edge_attr=ea), ref)      # same kernel, same bits
        drop.train()
        a, b = drop(x, ei, edge_attr=ea), drop(x, ei, edge_attr=ea)
        assert not torch.equal(a, b)                                     # fresh masks
        mean = torch.stack([drop(x, ei, edge_attr=ea) for _ in range(200)]).mean(0)
    # E[dropout(a)] = a: the mean of 200 draws is within a few standard errors of the plain output
    assert float((mean - ref).abs().max()) < 0.25 * float(ref.abs().max())
    assert float((mean - ref).abs().mean()) < 0.03 * float(ref.abs().mean()) + 0.01


@pytest.mark.parametrize("H,Dv", [(16, 8), (32, 4)])
def test_wider_head_layouts_decompose_onto_the_matrix_pipe_kernels(H, Dv, dev):
    """SPT-128 (KITTI-360: 16 heads, value dim 8) and 32 heads (ScanNet) run as G x J passes
    of the built (16, 4, 4, 32) matrix-pipe kernels (ops._matrix_pipe_split): heads are independent,
    the value dims of a head only enter linearly - the decomposition is exact.
    Against the generic VALU kernels of the same call (precision 0), forward and every gradient."""
    from superpoint_transformer_amd import _lib, ops, precision
    gen = torch.Generator().manual_seed(H * 10 + Dv)
    n, D, F = 400, 4, 32
    ei = _rand_graph(gen, n, 12.0)
    E = ei.shape[1]
    qkv = torch.randn(n, 2 * H * D + H * Dv, generator=gen)
    ea = torch.randn(E, F, generator=gen) * 0.4
    W = [(torch.randn(c, F, generator=gen) * 0.1, torch.randn(c, generator=gen) * 0.1)
         for c in (H * D, H * D, H * Dv)]
    gw = torch.randn(n, H * Dv, generator=gen)

    def run(mode_name):
        q = qkv.to(dev).requires_grad_()
        e = ea.to(dev).requires_grad_()
        ps = [(w.to(dev).requires_grad_(), b.to(dev).requires_grad_()) for w, b in W]
        if mode_name is None:                               # generic kernels: process default 0
            prev = _lib.lib.spt_attn_use_mfma(0)
            try:
                assert ops._matrix_pipe_split(q, e, ps[0][0], ps[1][0], ps[2][0], H, D) is None
                out = ops.edge_attention(q, ei.to(dev), e, *ps, num_heads=H, qk_dim=D, scale_a=0.7)
                (out * gw.to(dev)).sum().backward()
            finally:
                _lib.lib.spt_attn_use_mfma(prev)
        else:
            with precision.matrix_precision(mode_name):
                split = ops._matrix_pipe_split(q, e, ps[0][0], ps[1][0], ps[2][0], H, D)
                assert split == (H // 16, Dv // 4, Dv)
                out = ops.edge_attention(q, ei.to(dev), e, *ps, num_heads=H, qk_dim=D, scale_a=0.7)
                (out * gw.to(dev)).sum().backward()
        return [out.detach(), q.grad, e.grad] + [t.grad for p in ps for t in p]

    ref = run(None)
    got = run("f32")
    for a, r in zip(got, ref):
        assert (a - r).abs().max().item() <= 1e-5 + 2e-4 * r.abs().max().item()


@pytest.mark.parametrize("G,J", [(1, 2), (2, 1), (2, 2), (1, 3)])
def test_split_operand_kernels_equal_the_index_and_cat_formulation(G, J, dev):
    """spt_attn_split_pack_f32 / spt_attn_split_grad_f32 (the operand slabs of the head-group
    passes and the gradient's way back, one kernel each) against the torch formulation they
    replace: one index gather per pass layout, a sum over the value slices, permutes and a cat.
    Copies bit for bit; the q / k sums bit for bit for two slices (a + b), 1 ulp-level for three."""
    from superpoint_transformer_amd import _lib, ops
    gen = torch.Generator().manual_seed(100 * G + J)
    n, H = 777, 16 * G
    QK, Dv, P = 4 * H, 4 * J, G * J
    qkv = torch.randn(n, 2 * QK + H * Dv, generator=gen).to(dev)
    cols = ops._EdgeAttentionSplit._cols(H, G, J, dev)
    ref = qkv[:, cols.view(-1)].view(n, P, 192).transpose(0, 1).contiguous()
    qa = torch.full((P, n, 192), float("nan"), device=dev)
    st = _lib.lib.spt_attn_split_pack_f32(_lib.ptr(qkv), n, G, J, _lib.ptr(qa), _lib.stream_ptr(dev))
    _lib.check(st, "spt_attn_split_pack_f32")
    assert torch.equal(qa, ref)
    gqa = torch.randn(P, n, 192, generator=gen).to(dev)
    g5 = gqa.view(G, J, n, 192)
    gqk = g5[..., :128].sum(1) if J > 1 else g5[:, 0, :, :128]
    gref = torch.cat([gqk[..., :64].permute(1, 0, 2).reshape(n, QK),
                      gqk[..., 64:].permute(1, 0, 2).reshape(n, QK),
                      g5[..., 128:].reshape(G, J, n, 16, 4).permute(2, 0, 3, 1, 4).reshape(n, H * Dv)], 1)
    got = torch.full((n, 2 * QK + H * Dv), float("nan"), device=dev)
    st = _lib.lib.spt_attn_split_grad_f32(_lib.ptr(gqa), n, G, J, _lib.ptr(got), _lib.stream_ptr(dev))
    _lib.check(st, "spt_attn_split_grad_f32")
    assert torch.equal(got[:, 2 * QK:], gref[:, 2 * QK:])
    if J <= 2:
        assert torch.equal(got, gref)
    else:
        assert (got - gref).abs().max().item() <= 1e-6


@pytest.mark.parametrize("D,Dv,F", [(2, 1, 16), (4, 2, 32), (2, 4, 18)])
def test_narrower_head_layouts_are_zero_padded_onto_the_matrix_pipe_kernels(D, Dv, F, dev):
    """The nano family (configs/model/semantic/nano-2.yaml: 16 heads of qk_dim 2, value dim 1,
    16-D edge encodings) and other layouts NARROWER than the built (16, 4, 4, 32) one run on the
    matrix-pipe kernels with zero-padded operands (ops._matrix_pipe_pad: padded q / k dims add
    zero to the dot products, padded value dims and encoding columns meet zero weights, padded
    outputs are dropped - exact).  Against the generic VALU kernels of the same call, forward and
    every gradient."""
    from superpoint_transformer_amd import _lib, ops, precision
    gen = torch.Generator().manual_seed(D * 100 + Dv * 10 + F)
    n, H = 500, 16
    ei = _rand_graph(gen, n, 12.0)
    E = ei.shape[1]
    qkv = torch.randn(n, 2 * H * D + H * Dv, generator=gen)
    ea = torch.randn(E, F, generator=gen) * 0.4
    W = [(torch.randn(c, F, generator=gen) * 0.1, torch.randn(c, generator=gen) * 0.1)
         for c in (H * D, H * D, H * Dv)]
    gw = torch.randn(n, H * Dv, generator=gen)
    old_min = ops.PAD_MIN_EDGES

    def run(padded):
        q = qkv.to(dev).requires_grad_()
        e = ea.to(dev).requires_grad_()
        ps = [(w.to(dev).requires_grad_(), b.to(dev).requires_grad_()) for w, b in W]
        ops.PAD_MIN_EDGES = 0 if padded else 1 << 60
        try:
            with precision.matrix_precision("f32"):
                ecsr = ops.edge_csr_of(ei.to(dev), n)
                took = ops._matrix_pipe_pad(q, ecsr, e, ps[0][0], ps[0][1], ps[1][0], ps[1][1],
                                            ps[2][0], ps[2][1], H, D) is not None
                assert took == padded
                out = ops.edge_attention(q, ei.to(dev), e, *ps, num_heads=H, qk_dim=D, scale_a=0.7)
                (out * gw.to(dev)).sum().backward()
        finally:
            ops.PAD_MIN_EDGES = old_min
        return [out.detach(), q.grad, e.grad] + [t.grad for p in ps for t in p]

    ref = run(False)
    got = run(True)
    assert got[0].shape == (n, H * Dv)
    for a, r in zip(got, ref):
        assert (a - r).abs().max().item() <= 1e-5 + 2e-4 * r.abs().max().item()


@pytest.mark.parametrize("order", ["target", "source"])
def test_edge_lane_backward_stays_inside_an_exact_workspace_when_edges_are_fewer_than_nodes(order, dev):
    """Advisor (round 4, high): with e < n the target-order layout (640 B of records per node) is
    LARGER than the source-order one that `spt_edge_attn_bwd_ex_workspace_bytes` used to return -
    a caller allocating exactly that many bytes had ~10 MB written past its buffer.  Here the
    workspace handed to the op is EXACTLY the advertised size with a guard page behind it: the
    guard must survive, the query must cover the layout of either order, and the gradients of
    the two orders (chosen PER CALL through the mode word, precision.attention_backward_order)
    must agree with each other and with the f64 oracle."""
    from superpoint_transformer_amd import _lib, ops, precision, nn as N
    gen = torch.Generator().manual_seed(5)
    n, H, D, dim, F = 60_000, 16, 4, 64, 32
    e = 36_000                                             # fewer edges than nodes, no self loops
    src = torch.randint(0, n, (e,), generator=gen)
    tgt = torch.randint(0, n, (e,), generator=gen)
    ei = torch.stack([src, tgt])
    blk = N.SelfAttentionBlock(dim, num_heads=H, out_dim=None, qk_dim=D, in_rpe_dim=F,
                               k_rpe=True, q_rpe=True, v_rpe=True).to(dev)
    x = torch.randn(n, dim, generator=gen)
    ea = torch.randn(e, F, generator=gen) * 0.5
    gw = torch.randn(n, dim, generator=gen)
    nb = int(_lib.lib.spt_edge_attn_bwd_ex_workspace_bytes(n, e, H, D, D, F))
    tables = int(_lib.lib.spt_edge_attn_bwd_workspace_bytes(H, D, D, F))
    assert nb - tables >= n * 640 + e * 256               # the target-order layout fits
    GUARD = 1 << 16
    slab = torch.full((nb + GUARD,), 0xA5, dtype=torch.uint8, device=dev)
    real_ws = ops._workspace
    ops._workspace = lambda nbytes, d: (slab[:nb] if nbytes == nb else real_ws(nbytes, d))
    try:
        with precision.attention_backward_order(order):
            xd = x.to(dev).requires_grad_()
            ead = ea.to(dev).requires_grad_()
            out = blk(xd, ei.to(dev), edge_attr=ead)
            (out * gw.to(dev)).sum().backward()
        torch.cuda.synchronize()
    finally:
        ops._workspace = real_ws
    assert bool((slab[nb:] == 0xA5).all()), "the backward wrote past the advertised workspace"
    p = {k: v.detach().cpu().double().requires_grad_() for k, v in blk.named_parameters()}
    x64, ea64 = x.double().requires_grad_(), ea.double().requires_grad_()
    ref = O.self_attention(x64, ei, ea64, p, H, D)
    (ref * gw.double()).sum().backward()
    _check(out, ref, "out")
    # (60 000 node rows / 36 000 edge rows: the extreme element of the split-bf16 products - the
    # attention's and, since round 6, the qkv Linear's input gradient - sits at 1.3e-5 .. 1.5e-5 in
    # both orders alike, a hair over the element-wise bar the small graphs above meet: tensor scale here)
    _check(xd.grad, x64.grad, "g_x", rel_to_max=True)
    _check(ead.grad, ea64.grad, "g_edge_attr", rel_to_max=True)
    for k, v in blk.named_parameters():
        _check(v.grad, p[k].grad, "g_" + k, rel_to_max=True)
    # the process default was not touched by the per-call choice
    assert _lib.lib.spt_attn_bwd_el_target_order(-1) == 1


def _mirrored_graph(gen, n, deg, dev, loops=True):
    """[i<j | j>i | loops] as OnTheFlyHorizontalEdgeFeatures + NAGAddSelfLoops leave it, with the
    host knowledge their kernel leaves on the tensor (M pairs)."""
    m = int(n * deg / 2)
    a = torch.randint(0, n, (m,), generator=gen)
    b = torch.randint(0, n, (m,), generator=gen)
    keep = a != b
    lo, hi = torch.minimum(a[keep], b[keep]), torch.maximum(a[keep], b[keep])
    lp = torch.arange(n) if loops else torch.zeros(0, dtype=torch.long)
    ei = torch.stack([torch.cat([lo, hi, lp]), torch.cat([hi, lo, lp])]).to(dev)
    ei._spt_mirror_pairs = int(lo.numel())
    return ei


@pytest.mark.parametrize("n,deg,loops", [(5000, 16.0, True), (700, 3.0, False), (40, 90.0, True)])
def test_by_target_stream_from_the_mirror_structure(n, deg, loops, dev):
    """Round 6: the target-order tile records of a mirrored edge list come from the by-source view
    alone (spt_attn_mirror_prepare + spt_attn_pack_tile_ids_mirror: no second sort).  The records
    must describe the SAME edges as the sort-based ones - every stream position one edge with its
    (row, target, source, source-order position), targets in ascending runs - and the attention
    backward on them must match the f64 oracle and the sort-based route."""
    from superpoint_transformer_amd import csr, nn as N
    gen = torch.Generator().manual_seed(n + int(deg))
    ei = _mirrored_graph(gen, n, deg, dev, loops)
    E = ei.shape[1]
    ec = csr.edge_csr_of(ei, n)
    assert ec.mirrored(1 << 6) and not ec.mirrored(2 << 6)
    rec = ec.tile_ids(1 << 6)                                   # target order, mirror route
    assert ec._tview is None                                    # no target view was sorted
    csr.verify_adopted(block=True)                              # the structure check passed
    old = csr.use_mirror_views(False)
    try:
        csr.forget(ei)
        ref = csr.edge_csr_of(ei, n).tile_ids(1 << 6)           # sort-based records
    finally:
        csr.use_mirror_views(old)

    def rows(r):
        r = r.view(-1, 4, 16).permute(0, 2, 1).reshape(-1, 4)[:E].cpu()     # [E, (row, tgt, src, pos)]
        return r
    a, b = rows(rec), rows(ref)
    assert bool((a[1:, 1] >= a[:-1, 1]).all())                  # grouped by target, ascending
    assert torch.equal(a[:, 1], b[:, 1])                        # the same targets at the same positions
    key = lambda r: r[torch.argsort(r[:, 0])]                   # one record per edge row: compare as sets
    assert torch.equal(key(a), key(b))
    eic = ei.cpu()
    assert torch.equal(eic[1][a[:, 0].long()], a[:, 1].long()) and torch.equal(eic[0][a[:, 0].long()], a[:, 2].long())

    H, D, dim, F = 16, 4, 64, 32
    blk = N.SelfAttentionBlock(dim, num_heads=H, out_dim=None, qk_dim=D, in_rpe_dim=F,
                               k_rpe=True, q_rpe=True, v_rpe=True).to(dev)
    x = torch.randn(n, dim, generator=gen)
    ea = torch.randn(E, F, generator=gen) * 0.5
    gw = torch.randn(n, dim, generator=gen)

    def run(mirror):
        old = csr.use_mirror_views(mirror)
        try:
            csr.forget(ei)
            xd, ead = x.to(dev).requires_grad_(), ea.to(dev).requires_grad_()
            for p_ in blk.parameters():
                p_.grad = None
            out = blk(xd, ei, edge_attr=ead)
            (out * gw.to(dev)).sum().backward()
            torch.cuda.synchronize()
            return out.detach(), xd.grad, ead.grad, {k: v.grad.clone() for k, v in blk.named_parameters()}
        finally:
            csr.use_mirror_views(old)

    om, gxm, gem, gpm = run(True)
    os_, gxs, ges, gps = run(False)
    csr.verify_adopted(block=True)
    p = {k: v.detach().cpu().double().requires_grad_() for k, v in blk.named_parameters()}
    x64, ea64 = x.double().requires_grad_(), ea.double().requires_grad_()
    refo = O.self_attention(x64, eic, ea64, p, H, D)
    (refo * gw.double()).sum().backward()
    assert torch.equal(om, os_)                                 # the forward does not see the route
    _check(om, refo, "out")
    _check(gxm, x64.grad, "g_x (mirror)")
    _check(gxs, x64.grad, "g_x (sorted)")
    # (thousands of edge rows: the extreme element of the split-bf16 products sits a hair over the
    # element-wise bar in EITHER route, see the exact-workspace test below: tensor scale here)
    _check(gem, ea64.grad, "g_edge_attr (mirror)", rel_to_max=True)
    _check(ges, ea64.grad, "g_edge_attr (sorted)", rel_to_max=True)
    for k in gpm:
        _check(gpm[k], p[k].grad, "g_" + k + " (mirror)", rel_to_max=True)


def test_a_wrong_mirror_hint_is_caught(dev):
    """The declared structure is checked on the device: an edge list whose halves are not mirrors
    raises StaleCSRError when the verdict is read, the hint is removed and the next batch sorts."""
    from superpoint_transformer_amd import csr, nn as N
    gen = torch.Generator().manual_seed(9)
    n = 900
    ei = _mirrored_graph(gen, n, 8.0, dev)
    M = ei._spt_mirror_pairs
    bad = ei.clone()
    bad[1, M + 5] = (bad[1, M + 5] + 1) % n                     # one flipped edge points elsewhere
    bad = bad.clone()                                           # (a fresh tensor: an edited one loses the hint)
    bad._spt_mirror_pairs = M
    blk = N.SelfAttentionBlock(64, num_heads=16, out_dim=None, qk_dim=4, in_rpe_dim=32,
                               k_rpe=True, q_rpe=True, v_rpe=True).to(dev)
    x = torch.randn(n, 64, device=dev, requires_grad=True)
    ea = torch.randn(bad.shape[1], 32, device=dev, requires_grad=True)
    blk(x, bad, edge_attr=ea).sum().backward()
    with pytest.raises(csr.StaleCSRError, match="mirror"):
        csr.verify_adopted(block=True)
    assert not hasattr(bad, "_spt_mirror_pairs")
    assert not csr.edge_csr_of(bad, n).mirrored(1 << 6)         # sorts from now on
    csr.verify_adopted(block=True)
