"""GPU parity: fused UnitSphereNorm and GraphNorm (+LeakyReLU) against the CPU
oracle (float64) and the golden fixture produced by the reference's norm.py.

Bars: diameters bit-exact (bounding boxes are order-free f32 min/max);
normalised positions, GraphNorm outputs and gradients within
1e-5 * max(1, |ref|) of the float64 oracle."""
import pytest
import torch

from conftest import load_golden, t64, tl
from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu


def _close(a, ref64, tol=1e-5):
    a = a.detach().cpu().double()
    err = (a - ref64).abs() / ref64.abs().clamp(min=1)
    assert err.max().item() <= tol, f"max scaled err {err.max().item():.3e}"


def test_unit_sphere_norm_matches_reference_fixture(dev):
    from superpoint_transformer_amd import ops
    g = load_golden("unit_sphere_norm.npz")
    pos64, idx, w, ns = t64(g["pos"]), tl(g["idx"]), tl(g["w"]), int(g["num_super"])
    pos = pos64.float().to(dev)
    for key, kw in (("unw", dict(idx=idx.to(dev), w=None, num_super=ns)),
                    ("w", dict(idx=idx.to(dev), w=w.to(dev), num_super=ns)),
                    ("none_w", dict(idx=None, w=w.to(dev))),
                    ("none", dict(idx=None, w=None))):
        o, d = ops.unit_sphere_norm(pos, **kw)
        _close(o, t64(g["out_" + key]))
        assert torch.equal(d.cpu(), t64(g["diam_" + key]).float()), key


@pytest.mark.parametrize("n,nseg,weighted", [(1, 1, False), (5000, 37, True), (100000, 2900, False),
                                             (100000, 40000, True), (30000, 3, True), (700, 900, False)])
def test_unit_sphere_norm_random(n, nseg, weighted, dev):
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(n + nseg)
    pos = (torch.randn(n, 3, generator=g) * 5 + 20).float()
    idx = torch.randint(0, nseg, (n,), generator=g)
    w = torch.randint(0, 300, (n,), generator=g) if weighted else None
    o, d = ops.unit_sphere_norm(pos.to(dev), idx.to(dev), None if w is None else w.to(dev), nseg)
    ro, rd = O.unit_sphere_norm(pos.double(), idx, w, nseg)
    _close(o, ro)
    assert torch.equal(d.cpu(), rd.float())          # f32 boxes: exact


@pytest.mark.parametrize("n,nseg,cx,weighted", [(1, 1, 8, False), (5000, 37, 8, True), (100003, 2900, 128, False),
                                                (30000, 3, 256, True), (700, 900, 4, False), (4000, None, 8, True)])
def test_unit_sphere_assemble_is_the_norm_followed_by_the_concatenation(n, nseg, cx, weighted, dev):
    """ops.unit_sphere_assemble (one writing pass) against ops.unit_sphere_norm + the gather of the
    parent diameter + torch.cat, the way Stage._inject builds the stage input: bit for bit, and the
    gradient reaching x is the column slice of the output's.  ``nseg=None``: no super_index (the
    last level: all rows normalised together, norm.py:86-110)."""
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(n + cx)
    pos = (torch.randn(n, 3, generator=g) * 5 + 20).float().to(dev)
    idx = None if nseg is None else torch.randint(0, nseg, (n,), generator=g).to(dev)
    w = torch.randint(0, 300, (n,), generator=g).to(dev) if weighted else None
    x = torch.randn(n, cx, generator=g).to(dev)
    gw = torch.randn(n, cx + 4, generator=g).to(dev)
    npos, diam = ops.unit_sphere_norm(pos, idx, w, nseg)
    dp = diam.repeat(n, 1) if idx is None else diam[idx]
    xr = x.clone().requires_grad_()
    ref = torch.cat([dp, npos, xr], 1)
    (ref * gw).sum().backward()
    assert ops.unit_sphere_assemble_ok(x, pos)
    xa = x.clone().requires_grad_()
    out, diam2 = ops.unit_sphere_assemble(xa, pos, idx, w, nseg)
    (out * gw).sum().backward()
    assert torch.equal(out, ref) and torch.equal(diam2, diam)
    assert torch.equal(xa.grad, xr.grad)
    assert not ops.unit_sphere_assemble_ok(x[:, :3], pos)        # 3 columns: the plain route


@pytest.mark.parametrize("r,d,B,slope", [(1000, 64, 1, 1.0), (5000, 32, 3, 0.01), (40000, 128, 4, 0.01),
                                         (777, 18, 2, 1.0), (3, 64, 2, 0.2), (20000, 132, 1, 0.01),
                                         (9000, 7, 5, 1.0),
                                         # more graphs than one 64 KiB LDS table holds (31 at d=128,
                                         # 15 at d=256): several graph windows, one launch each
                                         (60000, 128, 40, 0.01), (50000, 256, 70, 1.0)])
@pytest.mark.parametrize("sorted_batch", [True, False])
def test_graph_norm_forward_backward(r, d, B, slope, sorted_batch, dev):
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(r + d)
    x = (torch.randn(r, d, generator=g) * 2 + 3).float()
    batch = torch.randint(0, B, (r,), generator=g)
    if sorted_batch:
        batch = batch.sort().values
    w = torch.randn(d, generator=g).float()
    b = torch.randn(d, generator=g).float()
    a = (1 + 0.3 * torch.randn(d, generator=g)).float()
    gw = torch.randn(r, d, generator=g).float()

    x64 = x.double().requires_grad_()
    p64 = [t.double().requires_grad_() for t in (w, b, a)]
    ref = O.graph_norm(x64, batch if B > 1 else None, *p64, eps=1e-5, batch_size=B)
    if slope != 1.0:
        # an element within f32 rounding of the LeakyReLU kink may legitimately
        # take the other slope: give those no upstream gradient
        gw = gw * (ref.detach().abs() > 1e-3).float()
        ref = torch.nn.functional.leaky_relu(ref, slope)
    (ref * gw.double()).sum().backward()

    xd = x.to(dev).requires_grad_()
    pd = [t.to(dev).requires_grad_() for t in (w, b, a)]
    y = ops.graph_norm(xd, batch.to(dev) if B > 1 else None, *pd, eps=1e-5,
                       num_graphs=B, act_slope=slope)
    (y * gw.to(dev)).sum().backward()

    # graphs of 1-2 rows are degenerate (var ~ 0, rstd ~ 1/sqrt(eps) = 316): the
    # f32 cancellation c1*g - c2*o is amplified by (o^2+eps)/eps there, in the
    # reference's f32 evaluation just as here
    tol = 2e-5 if r >= 100 else 2e-3
    _close(y, ref.detach(), tol=1e-5 if r >= 100 else 1e-4)
    _close(xd.grad, x64.grad, tol=tol)
    for pg, rg in zip(pd, p64):
        _close(pg.grad, rg.grad, tol=tol)


def test_graph_norm_large_mean_is_stable(dev):
    """|mean| >> std: the one-pass f64 statistics must not cancel."""
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(50000, 32, generator=g) * 0.01 + 1000.0).float()
    ones, zeros = torch.ones(32), torch.zeros(32)
    y = ops.graph_norm(x.to(dev), None, ones.to(dev), zeros.to(dev), ones.to(dev))
    ref = O.graph_norm(x.double(), None, ones.double(), zeros.double(), ones.double())
    # the saved statistics are f32 like the reference's: at |mean| = 1000 the mean
    # itself rounds by 3e-5 = 0.3% of sigma (and the f32 inputs are spaced 6e-5)
    _close(y, ref, tol=1e-2)


def _pyg_layer_norm(x, batch, w, b, eps, mode):
    """torch_geometric 2.3.0 nn/norm/layer_norm.py restated in f64 torch ops."""
    if mode == "node":
        return torch.nn.functional.layer_norm(x, (x.shape[1],), w, b, eps)
    if batch is None:
        x = x - x.mean()
        out = x / (x.std(unbiased=False) + eps)
    else:
        B = int(batch.max()) + 1
        norm = torch.bincount(batch, minlength=B).clamp(min=1).to(x.dtype).mul(x.shape[1]).view(-1, 1)
        mean = torch.zeros(B, x.shape[1], dtype=x.dtype).index_add_(0, batch, x).sum(-1, keepdim=True) / norm
        x = x - mean[batch]
        var = torch.zeros(B, x.shape[1], dtype=x.dtype).index_add_(0, batch, x * x).sum(-1, keepdim=True) / norm
        out = x / (var + eps).sqrt()[batch]
    return out * w + b if w is not None else out


def _pyg_instance_norm(x, batch, w, b, eps):
    """torch_geometric 2.3.0 nn/norm/instance_norm.py (training statistics) in f64."""
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long)
    B = int(batch.max()) + 1
    cnt = torch.bincount(batch, minlength=B).clamp(min=1).to(x.dtype).view(-1, 1)
    mean = torch.zeros(B, x.shape[1], dtype=x.dtype).index_add_(0, batch, x) / cnt
    out = x - mean[batch]
    var = torch.zeros(B, x.shape[1], dtype=x.dtype).index_add_(0, batch, out * out) / cnt
    out = out / (var + eps).sqrt()[batch]
    return out * w + b if w is not None else out


@pytest.mark.parametrize("kind", ["layer-graph", "layer-node", "layer-nobatch", "instance", "instance-affine"])
def test_layer_and_instance_norm_match_the_pyg_formulas(kind, dev):
    """The two other index-based norms src/nn/norm.py:5 re-exports (the shims used to raise):
    forward and gradients against the f64 restatement, unsorted group index with an empty group."""
    from superpoint_transformer_amd import nn as N
    torch.manual_seed(4)
    n, c = 5000, 64
    x = torch.randn(n, c) * 2 + 0.5
    batch = torch.randint(0, 7, (n,))
    batch[batch == 3] = 4                                   # group 3 is empty
    if kind.startswith("layer"):
        mode = "node" if kind == "layer-node" else "graph"
        m = N.LayerNorm(c, mode=mode)
    else:
        m = N.InstanceNorm(c, affine=kind.endswith("affine"))
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.3 + 1.0)
    b_in = None if kind == "layer-nobatch" else batch
    xd = x.to(dev).requires_grad_()
    md = m.to(dev)
    out = md(xd, None if b_in is None else b_in.to(dev))
    gw = torch.randn(n, c)
    (out * gw.to(dev)).sum().backward()
    x64 = x.double().requires_grad_()
    w = m.weight.detach().cpu().double() if m.weight is not None else None
    b = m.bias.detach().cpu().double() if m.bias is not None else None
    if kind.startswith("layer"):
        ref = _pyg_layer_norm(x64, b_in, w, b, m.eps, m.mode)
    else:
        ref = _pyg_instance_norm(x64, b_in, w, b, m.eps)
    (ref * gw.double()).sum().backward()
    _close(out, ref.detach(), 2e-5)
    _close(xd.grad, x64.grad, 1e-4)


@pytest.mark.parametrize("norm_mode", ["segment", "node"])
def test_spt_norm_modes_node_and_segment_match_the_oracle(norm_mode, dev):
    """`norm_mode` of SPT (src/models/components/spt.py:368-379 -> Data.norm_index,
    src/data/data.py:103-130): every GraphNorm normalises per node / per (segment, cloud) group
    instead of per cloud - statistics on the segment-CSR kernels.  Against the f64 oracle fed
    the same group index."""
    import copy
    from oracle import spt_model as OM
    from superpoint_transformer_amd import hotpath
    from superpoint_transformer_amd.synthetic import make_nag
    small = make_nag("R", seed=9, device="cpu", sizes=(20000, 700, 260, 9000, 7000, 2))
    torch.manual_seed(1)
    cfg = hotpath.spt64_config(small[0]["x"].shape[1], small[1]["edge_attr"].shape[1])
    cfg["norm_mode"] = norm_mode
    model = hotpath.SPTSegmenter(**cfg)
    ref = copy.deepcopy(model).double()

    def norm_index(lv, n):
        batch = lv.get("batch")
        if norm_mode == "node":
            return torch.arange(n)
        sup = lv.get("super_index")
        if sup is None:
            sup = torch.zeros(n, dtype=torch.long)
        return sup * 2 + batch if batch is not None else sup
    ref_levels = []
    for lv in small.levels:
        d = dict(lv)
        d["batch"] = norm_index(lv, lv["pos"].shape[0])
        ref_levels.append(d)
    outs = OM.spt_forward(ref.net, ref_levels, dtype=torch.float64)
    # groups of one or two nodes divide by sqrt(var + eps) ~ sqrt(eps): the reference's own f32
    # evaluation moves by 2e-2 of the output scale in 'segment' mode - the bar follows it, as in
    # tests/test_model_gpu.py
    outs32 = OM.spt_forward(copy.deepcopy(model).net, ref_levels, dtype=torch.float32)

    class View:
        levels = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in lv.items()}
                  for lv in small.levels]
        num_clouds = 2

        def __getitem__(self, i):
            return self.levels[i]
    gm = model.to(dev)
    feats = gm.net(View())
    feats = feats if isinstance(feats, (list, tuple)) else [feats]
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    outs32 = outs32 if isinstance(outs32, (list, tuple)) else [outs32]
    for a, r, r32 in zip(feats, outs, outs32):
        scale = max(r.abs().max().item(), 1e-6)
        own = (r32.double() - r).abs().max().item() / scale
        err = (a.detach().cpu().double() - r).abs().max().item() / scale
        assert err < max(2e-3, 3 * own), f"{norm_mode}: {err:.3e} (f32 oracle: {own:.3e})"
    feats[0].square().mean().backward()                     # the composite route differentiates
    g = gm.net.down_stages[0].transformer_blocks[0].sa.qkv.weight.grad
    # ('node': every group is one row, x - mean_scale * mean = 0 at initialisation: the features
    # collapse to the norms' biases and this gradient is legitimately zero)
    assert torch.isfinite(g).all() and (norm_mode == "node" or g.abs().sum().item() > 0)
