"""GPU parity of the child -> parent pools beyond max / min / mean / sum (src/nn/pool.py:80-360):
the attentive pools against the fixture produced by the reference's own modules in float64
(tests/golden/make_golden_pool.py) - forward, input gradients and parameter gradients - and
``StdPool`` against PyG's published formula.

Stated tolerance (f32 segment kernels vs the f64 reference): |err| <= 1e-5 * scale + 1e-4 * |ref|
with scale = max |ref| of the tensor (gradients of the parameters sum hundreds of rows)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "attentive_pool.npz"))


def _close(a, ref, name):
    ref = ref.double()
    err = (a.double().cpu() - ref).abs() - 1e-4 * ref.abs()
    bound = 1e-5 * max(float(ref.abs().max()), 1.0)
    assert float(err.max()) <= bound, f"{name}: {float(err.max()):.3e} > {bound:.3e}"


def _make(tag):
    from superpoint_transformer_amd.nn import AttentivePool, AttentivePoolWithLearntQueries
    if tag == "a":
        return AttentivePool(dim=64, q_in_dim=48, num_heads=16, in_dim=40, out_dim=96, qk_dim=4,
                             in_rpe_dim=9, k_rpe=True, q_rpe=True)
    if tag == "b":
        return AttentivePool(dim=32, q_in_dim=32, num_heads=4, qk_dim=8, qk_scale="d+g",
                             in_rpe_dim=5, k_rpe=True, q_rpe=True, heads_share_rpe=True)
    return AttentivePoolWithLearntQueries(dim=32, num_heads=8, qk_dim=2, qk_scale=0.7,
                                          in_rpe_dim=6, k_rpe=True, qkv_bias=False)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_attentive_pool_matches_the_reference(tag, dev):
    pool = _make(tag)
    state = {k[len(tag) + 5:]: torch.from_numpy(G[k]).float() for k in G.files
             if k.startswith(tag + "__p__")}
    assert set(state) == set(dict(pool.named_parameters()))        # the reference's names
    pool.load_state_dict(state)
    pool = pool.to(dev)

    def t(k):
        return torch.from_numpy(G[f"{tag}__{k}"])

    xc = t("x_child").float().to(dev).requires_grad_()
    xp = t("x_parent").float().to(dev).requires_grad_()
    ea = t("edge_attr").float().to(dev).requires_grad_()
    y = pool(xc, xp, t("index").to(dev), edge_attr=ea, num_pool=xp.shape[0])
    (y * t("gw").float().to(dev)).sum().backward()
    _close(y.detach(), t("out"), "out")
    _close(xc.grad, t("g_x_child"), "g_x_child")
    _close(ea.grad, t("g_edge_attr"), "g_edge_attr")
    if f"{tag}__g_x_parent" in G.files:
        _close(xp.grad, t("g_x_parent"), "g_x_parent")
    else:
        assert xp.grad is None                                      # learnt queries
    for k, p in pool.named_parameters():
        _close(p.grad, t("g__" + k), "grad " + k)


def test_attentive_pool_of_a_large_level(dev):
    """400 k children (the Linears take the skinny-GEMM path) against the same chain in f64."""
    from superpoint_transformer_amd.nn import AttentivePool
    g = torch.Generator().manual_seed(8)
    nc, n_parent = 400_000, 30_000
    pool = AttentivePool(dim=64, q_in_dim=64, num_heads=16, qk_dim=4, in_rpe_dim=9, k_rpe=True).to(dev)
    index = torch.randint(0, n_parent, (nc,), generator=g).to(dev)
    xc = torch.randn(nc, 64, generator=g).to(dev)
    xp = torch.randn(n_parent, 64, generator=g).to(dev)
    ea = (torch.randn(nc, 9, generator=g) * 0.5).to(dev)
    y = pool(xc, xp, index, edge_attr=ea)
    with torch.no_grad():                                       # the same chain in f64, on the host
        p = {k: v.double().cpu() for k, v in pool.state_dict().items()}
        ix, xcd, ead = index.cpu(), xc.double().cpu(), ea.double().cpu()
        q = (xp.double().cpu() @ p["q.weight"].T + p["q.bias"])[ix].view(nc, 16, 4)
        kv = xcd @ p["kv.weight"].T + p["kv.bias"]
        k = kv[:, :64].view(nc, 16, 4) + (ead @ p["k_rpe.weight"].T + p["k_rpe.bias"]).view(nc, 16, 4)
        cnt = torch.bincount(ix, minlength=n_parent).double()
        q = q * (4 ** -0.5) * (cnt ** -0.5)[ix].view(-1, 1, 1)
        c = (q * k).sum(-1)
        mx = torch.full((n_parent, 16), -1e300, dtype=torch.float64)
        mx.scatter_reduce_(0, ix.view(-1, 1).expand(nc, 16), c, "amax")
        e = (c - mx[ix]).exp()
        z = torch.zeros((n_parent, 16), dtype=torch.float64).index_add_(0, ix, e)
        a = e / (z[ix] + 1e-16)
        ref = torch.zeros((n_parent, 64), dtype=torch.float64).index_add_(
            0, ix, (kv[:, 64:].view(nc, 16, 4) * a.unsqueeze(-1)).view(nc, 64))
    _close(y.detach(), ref, "out")


def test_std_pool_and_factory(dev):
    from superpoint_transformer_amd.nn import StdPool, pool_factory
    from superpoint_transformer_amd.nn.pool import BaseAttentivePool
    g = torch.Generator().manual_seed(9)
    n, m = 5000, 300
    x = torch.randn(n, 24, generator=g)
    x[:, 3] = 1.25                                                  # constant column: std reads 0
    index = torch.randint(0, m - 4, (n,), generator=g)              # the last parents are empty
    pool = pool_factory("std")
    assert isinstance(pool, StdPool)
    y = pool(x.to(dev), None, index.to(dev), num_pool=m).cpu()
    xd = x.double()
    cnt = torch.bincount(index, minlength=m).clamp(min=1).double().view(-1, 1)
    m1 = torch.zeros(m, 24, dtype=torch.float64).index_add_(0, index, xd) / cnt
    m2 = torch.zeros(m, 24, dtype=torch.float64).index_add_(0, index, xd * xd) / cnt
    ref = (m2 - m1 * m1).clamp(min=1e-5).sqrt()
    ref = ref.masked_fill(ref <= 1e-5 ** 0.5, 0.0)
    live = ref[:, 0] > 0
    torch.testing.assert_close(y[live][:, [0, 1, 2, 4]].double(), ref[live][:, [0, 1, 2, 4]],
                               rtol=1e-4, atol=1e-5)
    assert float(y[:, 3].abs().max()) == 0.0 and float(y[~live].abs().max()) == 0.0
    inst = BaseAttentivePool(dim=8)
    assert pool_factory(inst) is inst
    with pytest.raises(NotImplementedError):
        pool_factory("median")


def _attentive_spt(seg_dim=8, v_dim=3):
    """SPT-64 tree whose first down stage pools with attention: the parents' handcrafted
    features query their points, vertical edge features through an MLP as RPE."""
    from superpoint_transformer_amd import hotpath
    from superpoint_transformer_amd.nn import SPT, AttentivePool
    cfg = hotpath.spt64_config(8, 18)
    inj = 4
    cfg.update(down_in_mlp=[[inj + 128 + seg_dim, 64, 64], [inj + 64 + seg_dim, 64, 64]],
               up_in_mlp=[[inj + 64 + seg_dim + 64, 64, 64]], v_edge_mlp=[v_dim, 6, 6],
               pool=[AttentivePool(dim=128, q_in_dim=seg_dim, num_heads=16, qk_dim=4,
                                   in_rpe_dim=6, k_rpe=True, q_rpe=True), "max"])
    return SPT(**cfg)


def test_spt_with_an_attentive_first_pool_against_the_oracle(dev):
    import copy
    from oracle import spt_model as OM
    from superpoint_transformer_amd.synthetic import make_nag
    torch.manual_seed(5)
    nag = make_nag("R", seed=31, device="cpu", sizes=(6000, 300, 60, 3000, 600, 2), segment_dim=8)
    levels = nag.levels
    g = torch.Generator().manual_seed(2)
    levels[0]["v_edge_attr"] = torch.randn(levels[0]["pos"].shape[0], 3, generator=g) * 0.5
    model = _attentive_spt()
    assert model.v_edge_mlps[0] is not None and model.v_edge_mlps[1] is not None
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ref_model = copy.deepcopy(model).double()
    ref = OM.spt_forward(ref_model, levels, dtype=torch.float64, keep_graph=True)
    w = [torch.randn(r.shape, generator=g, dtype=torch.float64) for r in ref]
    sum((r * ww).sum() for r, ww in zip(ref, w)).backward()

    class View:
        num_clouds = 2

        def __init__(self, lv):
            self.lv = lv

        def __getitem__(self, i):
            return self.lv[i]

    dlev = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in lv.items()} for lv in levels]
    gm = model.to(dev)
    out = gm(View(dlev))
    sum((o * ww.float().to(dev)).sum() for o, ww in zip(out, w)).backward()
    for a, r in zip(out, ref):                     # the bar of test_model_gpu's end-to-end cases
        a, r = a.detach().cpu().double(), r.detach()
        assert float(((a - r).abs() - 1e-3 * r.abs()).max()) <= 2e-4
    ref_grads = dict(ref_model.named_parameters())
    for name, p in gm.named_parameters():
        if name.startswith(("down_stages.0.down_pool_block", "v_edge_mlps.0")):
            r = ref_grads[name].grad
            err = float((p.grad.double().cpu() - r).abs().max())
            assert err <= 2e-3 * max(float(r.abs().max()), 1e-6), (name, err)
