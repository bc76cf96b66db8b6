"""GPU parity of the WHOLE hot path: SPT-64 forward + loss + backward on a
demo-room-shaped NAG (and on the reference's real demo room hierarchy) against
the float64 CPU oracle of oracle/spt_model.py.

Tolerance: logits |err| <= 2e-4 + 1e-3 |ref| (13 chained GraphNorm / MLP /
attention layers in f32).  Parameter gradients: within 1e-3 of the tensor's
largest entry, OR within 3x the deviation that a plain f32 evaluation of the
same oracle shows against the f64 one.  The second clause matters only for the
point-stage parameters below the max-pool: when a segment's two best children
differ by less than f32 rounding, f32 arithmetic (the reference's dtype) routes
the gradient to the other child than f64 does - a discrete, legitimate
difference of ~5e-3 that any f32 implementation shows (measured:
tools/diag_model_parity.py)."""
import copy

import numpy as np
import pytest
import torch

from conftest import demo_nag, tl
from oracle import spt_model as OM

pytestmark = pytest.mark.gpu


def _run_case(nag_levels, num_clouds, dev, tight=None):
    from superpoint_transformer_amd import hotpath
    torch.manual_seed(3)
    model = hotpath.SPTSegmenter(**hotpath.spt64_config(
        nag_levels[0]["x"].shape[1], nag_levels[1]["edge_attr"].shape[1]))
    with torch.no_grad():                      # leave the all-ones GraphNorm init
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    n = [lv["pos"].shape[0] for lv in nag_levels]
    g = torch.Generator().manual_seed(1)
    labels = [torch.randint(0, 13, (n[i],), generator=g) for i in (1, 2)]
    loss_fn = torch.nn.CrossEntropyLoss()

    # oracle, float64
    ref_model = copy.deepcopy(model).double()
    outs = OM.spt_forward(ref_model.net, nag_levels, dtype=torch.float64, keep_graph=True)
    ref_logits = [h(x) for h, x in zip(ref_model.head, outs)]
    ref_loss = sum(l * loss_fn(lg, y) for l, lg, y in zip((1.0, 50.0), ref_logits, labels))
    ref_loss.backward()

    # the same oracle in plain f32: the precision class of the reference itself
    ref32 = copy.deepcopy(model).float()
    outs32 = OM.spt_forward(ref32.net, nag_levels, dtype=torch.float32, keep_graph=True)
    l32 = [h(x) for h, x in zip(ref32.head, outs32)]
    sum(l * loss_fn(lg, y) for l, lg, y in zip((1.0, 50.0), l32, labels)).backward()
    grads32 = dict(ref32.named_parameters())

    class View:
        def __init__(self, levels):
            self.levels, self.num_clouds = levels, num_clouds

        def __getitem__(self, i):
            return self.levels[i]

    dlev = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in lv.items()}
            for lv in nag_levels]
    gm = model.to(dev)
    logits = gm(View(dlev))
    loss = sum(l * loss_fn(lg, y.to(dev)) for l, lg, y in zip((1.0, 50.0), logits, labels))
    loss.backward()

    for a, r in zip(logits, ref_logits):
        a, r = a.detach().cpu().double(), r.detach()
        assert ((a - r).abs() - 1e-3 * r.abs()).max().item() <= 2e-4
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * abs(ref_loss.item())
    ref_grads = dict(ref_model.named_parameters())

    def rel(gr, r):
        return ((gr.double() - r).abs() / r.abs().max().clamp(min=1e-2)).max().item()

    # which point-stage tensor an arg-max flip lands on is random: bound that
    # group by the WORST f32-oracle deviation inside the group
    below_pool = max(rel(grads32[k].grad, ref_grads[k].grad)
                     for k in ref_grads if k.startswith("net.first_stage."))
    worst = {}
    for k, p in gm.named_parameters():
        r = ref_grads[k].grad
        assert p.grad is not None and r is not None, k
        err = rel(p.grad.detach().cpu(), r)
        if tight is not None:                    # tie-free case: no arg-max clause, one flat bar
            worst[k] = err
            continue
        err32 = below_pool if k.startswith("net.first_stage.") else rel(grads32[k].grad, r)
        assert err <= max(1e-3, 3 * err32), f"{k}: hip {err:.3e} vs f32-oracle {err32:.3e}"
    if tight is not None:
        assert below_pool <= tight, f"the case is not tie-free in f32: {below_pool:.3e}"
        bad = {k: f"{v:.2e}" for k, v in worst.items() if v > tight}
        assert not bad, f"worst {max(worst.values()):.3e}; above {tight}: {bad}"


def test_spt64_train_step_on_synthetic_room(dev):
    from superpoint_transformer_amd.synthetic import make_nag
    nag = make_nag("R", seed=21, device="cpu", sizes=(20000, 600, 250, 9000, 7000, 2))
    _run_case(nag.levels, 2, dev)


def _pool_margin(levels):
    """Smallest gap between the best and the second-best child of any (segment, channel) of the
    max-pools in the f64 oracle, relative to the largest feature (test infrastructure)."""
    import copy as _copy
    from oracle import spt_oracle as O
    from superpoint_transformer_amd import hotpath
    torch.manual_seed(3)
    model = hotpath.SPTSegmenter(**hotpath.spt64_config(
        levels[0]["x"].shape[1], levels[1]["edge_attr"].shape[1]))
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ref = _copy.deepcopy(model).double()
    gaps, orig = [], O.scatter

    def spy(x, index, dim, out, dim_size, reduce):
        if reduce == "max" and x.shape[1] > 3:
            xs = x.detach()
            for s in range(int(index.max()) + 1):
                rows = xs[index == s]
                if rows.shape[0] >= 2:
                    top = rows.topk(2, dim=0).values
                    gaps.append(((top[0] - top[1]) / xs.abs().max()).min().item())
        return orig(x, index, dim, out, dim_size, reduce)

    O.scatter = spy
    try:
        with torch.no_grad():
            OM.spt_forward(ref.net, levels, dtype=torch.float64)
    finally:
        O.scatter = orig
    return min(gaps)


def test_spt64_gradients_tight_on_a_tie_free_case(dev):
    """The bar of the cases above has an escape clause for arg-max flips below the pools.  Here
    the case is small enough (300 points, 40 / 12 segments; seed picked on the CPU oracle) that
    the best and second-best child of every pooled (segment, channel) are >= 1e-5 of the feature
    range apart - 30x the f32 error of the three layers below - so no implementation in f32 can
    route a gradient differently, and EVERY parameter gradient has to be within 2e-4 of its
    tensor's largest entry (no f32-oracle clause)."""
    from superpoint_transformer_amd.synthetic import make_nag
    nag = make_nag("R", seed=23, device="cpu", sizes=(300, 40, 12, 300, 80, 2))
    assert _pool_margin(nag.levels) >= 1e-5
    _run_case(nag.levels, 2, dev, tight=2e-4)


def test_spt64_train_step_on_reference_demo_room_hierarchy(dev):
    """Real hierarchy + edges of notebooks/demo_nag_v3.h5 (levels 0-2); the
    stored 7-D edge features are zero-padded to the 18-D RPE input the model
    expects (the on-the-fly edge-feature transform is a 'next' row)."""
    lv = demo_nag()
    g = torch.Generator().manual_seed(0)
    levels = []
    for i in range(3):
        d = dict(pos=torch.from_numpy(lv[i]["pos"]).float(),
                 super_index=tl(lv[i]["super_index"]) if i < 2 else None,
                 batch=None, x=None)
        if i == 0:
            feats = [torch.from_numpy(lv[0][k]).float() for k in
                     ("linearity", "planarity", "scattering", "verticality", "elevation")]
            rgb = torch.from_numpy(lv[0]["rgb"]).float() / 255
            d["x"] = torch.cat(feats + [rgb], dim=1)                     # 8 point features
        else:
            ei = tl(lv[i]["edge_index"])
            loops = torch.arange(d["pos"].shape[0])
            full = torch.cat([ei, ei.flip(0), torch.stack([loops, loops])], dim=1)
            ea7 = torch.from_numpy(lv[i]["edge_attr"]).float()
            ea = torch.zeros(full.shape[1], 18)
            ea[:ea7.shape[0], :7] = ea7
            ea[ea7.shape[0]:2 * ea7.shape[0], :7] = -ea7
            d["edge_index"], d["edge_attr"] = full, ea
            sub = lv[i]["sub_pointers"].astype(np.int64)
            d["node_size"] = torch.from_numpy(sub[1:] - sub[:-1]) if i == 1 else None
        levels.append(d)
    ns1 = levels[1]["node_size"]
    levels[2]["node_size"] = torch.zeros(levels[2]["pos"].shape[0], dtype=torch.long).index_add_(
        0, levels[1]["super_index"], ns1)
    _run_case(levels, 1, dev)
