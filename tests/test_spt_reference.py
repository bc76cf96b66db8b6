"""Whole-backbone parity against the REFERENCE'S OWN ``SPT.forward``
(src/models/components/spt.py:760-944, run in f64 by
tests/golden/make_golden_spt.py) for the two shipped widths: SPT-64 (S3DIS /
DALES) and SPT-128 (KITTI-360: value dim 8, FFN on), and for the nano family
(nano-2 on the s3dis_nano features: no level-0 stage, the NAG starts at level 1; dims 16,
qk_dim 2, value dim 1, 16-D edge encodings).

CPU: the reference's state dict loads into the product container with
``strict=True`` (same module tree and parameter names) and the oracle
restatement oracle/spt_model.py reproduces outputs (1e-9) and all parameter
gradients (f32-rounded in the fixture: 1e-6).
GPU: the HIP model on the same inputs.  Outputs: |err| <= 2e-4 + 1e-3 |ref|
(13+ chained f32 layers).  Gradients: within max(1e-3, 3 x the deviation a
plain f32 evaluation of the oracle shows) of each tensor's largest entry -
the second clause covers arg-max flips under the max-pool, which any f32
implementation shows (see tests/test_model_gpu.py)."""
import copy

import pytest
import torch

from conftest import load_golden, t64, tl
from oracle import spt_model as OM


def _load(which):
    from superpoint_transformer_amd import hotpath
    from superpoint_transformer_amd.nn import SPT
    g = load_golden(f"spt_forward_{which}.npz")
    cfg = {"spt64": hotpath.spt64_config, "spt128": hotpath.spt128_config,
           "nano2": hotpath.nano2_config}[which]()
    net = SPT(**cfg)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p__")}
    net.load_state_dict(sd, strict=True)          # names and shapes = the reference's
    levels = []
    for i in ((1, 2) if which == "nano2" else range(3)):     # nano: the NAG starts at level 1
        lv = {}
        for k in ("pos", "x", "edge_attr"):
            if f"l{i}__{k}" in g:
                lv[k] = t64(g[f"l{i}__{k}"])
        for k in ("super_index", "batch", "node_size", "edge_index"):
            lv[k] = tl(g[f"l{i}__{k}"]) if f"l{i}__{k}" in g else None
        lv.setdefault("x", None)
        levels.append(lv)
    outs = [t64(g[f"out{i}"]) for i in range(2)]
    gws = [t64(g[f"gw{i}"]) for i in range(2)]
    grads = {k[3:]: t64(v) for k, v in g.items() if k.startswith("g__")}
    return net, levels, outs, gws, grads, int(g["num_clouds"])


@pytest.mark.parametrize("which", ["spt64", "spt128", "nano2"])
def test_oracle_model_matches_reference_spt_forward(which):
    net, levels, outs, gws, grads, _ = _load(which)
    net = net.double()
    got = OM.spt_forward(net, levels, dtype=torch.float64, keep_graph=True)
    assert len(got) == len(outs)
    for a, r in zip(got, outs):
        torch.testing.assert_close(a, r, rtol=1e-9, atol=1e-10)
    sum((a * w).sum() for a, w in zip(got, gws)).backward()
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        scale = grads[k].abs().max().clamp(min=1e-12)
        assert ((p.grad - grads[k]).abs().max() / scale).item() < 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["spt64", "spt128", "nano2"])
def test_hip_model_matches_reference_spt_forward(which, dev):
    net, levels, outs, gws, grads, clouds = _load(which)

    # plain f32 evaluation of the oracle: the precision class of the reference itself
    net32 = copy.deepcopy(net).float()
    o32 = OM.spt_forward(net32, levels, dtype=torch.float32, keep_graph=True)
    sum((a * w.float()).sum() for a, w in zip(o32, gws)).backward()
    g32 = {k: p.grad for k, p in net32.named_parameters()}

    class View:
        num_clouds = clouds

        def __init__(self, lv):
            self.levels = lv

        def __getitem__(self, i):                  # absolute level index
            return self.levels[i - (1 if which == "nano2" else 0)]

    dl = [{k: (v.to(dev).float() if torch.is_tensor(v) and v.is_floating_point()
               else v.to(dev) if torch.is_tensor(v) else v) for k, v in lv.items()}
          for lv in levels]
    gm = net.to(dev)
    got = gm(View(dl))
    for a, r in zip(got, outs):
        a = a.detach().cpu().double()
        assert ((a - r).abs() - 1e-3 * r.abs()).max().item() <= 2e-4
    sum((a * w.to(dev).float()).sum() for a, w in zip(got, gws)).backward()
    torch.cuda.synchronize()

    def rel(a, r):
        return ((a.double() - r).abs().max() / r.abs().max().clamp(min=1e-2)).item()

    below_pool = max(rel(g32[k], grads[k]) for k in grads if k.startswith("first_stage."))
    hip = {}
    for k, p in gm.named_parameters():
        assert p.grad is not None, k
        hip[k] = p.grad.detach().cpu().double()
    bar = {k: max(1e-3, 3 * (below_pool if k.startswith("first_stage.") else rel(g32[k], grads[k])))
           for k in grads}

    def failures(ref):
        return [f"{k}: hip {rel(hip[k], ref[k]):.3e} vs bar {bar[k]:.3e}" for k in ref
                if rel(hip[k], ref[k]) > bar[k]]

    bad = failures(grads)
    if bad:
        # LeakyReLU inputs within f32 rounding of zero: which side of the kink an f32 evaluation
        # lands on is not defined by the reference (one flipped element moves every gradient
        # upstream of it by ~1e-2).  Find the derivative choices AT those elements that explain the
        # HIP gradients and hold the HIP path to the same bars against THAT exact f64 gradient.
        ref2, flips, cand = _kink_resolved_gradients(net, net32, levels, gws, grads, hip)
        bad2 = failures(ref2)
        assert not bad2, (f"{len(bad)} tensors off the reference; {len(flips)} of {len(cand)} kink "
                          f"elements flipped still leaves: {bad2[:5]}")


class _KinkLeaky(torch.autograd.Function):
    """leaky_relu whose derivative follows the given side mask (identical to the plain one away
    from x == 0; at elements the caller marks, the other one-sided derivative)."""

    @staticmethod
    def forward(ctx, x, slope, pos):
        ctx.save_for_backward(pos)
        ctx.slope = slope
        return torch.nn.functional.leaky_relu(x, slope)

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        return g * torch.where(pos, 1.0, ctx.slope).to(g.dtype), None, None


def _oracle_gradients(net, levels, gws, dtype, flips=(), record=None):
    """Parameter gradients of the oracle restatement; ``flips`` = [(leaky call, flat element)] whose
    derivative is taken on the other side; ``record`` collects every LeakyReLU input."""
    calls = [0]

    def leaky(x, slope):
        i = calls[0]
        calls[0] += 1
        if record is not None:
            record.append(x.detach().clone())
        pos = x.detach() > 0
        for c, e in flips:
            if c == i:
                pos.view(-1)[e] ^= True
        return _KinkLeaky.apply(x, slope, pos)

    m = copy.deepcopy(net).cpu().to(dtype)
    OM.LEAKY = leaky
    try:
        got = OM.spt_forward(m, levels, dtype=dtype, keep_graph=True)
        sum((a * w.to(dtype)).sum() for a, w in zip(got, gws)).backward()
    finally:
        OM.LEAKY = None
    return {k: p.grad.double() for k, p in m.named_parameters()}


def _kink_candidates(h64, h32):
    cand = []
    for i, (a, b) in enumerate(zip(h64, h32)):
        tau = 16 * (a - b.double()).square().mean().sqrt() + 1e-12
        cand += [(i, int(e)) for e in (a.abs().view(-1) < tau).nonzero().flatten()]
    return cand


def _kink_resolved_gradients(net, net32, levels, gws, grads, hip, max_candidates=32):
    """Exact f64 oracle gradients with the LeakyReLU derivative taken on the side that explains
    ``hip`` at the elements whose f64 input lies within f32 arithmetic error of zero
    (|h| < 16 x the layer's rms |h64 - h32|).  The effects of sparse flips add up to first order:
    one oracle backward per candidate gives its effect, a least-squares fit against the observed
    deviation picks the flipped set, and one more backward evaluates that set exactly."""
    h64, h32 = [], []
    base = _oracle_gradients(net, levels, gws, torch.float64, record=h64)
    for k in grads:                                     # the restatement IS the fixture's reference
        assert (base[k] - grads[k]).abs().max() <= 1e-6 * grads[k].abs().max().clamp(min=1e-12), k
    _oracle_gradients(net32, levels, gws, torch.float32, record=h32)
    cand = _kink_candidates(h64, h32)
    cand = sorted(set(cand), key=cand.index)
    assert 0 < len(cand) <= max_candidates, f"{len(cand)} LeakyReLU inputs within f32 error of zero"
    keys = list(grads)
    scale = {k: grads[k].abs().max().clamp(min=1e-2) for k in keys}

    def flat(d, ref=None):
        return torch.cat([((d[k] - (ref[k] if ref else 0)) / scale[k]).flatten() for k in keys])

    cols = torch.stack([flat(_oracle_gradients(net, levels, gws, torch.float64, flips=[c]), base)
                        for c in cand], 1)
    live = cols.abs().amax(0) > 0                      # (an element under a max-pool that did not
    sol = torch.zeros(len(cand), dtype=cols.dtype)     #  select it has no effect at all)
    sol[live] = torch.linalg.lstsq(cols[:, live], flat(hip, base).unsqueeze(1),
                                   driver="gelsd").solution.flatten()
    flips = [c for c, s_ in zip(cand, sol) if s_ > 0.5]
    return _oracle_gradients(net, levels, gws, torch.float64, flips=flips), flips, cand


def test_kink_resolution_recovers_a_planted_flip():
    """The helper the GPU test falls back on: gradients evaluated with the derivative of two
    near-zero LeakyReLU inputs taken on the other side are (a) off the reference by far more than
    the parity bar and (b) traced back to exactly those two elements."""
    net, levels, outs, gws, grads, _ = _load("spt128")
    net32 = copy.deepcopy(net).float()
    h64 = []
    _oracle_gradients(net, levels, gws, torch.float64, record=h64)
    # the two inputs closest to zero in the last two LeakyReLU calls
    planted = [(i, int(h64[i].abs().view(-1).argmin())) for i in (len(h64) - 1, len(h64) - 2)]
    fake = _oracle_gradients(net, levels, gws, torch.float64, flips=planted)
    worst = max(((fake[k] - grads[k]).abs().max() / grads[k].abs().max().clamp(min=1e-2)).item()
                for k in grads)
    assert worst > 1e-3
    import unittest.mock as mock
    plain = _kink_candidates
    with mock.patch(__name__ + "._kink_candidates", lambda a, b: planted + plain(a, b)):
        ref2, flips, cand = _kink_resolved_gradients(net, net32, levels, gws, grads, fake)
    assert sorted(flips) == sorted(planted)
    for k in grads:
        assert (ref2[k] - fake[k]).abs().max() <= 1e-12 * max(1.0, fake[k].abs().max().item())
