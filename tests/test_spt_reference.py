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
    for k, p in gm.named_parameters():
        assert p.grad is not None, k
        err = rel(p.grad.detach().cpu(), grads[k])
        err32 = below_pool if k.startswith("first_stage.") else rel(g32[k], grads[k])
        assert err <= max(1e-3, 3 * err32), f"{k}: hip {err:.3e} vs f32-oracle {err32:.3e}"
