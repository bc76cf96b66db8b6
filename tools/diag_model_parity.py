"""Per-parameter gradient error of the SPT-64 train step vs the f64 oracle
(diagnostic; run on the GPU box)."""
import copy
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import spt_model as OM
from superpoint_transformer_amd import hotpath
from superpoint_transformer_amd.synthetic import make_nag

dev = torch.device("cuda:0")
for clouds in (1, 2):
    nag = make_nag("R", seed=21, device="cpu", sizes=(20000, 600, 250, 9000, 7000, clouds))
    if clouds == 1:
        for lv in nag.levels:
            lv["batch"] = None
    torch.manual_seed(3)
    model = hotpath.SPTSegmenter(**hotpath.spt64_config())
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    n = nag.num_points
    g = torch.Generator().manual_seed(1)
    labels = [torch.randint(0, 13, (n[i],), generator=g) for i in (1, 2)]
    lf = torch.nn.CrossEntropyLoss()
    ref = copy.deepcopy(model).double()
    outs = OM.spt_forward(ref.net, nag.levels, dtype=torch.float64, keep_graph=True)
    rl = [h(x) for h, x in zip(ref.head, outs)]
    sum(l * lf(a, y) for l, a, y in zip((1.0, 50.0), rl, labels)).backward()

    # f32 CPU oracle too: how much of the error is plain f32 arithmetic?
    ref32 = copy.deepcopy(model).float()
    outs32 = OM.spt_forward(ref32.net, nag.levels, dtype=torch.float32, keep_graph=True)
    rl32 = [h(x) for h, x in zip(ref32.head, outs32)]
    sum(l * lf(a, y) for l, a, y in zip((1.0, 50.0), rl32, labels)).backward()

    class V:
        levels = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in lv.items()}
                  for lv in nag.levels]
        num_clouds = clouds

        def __getitem__(self, i):
            return self.levels[i]

    gm = model.to(dev)
    lg = gm(V())
    sum(l * lf(a, y.to(dev)) for l, a, y in zip((1.0, 50.0), lg, labels)).backward()
    print(f"== clouds={clouds} logits err",
          [float((a.detach().cpu().double() - r.detach()).abs().max()) for a, r in zip(lg, rl)],
          "f32-cpu logits err",
          [float((a.detach().double() - r.detach()).abs().max()) for a, r in zip(rl32, rl)])
    rg = dict(ref.named_parameters())
    rg32 = dict(ref32.named_parameters())
    rows = []
    for k, p in gm.named_parameters():
        r = rg[k].grad
        sc = r.abs().max().clamp(min=1e-2)
        e = float(((p.grad.cpu().double() - r).abs() / sc).max())
        e32 = float(((rg32[k].grad.double() - r).abs() / sc).max())
        rows.append((e, e32, k))
    rows.sort(reverse=True)
    for e, e32, k in rows[:12]:
        print(f"  hip {e:.2e}   cpu-f32 {e32:.2e}   {k}")
