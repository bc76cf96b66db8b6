"""Run the SPT-64 forward (+ backward) several times on one batch and report the first module
whose output is not bitwise reproducible (forward hooks on every submodule, in call order)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from superpoint_transformer_amd import csr, hotpath
from superpoint_transformer_amd.synthetic import make_nag

dev = torch.device("cuda:0")
clouds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
adopt = [bool(int(c)) for c in (sys.argv[2] if len(sys.argv) > 2 else "000")]
nag = make_nag("R", seed=11, device=dev, sizes=(30_000, 900, 380, 9_000, 7_000, clouds))
torch.manual_seed(0)
model = hotpath.SPTSegmenter(**hotpath.spt64_config(8, 18)).to(dev)
view = hotpath._NagView(nag)
runs = []
for it, ad in enumerate(adopt):
    csr.use_sub_views(ad)
    for lv in nag.levels:
        csr.forget(lv.get("super_index"), lv.get("edge_index"), lv.get("batch"))
    rec = []
    hooks = []
    for name, m in model.named_modules():
        def hook(mod, inp, out, name=name):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if torch.is_tensor(o) and o.is_floating_point():
                    rec.append((f"{name}[{i}]", o.detach().clone()))
        hooks.append(m.register_forward_hook(hook))
    model.zero_grad(set_to_none=True)
    out = model(view)
    sum(o.square().mean() for o in out).backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    grads = [(k, p.grad.clone()) for k, p in model.named_parameters() if p.grad is not None]
    runs.append((rec, grads))
base = runs[0]
for r, (rec, grads) in enumerate(runs[1:], 1):
    bad = [(n, float((a - b).abs().max())) for (n, a), (_, b) in zip(base[0], rec) if not torch.equal(a, b)]
    badg = [(n, float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))) for (n, a), (_, b) in zip(base[1], grads) if not torch.equal(a, b)]
    print(f"run {r} (adopt={adopt[r]}) vs run 0 (adopt={adopt[0]}): {len(bad)} of {len(rec)} forward outputs differ; first:", bad[:3])
    print(f"   {len(badg)} of {len(grads)} parameter gradients differ; first:", badg[:3])
