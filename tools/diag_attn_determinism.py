"""Edge attention backward twice on the same inputs: which gradient blocks are bitwise reproducible
(dq / dk|dv columns of gqkv, d edge_attr, encoder gradients), for the edge-lane backward in target
order (SPT_EL_TARGET_ORDER=1, default) or source order (=0)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from superpoint_transformer_amd import ops

dev = torch.device("cuda:0")
for n, e in ((900, 14000), (428571, 7030000)):
    g = torch.Generator(device=dev).manual_seed(0)
    s = torch.randint(0, n, (e,), device=dev, generator=g)
    t = torch.randint(0, n, (e,), device=dev, generator=g)
    ei = torch.stack([s, t])
    qkv0 = torch.randn(n, 192, device=dev, generator=g)
    ea0 = torch.randn(e, 32, device=dev, generator=g) * 0.3
    W = [(torch.randn(64, 32, device=dev, generator=g).mul_(0.1).requires_grad_(),
          torch.randn(64, device=dev, generator=g).mul_(0.1).requires_grad_()) for _ in range(3)]
    gw = torch.randn(n, 64, device=dev, generator=g)
    res = []
    for rep in range(3):
        qkv = qkv0.clone().requires_grad_()
        ea = ea0.clone().requires_grad_()
        for w, b in W:
            w.grad = b.grad = None
        out = ops.edge_attention(qkv, ei, ea, *W, num_heads=16, qk_dim=4, scale_a=0.5)
        out.backward(gw)
        torch.cuda.synchronize()
        res.append((qkv.grad[:, :64].clone(), qkv.grad[:, 64:].clone(), ea.grad.clone(),
                    torch.cat([w.grad.reshape(-1) for w, _ in W]), torch.cat([b.grad for _, b in W])))
    names = ("dq", "dk|dv", "d edge_attr", "dW", "db")
    for r in (1, 2):
        print(n, e, "run", r, {nm: (int((a != b).sum()), float((a - b).abs().max() / b.abs().max()))
                               for nm, a, b in zip(names, res[r], res[0])})
