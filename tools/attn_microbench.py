"""Edge-attention fwd/bwd at scene-S level-1 shape (N=428 571, E=7.03 M), with an
optional locality knob: --local W draws targets within +-W of the source index
(real superpoint graphs are spatially local; the synthetic bench graph is not)."""
import argparse
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from superpoint_transformer_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--local", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--n", type=int, default=428571)
ap.add_argument("--e", type=int, default=7030000)
ap.add_argument("--mode", type=int, default=2, help="2 split-bf16 MFMA, 1 f32 MFMA, 0 VALU")
ap.add_argument("--packed", type=int, default=2, help="backward tiles over the edge stream (1) or per node (0)")
a = ap.parse_args()
dev = torch.device("cuda:0")
from superpoint_transformer_amd import _lib
_lib.lib.spt_attn_use_mfma(a.mode)
_lib.lib.spt_attn_bwd_packed(a.packed)  # 2 edge-lane, 1 packed, 0 per node
g = torch.Generator(device=dev).manual_seed(0)
n, e = a.n, a.e
s = torch.randint(0, n, (e,), device=dev, generator=g)
if a.local:
    t = (s + torch.randint(-a.local, a.local + 1, (e,), device=dev, generator=g)).clamp_(0, n - 1)
else:
    t = torch.randint(0, n, (e,), device=dev, generator=g)
ei = torch.stack([s, t])
qkv = torch.randn(n, 192, device=dev, generator=g).requires_grad_()
ea = (torch.randn(e, 32, device=dev, generator=g) * 0.3).requires_grad_()
W = [(torch.randn(64, 32, device=dev, generator=g).mul_(0.1).requires_grad_(),
      torch.randn(64, device=dev, generator=g).mul_(0.1).requires_grad_()) for _ in range(3)]
gw = torch.randn(n, 64, device=dev, generator=g)
for r in range(a.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ops.edge_attention(qkv, ei, ea, *W, num_heads=16, qk_dim=4, scale_a=0.5)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.backward(gw)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if r:
        print(f"fwd {1e3 * (t1 - t0):.3f} ms   bwd {1e3 * (t2 - t1):.3f} ms")
