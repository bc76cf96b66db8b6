"""Time the pooled / dense backward of one fused MLP layer at scene-S size through the C ABI
(spt_fused_linear_bwd_pooled_ex_f32 / spt_fused_linear_bwd_ex_f32), DMA-staged vs register-staged.
    python tools/fmlp_bwd_bench.py [--rows 15000000] [--segs 428571] [--K 64] [--N 128] [--order shuffled|runs|sorted]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import _lib, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=15_000_000)
ap.add_argument("--segs", type=int, default=428_571)
ap.add_argument("--K", type=int, default=64)
ap.add_argument("--N", type=int, default=128)
ap.add_argument("--order", default="shuffled")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dense", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
rows, S, K, N = a.rows, a.segs, a.K, a.N
sizes = synthetic._segment_sizes(g, rows, S, "lognormal", dev)
si = torch.repeat_interleave(torch.arange(S, device=dev), sizes)
if a.order == "shuffled":
    si = si[torch.randperm(rows, generator=g, device=dev)]
elif a.order == "runs":                       # runs of ~3 rows of a segment, runs shuffled (the demo NAG's layout)
    run = torch.arange(rows, device=dev) // 3
    key = torch.rand(int(run.max()) + 1, generator=g, device=dev)[run]
    si = si[torch.argsort(key, stable=True)]
perm = torch.argsort(si, stable=True).int()
pos_seg = si[perm.long()].int()
rowptr = torch.zeros(S + 1, dtype=torch.long, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(si, minlength=S), 0)
off = (torch.rand(S, N, generator=g, device=dev) * (rowptr[1:] - rowptr[:-1]).view(-1, 1)).long()
arg = perm[(rowptr[:-1].view(-1, 1) + off).clamp(max=rows - 1)].int().contiguous()
gout = torch.randn(S, N, device=dev, generator=g)
h = torch.randn(rows, N, device=dev, generator=g)
x = torch.randn(rows, K, device=dev, generator=g)
W = torch.randn(N, K, device=dev, generator=g) * 0.1
tabN = [torch.rand(N, device=dev, generator=g) + 0.5 for _ in range(6)]
tabK = [torch.rand(K, device=dev, generator=g) + 0.5 for _ in range(3)]
gx = torch.empty(rows, K, device=dev)
gW = torch.empty(N, K, device=dev)
prev = torch.empty(2 * K + 1, dtype=torch.float64, device=dev)
ws = torch.empty(_lib.lib.spt_fused_linear_workspace_bytes(K, N), dtype=torch.uint8, device=dev)
gy = torch.randn(rows, N, device=dev, generator=g) if a.dense else None
P = _lib.ptr


def run(mode):
    if a.dense:
        st = _lib.lib.spt_fused_linear_bwd_ex_f32(
            P(gy), P(h), 0, rows, N, P(tabN[0]), P(tabN[1]), P(tabN[2]), 0.01, P(tabN[3]), P(tabN[4]),
            P(tabN[5]), P(x), K, P(tabK[0]), P(tabK[1]), P(tabK[2]), 0.01, P(W), P(gx), P(gW), 0, P(prev),
            mode, P(ws), ws.numel(), _lib.stream_ptr(dev))
    else:
        st = _lib.lib.spt_fused_linear_bwd_pooled_ex_f32(
            P(gout), P(arg), P(perm), P(pos_seg), P(h), 0, rows, N, P(tabN[0]), P(tabN[1]), P(tabN[2]),
            0.01, P(tabN[3]), P(tabN[4]), P(tabN[5]), P(x), K, P(tabK[0]), P(tabK[1]), P(tabK[2]), 0.01,
            P(W), P(gx), P(gW), 0, P(prev), mode, P(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(st, "bwd")


res = {}
for name, mode in (("dma", 1), ("register-staged", 1 | 4)):
    run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run(mode)
    e1.record()
    torch.cuda.synchronize()
    res[name] = (gx.clone(), gW.clone(), prev.clone())
    print(f"{name:16s} {e0.elapsed_time(e1) / a.reps:7.3f} ms  ({'dense' if a.dense else 'pooled'} {K}->{N}, "
          f"{rows} rows, {a.order})")
d = [(u - v).abs().max().item() / max(v.abs().max().item(), 1e-30) for u, v in zip(res["dma"], res["register-staged"])]
print("dma vs register-staged: max rel diff gx %.2e gW %.2e stats %.2e" % tuple(d))
