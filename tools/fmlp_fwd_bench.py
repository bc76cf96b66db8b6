"""Time the forward of one fused MLP layer at scene-S size through the C ABI
(spt_fused_linear_fwd_ex_f32).   python tools/fmlp_fwd_bench.py [--rows N] [--K 64] [--N 128] [--mode 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=15_000_000)
ap.add_argument("--K", type=int, default=64)
ap.add_argument("--N", type=int, default=128)
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
rows, K, N = a.rows, a.K, a.N
x = torch.randn(rows, K, device=dev, generator=g)
W = torch.randn(N, K, device=dev, generator=g) * 0.1
tabK = [torch.rand(K, device=dev, generator=g) + 0.5 for _ in range(3)]
h = torch.empty(rows, N, device=dev)
tot = torch.empty(2 * N + 1, dtype=torch.float64, device=dev)
ws = torch.empty(_lib.lib.spt_fused_linear_workspace_bytes(K, N), dtype=torch.uint8, device=dev)
P = _lib.ptr


def run():
    st = _lib.lib.spt_fused_linear_fwd_ex_f32(P(x), 0, rows, K, P(W), N, P(tabK[0]), P(tabK[1]), P(tabK[2]),
                                              0.01, P(h), P(tot), a.mode, P(ws), ws.numel(),
                                              _lib.stream_ptr(dev))
    _lib.check(st, "fwd")


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print(f"fwd {K}->{N} mode {a.mode}: {ms:7.3f} ms  ({rows * 4 * (K + N) / ms / 1e6:7.1f} GB/s)  "
      f"checksum {h.double().sum().item():.6e} {tot[:N].sum().item():.6e}")
