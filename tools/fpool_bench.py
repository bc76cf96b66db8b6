"""Time the pool-fused top layer (csrc/fused_pool.hip) at scene-S size through the C ABI:
spt_fused_linear_fwd_pool_runs_f32 / spt_fused_linear_bwd_pool_runs_f32.
    python tools/fpool_bench.py [--rows 15000000] [--segs 428571] [--K 64] [--N 128] [--mode 1] [--what fwd|bwd|both]"""
import argparse
import ctypes
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
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--order", default="shuffled")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--what", default="both")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
rows, S, K, N = a.rows, a.segs, a.K, a.N
sizes = synthetic._segment_sizes(g, rows, S, "lognormal", dev)
si = torch.repeat_interleave(torch.arange(S, device=dev), sizes)
if a.order == "shuffled":
    si = si[torch.randperm(rows, generator=g, device=dev)]
perm = torch.argsort(si, stable=True).int()
pos_seg = si[perm.long()].int()
rowptr = torch.zeros(S + 1, dtype=torch.int32, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(si, minlength=S), 0).int()
x = torch.randn(rows, K, device=dev, generator=g)
W = torch.randn(N, K, device=dev, generator=g) * 0.1
gnw = torch.randn(N, device=dev, generator=g)
gnb, gms = torch.randn(N, device=dev, generator=g) * 0.1, torch.rand(N, device=dev, generator=g)
pam, psc = torch.randn(1, K, device=dev, generator=g) * 0.1, torch.rand(1, K, device=dev, generator=g) + 0.5
pbs = torch.randn(K, device=dev, generator=g) * 0.1
out, raw = torch.empty(S, N, device=dev), torch.empty(S, N, device=dev)
arg, argpos = (torch.empty(S, N, dtype=torch.int32, device=dev) for _ in range(2))
glen = int(_lib.lib.spt_fused_linear_pool_gram_len(K))
gram = torch.empty(1, glen, dtype=torch.float64, device=dev)
mean, rstd, am, sc = (torch.empty(1, N, device=dev) for _ in range(4))
ws = torch.empty(_lib.lib.spt_fused_linear_pool_workspace_bytes(K, N), dtype=torch.uint8, device=dev)
r0, r1, g0 = (ctypes.c_int64 * 1)(0), (ctypes.c_int64 * 1)(rows), (ctypes.c_int32 * 1)(0)
gout = torch.randn(S, N, device=dev, generator=g)
c1, c2, c3 = (torch.rand(1, N, device=dev, generator=g) * 0.1 for _ in range(3))
gm = torch.empty(S, N, device=dev)
gx = torch.empty(rows, K, device=dev)
gW = torch.empty(N, K, device=dev)
ptot = torch.empty(1, 2 * K + 1, dtype=torch.float64, device=dev)
P = _lib.ptr
sp = _lib.stream_ptr(dev)


def fwd():
    st = _lib.lib.spt_fused_linear_fwd_pool_runs_f32(
        P(x), P(perm), P(pos_seg), P(rowptr), None, S, rows, 1, r0, r1, g0, 1, K, P(W), N, P(gnw), P(gnb),
        P(gms), 1e-5, 0.01, P(pam), P(psc), P(pbs), 0.01, P(out), P(arg), P(argpos), P(raw), P(gram), None,
        P(mean), P(rstd), P(am), P(sc), a.mode, P(ws), ws.numel(), sp)
    _lib.check(st, "fwd_pool")


def bwd():
    st = _lib.lib.spt_fused_linear_bwd_pool_runs_f32(
        P(gout), P(raw), P(argpos), P(perm), P(pos_seg), None, S, 1, r0, r1, g0, 1, N, P(am), P(sc), P(gnb),
        0.01, P(c1), P(c2), P(c3), P(x), K, P(pam), P(psc), P(pbs), 0.01, P(W), P(gram), P(gm), P(gx),
        P(gW), P(ptot), a.mode, P(ws), ws.numel(), sp)
    _lib.check(st, "bwd_pool")


fwd()
torch.cuda.synchronize()
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    if a.what not in (name, "both"):
        continue
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name} {e0.elapsed_time(e1) / a.reps:7.3f} ms per call (all kernels of the call; {K}->{N}, {rows} rows, mode {a.mode})")
