"""The north-star kernel alone: ops.segment_reduce(x [N0, 128], super_index, max, arg) on scene S's own
level-0 index (spt::segmax_stream_kernel<false>) - what bench.py's `roofline` times after the step.
    python tools/segmax_bench.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import ops  # noqa: E402
from superpoint_transformer_amd.synthetic import make_nag  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--scene", default="S")
a = ap.parse_args()
dev = torch.device("cuda:0")
nag = make_nag(a.scene, seed=1234, device=dev)
si, n1 = nag[0]["super_index"], nag[1]["pos"].shape[0]
x = torch.randn(si.numel(), 128, device=dev)
for _ in range(2):
    ops.segment_reduce(x, si, n1, "max", return_arg=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    ops.segment_reduce(x, si, n1, "max", return_arg=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
b = si.numel() * (4 * 128 + 4) + n1 * (8 * 128 + 4)
print(f"segment max + arg, [{si.numel()}, 128] -> [{n1}, 128]: {ms:.3f} ms, {b / ms / 1e6:.0f} GB/s = {b / ms / 1e6 / 8000:.3f} of 8 TB/s")
