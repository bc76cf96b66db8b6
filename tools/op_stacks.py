"""Which lines of the package launch torch's own kernels in a train step (copies, fills, elementwise,
index, cat): torch.profiler with stacks, grouped by (op, innermost package frame).
    python tools/op_stacks.py [scene=T] [model=spt64]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import hotpath, synthetic  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "T"
model = sys.argv[2] if len(sys.argv) > 2 else "spt64"
dev = torch.device("cuda:0")
nag = synthetic.make_nag(scene, device=dev)
step = hotpath.build(nag, dev, 1, "all", model=model)
for _ in range(3):
    step.step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step.step()
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.events():
    self_dev = getattr(ev, "self_device_time_total", 0) or 0
    if not ev.name.startswith("aten::") or self_dev <= 0:
        continue                                   # aten ops that own device time themselves
    frame = next((f for f in (ev.stack or []) if "superpoint_transformer_amd" in f or "bench.py" in f), None)
    if frame is None:
        frame = next(iter(ev.stack or ["?"]), "?")
    key = (ev.name, frame.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", ""))
    agg[key] += 1
    tim[key] += self_dev
print(f"scene {scene} model {model}: leaf aten ops with device time, per step")
for key, n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:6.1f} x  {tim[key] / N:8.1f} us  {key[0]:28s} {key[1][:150]}")
