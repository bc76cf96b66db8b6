"""Which lines of the package launch torch's own kernels in a train step (copies, fills, elementwise,
index, cat, sum): a TorchDispatchMode that records the innermost package frame of every aten op that
allocates or writes device memory, backward included (single-threaded autograd engine, so that the
mode's thread sees the backward nodes).  Counts per step; device time is in tools/op_stacks.py.
    python tools/op_sites.py [scene=T] [model=spt64]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superpoint_transformer_amd import hotpath, synthetic  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "T"
model = sys.argv[2] if len(sys.argv) > 2 else "spt64"
dev = torch.device("cuda:0")
nag = synthetic.make_nag(scene, device=dev)
step = hotpath.build(nag, dev, 1, "all", model=model)
for _ in range(3):
    step.step()
torch.cuda.synchronize()

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose",
        "aten.permute", "aten.slice", "aten.select", "aten.expand", "aten.unsqueeze", "aten.squeeze",
        "aten.as_strided", "aten.empty", "aten.reshape", "aten.unbind", "aten.split", "aten.sym_",
        "aten.is_", "aten.stride", "aten.size", "aten._local_scalar_dense", "aten.lift_fresh",
        "aten.narrow", "aten.unfold", "aten.new_empty", "aten.empty_like", "aten.resize_", "prim.")
agg = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            frame = "?"
            for fs in reversed(traceback.extract_stack()):
                if ("superpoint_transformer_amd" in fs.filename or fs.filename.endswith("bench.py")) \
                        and "tools/" not in fs.filename:
                    frame = f"{os.path.relpath(fs.filename, ROOT)}:{fs.lineno} {fs.name}"
                    break
            agg[(name, frame)] += 1
        return func(*args, **(kwargs or {}))


N = 2
torch.autograd.set_multithreading_enabled(False)
with Sites():
    for _ in range(N):
        step.step()
torch.cuda.synchronize()
print(f"scene {scene} model {model}: aten ops per step by call site")
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:6.1f} x  {name:34s} {frame}")
