"""torch.profiler view of one train step: which aten ops own the non-spt kernels (scene S)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import hotpath, synthetic  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "S"
dev = torch.device("cuda:0")
nag = synthetic.make_nag(scene, device=dev)
model = sys.argv[2] if len(sys.argv) > 2 else "spt64"
step = hotpath.build(nag, dev, 1, "all", model=model)
for _ in range(3):
    step.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step.step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=160,
                                                         max_name_column_width=42, max_shapes_column_width=60))
