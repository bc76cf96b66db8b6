"""kNN + geometric features timing at several sizes / cell sizes (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import neighbors as NB
from superpoint_transformer_amd.synthetic import make_voxel_cloud

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
voxel, k, r = 0.03, 45, 2.0
pos = make_voxel_cloud(n, voxel=voxel, device=dev)
for cell in (None, 0.06, 0.09, 0.12, 0.18, 0.24):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d, i = NB.frnn_grid_points(pos, pos, k + 1, r, cell_size=cell)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    cs = NB._grid_for(pos, r, k + 1, cell)[0]
    print(f"n={n} cell={cell} (used {cs:.3f}) knn {1e3 * (t1 - t0):.1f} ms  -> {n / (t1 - t0) / 1e6:.1f} Mpts/s",
          "found/row", float((i >= 0).sum(1).float().mean()))
torch.cuda.synchronize()
t0 = time.perf_counter()
f = NB.geometric_features(pos, i[:, 1:], k_min=1)
torch.cuda.synchronize()
print(f"geof {1e3 * (time.perf_counter() - t0):.1f} ms")
