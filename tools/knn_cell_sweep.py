"""knn_1 at 15 M voxelised points for several grid cell sizes (the result does not depend on
the cell size, only the speed): where does the points-per-cell heuristic of _grid_for sit?"""
import sys, time
import torch
sys.path.insert(0, ".")
from superpoint_transformer_amd import neighbors as NB
from superpoint_transformer_amd.synthetic import make_voxel_cloud
dev = torch.device("cuda:0")
pos = make_voxel_cloud(15_000_000, voxel=0.03, seed=4321, device=dev)
s0 = NB._grid_for(pos, 2.0, 46)[0]
print("heuristic cell size", round(s0, 4))
for f in (0.5, 0.7, 0.85, 1.0, 1.2, 1.5, 2.0):
    cs = s0 * f
    NB.frnn_grid_points(pos, pos, 46, 2.0, cell_size=cs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        NB.frnn_grid_points(pos, pos, 46, 2.0, cell_size=cs)
    torch.cuda.synchronize()
    print(f"cell x{f}: {cs:.4f} m  {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
