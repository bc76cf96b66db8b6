"""Where the time of knn_1 goes at 15 M points: grid estimation (host) vs device work."""
import sys, time
import torch
sys.path.insert(0, ".")
from superpoint_transformer_amd import neighbors as NB
from superpoint_transformer_amd.synthetic import make_voxel_cloud
dev = torch.device("cuda:0")
pos = make_voxel_cloud(15_000_000, voxel=0.03, seed=4321, device=dev)
NB.knn_1(pos, 45, 2.0)
torch.cuda.synchronize()
for name, fn in (("_grid_for", lambda: NB._grid_for(pos, 2.0, 46)),
                 ("knn_1", lambda: NB.knn_1(pos, 45, 2.0)),
                 ("knn_1 fixed cell", lambda: NB.frnn_grid_points(pos, pos, 46, 2.0, cell_size=CS))):
    if name == "knn_1 fixed cell":
        CS = NB._grid_for(pos, 2.0, 46)[0]
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 3 * 1e3, 2), "ms")
