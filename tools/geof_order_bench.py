"""point_geof_dense_kernel at 15 M shuffled voxel points: stored vs spatial visiting order."""
import torch, sys
sys.path.insert(0, ".")
from superpoint_transformer_amd import neighbors as NB
from superpoint_transformer_amd.synthetic import make_voxel_cloud
dev = torch.device("cuda:0")
pos = make_voxel_cloud(15_000_000, voxel=0.03, seed=4321, device=dev)
nb, _ = NB.knn_1(pos, 45, 2.0)
order = NB.spatial_order(pos)
for name, o in (("stored order", None), ("spatial order", order)):
    NB.geometric_features(pos, nb, k_min=1, order=o)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3):
        NB.geometric_features(pos, nb, k_min=1, order=o)
    ev[1].record(); torch.cuda.synchronize()
    print(name, round(ev[0].elapsed_time(ev[1]) / 3, 2), "ms")
