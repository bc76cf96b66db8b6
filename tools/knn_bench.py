"""Preprocess leg alone (knn_1 + geometric_features) for a BASELINE config:
    python tools/knn_bench.py S|D [n_points] [reps]
S = S3DIS (3 cm voxels, k=45, r=2 m), D = DALES (10 cm voxels, k=25, r=10 m)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import neighbors as NB
from superpoint_transformer_amd.synthetic import SCENES, make_voxel_cloud

CFG = {"S": (0.03, 45, 2.0), "D": (0.10, 25, 10.0), "V": (0.10, 25, 10.0)}   # V: dense volume
GEOM = {"D": dict(patch=20.0, extent=(350.0, 350.0, 30.0))}
scene = sys.argv[1] if len(sys.argv) > 1 else "S"
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else SCENES["D" if scene == "V" else scene][0]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
voxel, k, r = CFG[scene]
dev = torch.device("cuda:0")
pos = make_voxel_cloud(n, voxel=voxel, seed=4321, device=dev, **GEOM.get(scene, {}))
for rep in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb, d = NB.knn_1(pos, k, r)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    f = NB.geometric_features(pos, nb, k_min=1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rep:
        print(f"scene {scene} n={pos.shape[0]} k={k} r={r}: knn_1 {1e3 * (t1 - t0):.2f} ms  geof "
              f"{1e3 * (t2 - t1):.2f} ms  -> {pos.shape[0] / (t2 - t0) / 1e6:.1f} Mpts/s  "
              f"found/row {float((nb >= 0).sum(1).float().mean()):.2f}")
