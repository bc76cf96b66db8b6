"""Preprocessing leg, the two entries against the one-call entry (GPU box):
    python tools/pre_fused_bench.py [S|D] [n_points] [reps]
prints one line per scene setting: knn_1, geometric_features, knn_1_features (ms, Mpoints/s)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PRE_CFG, PRE_GEOM
from superpoint_transformer_amd import neighbors as NB
from superpoint_transformer_amd.synthetic import make_voxel_cloud

scene = sys.argv[1] if len(sys.argv) > 1 else "S"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (15_000_000 if scene == "S" else 12_000_000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
voxel, k, r = PRE_CFG[scene]
pos = make_voxel_cloud(n, voxel=voxel, seed=4321, device=dev, **PRE_GEOM.get(scene, {}))
n = pos.shape[0]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del out
    return sorted(ts)[len(ts) // 2] * 1e3


t_knn = timed(lambda: NB.knn_1(pos, k, r))
nb, _ = NB.knn_1(pos, k, r)
t_geof = timed(lambda: NB.geometric_features(pos, nb, k_min=1))
del nb
t_fused = timed(lambda: NB.knn_1_features(pos, k, r, k_min=1))
print(f"scene {scene} n={n} k={k} r={r}: knn_1 {t_knn:.2f} ms  geof {t_geof:.2f} ms  -> "
      f"{n / (t_knn + t_geof) / 1e3:.1f} Mpts/s | knn_1_features {t_fused:.2f} ms -> {n / t_fused / 1e3:.1f} Mpts/s")
