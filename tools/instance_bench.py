"""Per-batch panoptic target construction (cfg #5): OnTheFlyInstanceGraph in the
configs' mode ('radius-atomic', k_max 30, radius 0.1: configs/datamodule/semantic/default.yaml:29-30,
365-371) on a synthetic scene, with the share of its three parts.

    python tools/instance_bench.py [T|S|D ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd.data import NAG, Data                      # noqa: E402
from superpoint_transformer_amd.instance import InstanceData              # noqa: E402
from superpoint_transformer_amd.neighbors import cluster_radius_nn_graph  # noqa: E402
from superpoint_transformer_amd.synthetic import make_nag                 # noqa: E402
from superpoint_transformer_amd.transforms import OnTheFlyInstanceGraph   # noqa: E402

NUM_CLASSES = 13


def timed(fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts[1:]), out


def main():
    dev = torch.device("cuda:0")
    for scene in (sys.argv[1:] or ["T"]):
        syn = make_nag(scene, seed=7, device=dev)
        l0, l1 = syn[0], syn[1]
        n1 = l1["pos"].shape[0]
        g = torch.Generator(device=dev).manual_seed(1)
        # objects = the level-2 segments; every level-1 segment lies mostly in its parent object
        # and shares the rest of its points with another one; one class in ten is void
        size = l1["node_size"]
        main_part = (size * 4 + 4) // 5
        other = torch.randint(0, int(l1["super_index"].max()) + 1, (n1,), device=dev, generator=g)
        seg = torch.arange(n1, device=dev)
        cl = torch.cat([seg, seg])
        ob = torch.cat([l1["super_index"], other])
        cnt = torch.cat([main_part, (size - main_part).clamp(min=1)])
        y = ob % (NUM_CLASSES + 1)
        obj = InstanceData(cl, ob, cnt, y, dense=True)
        nag = NAG([Data(pos=l0["pos"], super_index=l0["super_index"], batch=l0["batch"]),
                   Data(pos=l1["pos"], batch=l1["batch"], obj=obj)])
        t = OnTheFlyInstanceGraph(level=1, num_classes=NUM_CLASSES, adjacency_mode="radius-atomic",
                                  k_max=30, radius=0.1)
        total, out = timed(lambda: t(nag))
        e = out[1].obj_edge_index
        t_graph, (ei, _) = timed(lambda: cluster_radius_nn_graph(
            l0["pos"], l0["super_index"], k_max=30, gap=0.1, batch=l1["batch"]))
        t_aff, _ = timed(lambda: obj.instance_graph(ei, num_classes=NUM_CLASSES))
        t_cen, _ = timed(lambda: (nag[1].estimate_instance_centroid("iou"),
                                  obj.major(NUM_CLASSES)))
        print(f"scene {scene}: {l0['pos'].shape[0]} points, {n1} segments, {obj.num_overlaps} overlaps, "
              f"{e.shape[1]} instance-graph edges: OnTheFlyInstanceGraph {total:.2f} ms "
              f"(cluster_radius_nn_graph {t_graph:.2f}, instance_graph {t_aff:.2f}, "
              f"centroids + major {t_cen:.2f})", flush=True)


if __name__ == "__main__":
    main()
