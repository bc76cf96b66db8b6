"""Benchmark of the SPT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scene S|T|D|R]

One "step" = one pass of the hot path over one synthetic
NAG batch already resident in HBM.  N > 1: one process per GPU (torchrun env),
every rank owns a different scene of the same shape (scenes shard data
parallel, "weak" scaling), the only collective is the gradient all-reduce of
the model parameters (RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scene", default="S")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-scale", type=float, default=None)
    p.add_argument("--stages", default="all")
    return p.parse_args()


def cpu_baseline(scene, scale):
    """The reference's CPU-tensor path for the same step (oracle = restatement
    of torch_scatter's semantics, pinned against the reference's own modules)
    on a bounded sample of the workload, timed on this box's host cores."""
    from oracle import spt_oracle as O
    from superpoint_transformer_amd.synthetic import SCENES, make_nag
    n0_full = SCENES[scene][0]
    if scale is None:
        scale = min(1.0, 600_000 / n0_full)
    nag = make_nag(scene, seed=1234, device="cpu", scale=scale)
    n0, n1, n2 = nag.num_points
    g = torch.Generator().manual_seed(99)
    x0 = torch.randn(n0, 128, generator=g).requires_grad_()
    x1 = torch.randn(n1, 64, generator=g).requires_grad_()
    g1 = torch.randn(n1, 128, generator=g)
    g2 = torch.randn(n2, 64, generator=g)
    si0, si1 = nag[0]["super_index"], nag[1]["super_index"]

    def step():
        p1, _ = O.scatter_max(x0, si0, dim_size=n1)
        p2, _ = O.scatter_max(x1, si1, dim_size=n2)
        u1 = O.index_unpool(p2, si1)
        O.scatter_sum(u1.detach(), si1, dim_size=n2)
        torch.autograd.backward([p1, p2], [g1, g2])
        x0.grad = None
        x1.grad = None

    step()
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        step()
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(n0 / dt / 1e6, 4), "unit": "Mpoints/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"scene {scene} scaled x{scale:.4g}: N=({n0},{n1},{n2}), "
                      f"{reps} reps, same step on torch-CPU via oracle/spt_oracle.py"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path exists)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from superpoint_transformer_amd import hotpath
    from superpoint_transformer_amd.synthetic import SCENES, make_nag

    nag = make_nag(args.scene, seed=1234 + rank, device=dev)
    path = hotpath.build(nag, dev, world=world, stages=args.stages)

    for _ in range(args.warmup):
        path.step()
    path.reset_kernel_timers()

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        path.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n0 = nag.num_points[0]
    value = world * n0 * args.steps / dt / 1e6
    roof = path.roofline(HBM_PEAK_GBS)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.scene, args.cpu_scale)

    if rank == 0:
        line = {
            "metric": "Mpoints/s, SPT hot path fwd+bwd",
            "value": round(value, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": path.describe(args.scene, SCENES.get(args.scene)),
                "scene": args.scene,
                "points_per_gpu": n0,
                "parallelism": f"dp{world}",
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
