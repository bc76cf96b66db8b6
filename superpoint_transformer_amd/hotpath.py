"""The benchmarked hot path: one step = one pass over one NAG batch.

v0 (this file grows with the kernels): the hierarchical segment-CSR scatter
chain of SPT-64 - per batch CSR build of every level's ``super_index``, max
pool child->parent forward + backward (src/nn/stage.py:429-431), IndexUnpool
parent->child forward + backward (src/nn/unpool.py:12-13).
"""
import torch

from . import ops
from .csr import build_csr


class _KernelTimer:
    """HIP events around ONE kernel launch on torch's current stream (the
    stream every launch of this library uses)."""

    def __init__(self):
        self.pairs = []

    def time(self, fn):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.pairs.append((a, b))
        return out

    def mean_ms(self):
        if not self.pairs:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs)

    def reset(self):
        self.pairs = []


class ScatterChain:
    name = "segment-CSR scatter chain (CSR build + max-pool fwd/bwd over L0->L1->L2 + unpool fwd/bwd)"

    def __init__(self, nag, dev, world=1):
        self.nag, self.dev, self.world = nag, dev, world
        n0, n1, n2 = nag.num_points
        self.n = (n0, n1, n2)
        g = torch.Generator(device=dev).manual_seed(99)
        self.x0 = torch.randn(n0, 128, device=dev, generator=g)
        self.x1 = torch.randn(n1, 64, device=dev, generator=g)
        self.g1 = torch.randn(n1, 128, device=dev, generator=g)
        self.g2 = torch.randn(n2, 64, device=dev, generator=g)
        self.timer = _KernelTimer()

    def reset_kernel_timers(self):
        self.timer.reset()

    def step(self):
        nag = self.nag
        n0, n1, n2 = self.n
        csr0 = build_csr(nag[0]["super_index"], n1)
        csr1 = build_csr(nag[1]["super_index"], n2)
        # forward
        p1, a1 = self.timer.time(lambda: ops._seg_reduce_fwd(self.x0, csr0, 3, True))
        p2, a2 = ops._seg_reduce_fwd(self.x1, csr1, 3, True)
        u1 = ops._gather_fwd(p2, csr1.idx)
        # backward
        gp2, _ = ops._seg_reduce_fwd(u1, csr1, 0, False)          # unpool bwd
        gx1 = ops._seg_reduce_bwd(self.g2, a2, csr1, 3, n1)
        gx0 = ops._seg_reduce_bwd(self.g1, a1, csr0, 3, n0)
        return gx0, gx1, gp2

    def roofline(self, peak_gbs):
        n0, n1, _ = self.n
        c = 128
        # algorithmic bytes of segment max fwd with arg, L0->L1:
        # child row 4c + 4 (perm); parent row 4c (out) + 4c (arg) + 4 (rowptr)
        bytes_ = n0 * (4 * c + 4) + n1 * (8 * c + 4)
        ms = self.timer.mean_ms()
        ach = bytes_ / (ms * 1e-3) / 1e9 if ms else None
        return {"bound": "hbm", "kernel": "segcsr_reduce_kernel<MAX,VEC4,ARG> L0->L1 C=128",
                "achieved": round(ach, 1) if ach else None, "peak": peak_gbs,
                "unit": "GB/s", "frac": round(ach / peak_gbs, 4) if ach else None,
                "traffic": None, "bytes_per_launch": bytes_,
                "ms_per_launch": round(ms, 4) if ms else None}

    def describe(self, scene, sizes):
        return f"{self.name}; scene {scene} {sizes}"


def build(nag, dev, world=1, stages="all"):
    return ScatterChain(nag, dev, world)
