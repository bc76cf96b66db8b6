"""Data-parallel plumbing: scenes shard across ranks, gradients meet once per
step.

The reference trains with Lightning DDP (configs/trainer/ddp.yaml): each rank
loads different clouds, one gradient all-reduce per step.  The whole SPT-64
gradient is ~0.85 MB, so on an xGMI mesh the exchange is latency-bound: instead
of DDP's bucketing / hook machinery, autograd writes plain per-parameter
gradients, ONE multi-tensor copy packs them into a flat buffer when a
collective is due, and a single ``all_reduce`` (RCCL on GPUs - backend "nccl" -,
gloo in the CPU tests) runs after backward; ``p.grad`` aliases the flat buffer
from then on.  With one rank nothing is packed or sent."""
import torch
import torch.distributed as dist

__all__ = ["FlatGradAllReduce", "shard_items", "broadcast_parameters", "max_over_ranks"]


def shard_items(num_items, rank, world):
    """Scene / tile indices owned by ``rank``: ``rank, rank+world, ...`` (the
    DistributedSampler partition without padding)."""
    return list(range(rank, num_items, world))


def broadcast_parameters(params, src=0, group=None):
    """Make every rank start from rank ``src``'s weights."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for p in params:
        dist.broadcast(p.data, src=src, group=group)


def max_over_ranks(value, device, group=None):
    """Slowest rank's time (the bench contract)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class FlatGradAllReduce:
    """One flat f32 buffer for all gradients of ``params``; ``reduce()`` averages it over the
    ranks with ONE collective.

    Autograd writes each parameter's gradient into its own tensor (``zero()`` sets ``p.grad`` to
    None, so the first gradient of a step is taken as is: no per-parameter ``add_`` launch - at
    train-batch sizes those ~200 five-microsecond launches were 1 ms of a 20 ms step);
    ``pack()`` then gathers them into the flat buffer with one multi-tensor copy and re-points
    every ``p.grad`` at its view, so the collective and the optimizer both see the flat buffer.
    In a one-rank group nothing is packed or communicated."""

    def __init__(self, params, group=None, always=False):
        """``always``: run the collective even in a one-rank group (exercises the RCCL path on a
        one-GPU box; a sum over one rank leaves the buffer unchanged)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.always = bool(always)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("flat gradient bucket expects f32 parameters")
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._packed = False

    def zero(self):
        """Use instead of ``optimizer.zero_grad()``: gradients start from None."""
        for p in self.params:
            p.grad = None
        self._packed = False

    def pack(self):
        """Gradients -> flat buffer (a parameter without a gradient contributes zeros);
        ``p.grad`` aliases the buffer afterwards.  Idempotent until the next ``zero()``."""
        if self._packed:
            return self.flat
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._packed = True
        return self.flat

    def reduce(self):
        """Average the gradients over the ranks.  Returns the flat buffer when it holds them
        (a collective ran, or ``pack()`` was called since the last ``zero()``), else ``None``:
        in a one-rank group nothing is packed, ``p.grad`` are the per-parameter tensors autograd
        wrote and ``flat`` is NOT valid - call ``pack()`` first to read the gradients flat
        (norm logging, clipping on one buffer)."""
        if self.world > 1 or (self.always and dist.is_initialized()):
            self.pack()
            # RCCL averages inside the collective (ReduceOp.AVG: no separate divide launch); gloo
            # (the CPU tests' backend) has no AVG: sum, then one divide
            backend = dist.get_backend(self.group)
            if backend == "nccl":
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.div_(self.world)
        return self.flat if self._packed else None

    def time_allreduce_us(self, reps=20):
        """Median wall time (microseconds) of the step's one collective on the flat buffer as it
        is (a measurement helper for the bench line; the buffer's contents are averaged ``reps``
        times, so call it after the timed region).  ``None`` without a process group."""
        if not dist.is_initialized():
            return None
        import time
        sync = torch.cuda.synchronize if self.flat.is_cuda else (lambda: None)
        times = []
        for _ in range(reps + 3):
            sync()
            t0 = time.perf_counter()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            sync()
            times.append((time.perf_counter() - t0) * 1e6)
        times = sorted(times[3:])
        return times[len(times) // 2]

    def check_views(self):
        """True when every ``p.grad`` still aliases the flat buffer."""
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base
                   for p in self.params)
