"""N > 1 path on CPU: two processes over gloo (127.0.0.1).  Covers the scene
sharding, the initial weight broadcast, the single flat-bucket gradient
all-reduce and the max-over-ranks timing used by bench.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from superpoint_transformer_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                 # different init per rank on purpose
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LeakyReLU(),
                                    torch.nn.Linear(5, 3))
        parallel.broadcast_parameters(model.parameters(), src=0)
        w0 = torch.cat([p.detach().flatten() for p in model.parameters()])
        bucket = parallel.FlatGradAllReduce(model.parameters())
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        scenes = parallel.shard_items(7, rank, world)
        local = []
        for step in range(2):
            bucket.zero()
            g = torch.Generator().manual_seed(1000 * step + scenes[step % len(scenes)])
            x = torch.randn(16, 6, generator=g)
            model(x).square().mean().backward()
            assert not bucket.check_views()           # autograd wrote plain per-parameter tensors
            local.append(bucket.pack().clone())
            assert bucket.check_views()               # packed: every p.grad aliases the flat buffer
            bucket.reduce()
            opt.step()
        w = torch.cat([p.detach().flatten() for p in model.parameters()])
        t = parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
        out[rank] = dict(w0=w0, w=w, local0=local[0], reduced_last=bucket.flat.clone(),
                         scenes=scenes, t=t)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_data_parallel_step():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["w0"], b["w0"])                       # broadcast from rank 0
    assert not torch.equal(a["local0"], b["local0"])           # different scenes per rank
    assert torch.equal(a["w"], b["w"])                         # identical after averaged steps
    assert torch.equal(a["reduced_last"], b["reduced_last"])
    assert sorted(a["scenes"] + b["scenes"]) == list(range(7))  # disjoint cover
    assert a["t"] == b["t"] == 2.0                             # slowest rank


def test_single_process_is_a_no_op():
    model = torch.nn.Linear(4, 2)
    bucket = parallel.FlatGradAllReduce(model.parameters())
    bucket.zero()
    model(torch.ones(3, 4)).sum().backward()
    assert torch.equal(bucket.reduce(), torch.zeros(10)) and not bucket.check_views()   # untouched
    before = bucket.pack().clone()
    assert torch.equal(before[:8], torch.full((8,), 3.0)) and bucket.check_views()
    assert torch.equal(bucket.reduce(), before)
    assert parallel.shard_items(5, 0, 1) == [0, 1, 2, 3, 4]
