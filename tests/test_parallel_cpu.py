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
    # one rank: nothing is packed, reduce() says so instead of handing out a stale buffer
    assert bucket.reduce() is None and not bucket.check_views()
    before = bucket.pack().clone()
    assert torch.equal(before[:8], torch.full((8,), 3.0)) and bucket.check_views()
    assert torch.equal(before, torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    assert torch.equal(bucket.reduce(), before)                 # packed: the flat gradients
    assert parallel.shard_items(5, 0, 1) == [0, 1, 2, 3, 4]


def test_edge_attr_gradient_share_protocol_on_the_host():
    """ops.EdgeAttrGradShare (host-side bookkeeping of the blocks that accumulate d edge_attr into
    one buffer): the block of the LAST forward rank walks first and starts the buffer, later
    walkers accumulate, rank 0 hands it over and resets; a rank seen twice without a hand-over (a
    partial autograd.grad on a retained graph, then the full backward) starts a fresh buffer, and
    abort() drops a half-built one."""
    from superpoint_transformer_amd import ops
    like = torch.zeros(5, 3)
    sh = ops.EdgeAttrGradShare()
    assert [sh.enter(like) for _ in range(3)] == [0, 1, 2]
    with __import__("pytest").raises(ValueError):
        sh.enter(torch.zeros(5, 3))                       # another edge_attr tensor
    b2, acc = sh.acquire(2, like)
    assert acc == 0 and b2.shape == like.shape
    b1, acc = sh.acquire(1, like)
    assert acc == 1 and b1 is b2
    assert sh.release(1) is None and sh.release(2) is None
    b0, acc = sh.acquire(0, like)
    assert acc == 1 and sh.release(0) is b2 and sh.buf is None and not sh.seen
    # partial walk (only rank 2), then the full one: rank 2 again -> fresh buffer, no stale sum
    p, acc = sh.acquire(2, like)
    q, acc2 = sh.acquire(2, like)
    assert acc == 0 and acc2 == 0 and q is not p
    sh.abort()
    assert sh.buf is None and not sh.seen
