"""GPU parity: CSR build, segment reduce fwd/bwd, gather, int64 sums - HIP
kernels (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: perm / rowptr / arg / int sums / min / max values bit-exact;
float sum / mean within 1e-5 * max(1,|ref|) of the float64 oracle."""
import numpy as np
import pytest
import torch

from conftest import demo_nag, tl
from oracle import spt_oracle as O

pytestmark = pytest.mark.gpu


def _ops():
    from superpoint_transformer_amd import ops, csr
    return ops, csr


def _close(a, ref64, tol=1e-5):
    a = a.detach().cpu().double()
    err = (a - ref64).abs() / ref64.abs().clamp(min=1)
    assert err.max().item() <= tol, f"max scaled err {err.max().item():.3e}"


CASES = [
    # n, num_seg, c
    (1, 1, 1), (5, 3, 3), (1000, 37, 3), (1000, 37, 64), (777, 901, 128),
    (4096 * 3 + 17, 300, 32), (50000, 1500, 12), (20000, 9000, 64),
    (30000, 7, 132), (2500, 40, 260), (3000, 100, 2), (64, 1, 5),
]


@pytest.mark.parametrize("n,num_seg,c", CASES)
def test_csr_build_is_stable_sort(n, num_seg, c, dev):
    _, csr = _ops()
    g = torch.Generator().manual_seed(n * 31 + num_seg)
    idx = torch.randint(0, num_seg, (n,), generator=g)
    view = csr.build_csr(idx.to(dev), num_seg)
    perm, rowptr = O.csr_view(idx, num_seg)
    assert torch.equal(view.rowptr.cpu(), rowptr)
    assert torch.equal(view.perm.cpu(), perm)


def test_csr_build_edge_cases(dev):
    _, csr = _ops()
    # empty input, all-one-segment, already sorted, reverse sorted, huge num_seg
    v = csr.build_csr(torch.zeros(0, dtype=torch.long, device=dev), 5)
    assert v.rowptr.cpu().tolist() == [0] * 6
    for idx, ns in ((torch.zeros(9000, dtype=torch.long), 1),
                    (torch.arange(70000) // 7, 10000),
                    (torch.arange(70000).flip(0) // 7, 10000),
                    (torch.randint(0, 3_000_000, (5000,)), 3_000_000),
                    (torch.full((5000,), 123456), 200000)):
        v = csr.build_csr(idx.to(dev), ns)
        perm, rowptr = O.csr_view(idx, ns)
        assert torch.equal(v.rowptr.cpu(), rowptr)
        assert torch.equal(v.perm.cpu(), perm)


def test_csr_build_demo_nag_matches_reference_cluster(dev):
    """The reference's own Cluster CSR (nag[i+1].sub) is the golden vector."""
    _, csr = _ops()
    lv = demo_nag()
    for i in range(3):
        si = tl(lv[i]["super_index"])
        n_sup = lv[i + 1]["pos"].shape[0]
        v = csr.build_csr(si.to(dev), n_sup)
        ptr = lv[i + 1]["sub_pointers"].astype(np.int64)
        assert np.array_equal(v.rowptr.cpu().numpy().astype(np.int64), ptr)
        pts = lv[i + 1]["sub_points"].astype(np.int64)
        seg = np.repeat(np.arange(n_sup), ptr[1:] - ptr[:-1])
        assert np.array_equal(v.perm.cpu().numpy().astype(np.int64),
                              pts[np.lexsort((pts, seg))])


@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
@pytest.mark.parametrize("n,num_seg,c", CASES)
def test_segment_reduce_forward_backward(reduce, n, num_seg, c, dev):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(n + 7 * c)
    idx = torch.randint(0, num_seg, (n,), generator=g)
    x = torch.randn(n, c, generator=g)
    if reduce in ("min", "max"):          # force ties: few distinct values
        x = torch.randint(-3, 4, (n, c), generator=g).float()
    gw = torch.randn(num_seg, c, generator=g)

    xd = x.to(dev).requires_grad_()
    out, arg = ops.segment_reduce(xd, idx.to(dev), num_seg, reduce, return_arg=True)
    (out * gw.to(dev)).sum().backward()

    x64 = x.double().requires_grad_()
    if reduce in ("min", "max"):
        fn = O.scatter_max if reduce == "max" else O.scatter_min
        ref, rarg = fn(x64, idx, dim_size=num_seg)
        assert torch.equal(out.cpu().double(), ref.detach())       # bit-exact
        assert torch.equal(arg.cpu().long(), rarg)
    else:
        ref = O.scatter(x64, idx, 0, None, num_seg, reduce)
        _close(out, ref.detach())
        assert arg is None
    (ref * gw.double()).sum().backward()
    _close(xd.grad, x64.grad)
    if reduce in ("sum", "min", "max"):    # pure routing: exact
        assert torch.equal(xd.grad.cpu().double(), x64.grad)


def test_segment_reduce_empty_segments_and_dim_size(dev):
    ops, _ = _ops()
    x = torch.tensor([[1., -2.], [3., 4.]], device=dev)
    idx = torch.tensor([4, 4], device=dev)
    for r, exp in (("sum", [4., 2.]), ("mean", [2., 1.]), ("min", [1., -2.]), ("max", [3., 4.])):
        out = ops.segment_reduce(x, idx, 7, r)
        assert out.shape == (7, 2)
        assert out[4].cpu().tolist() == exp
        assert out[[0, 1, 2, 3, 5, 6]].abs().sum().item() == 0
    _, arg = ops.segment_reduce(x, idx, 7, "max", return_arg=True)
    assert arg[0].cpu().tolist() == [2, 2] and arg[4].cpu().tolist() == [1, 1]


@pytest.mark.parametrize("n,num_src,c", [(1000, 37, 64), (5000, 9000, 128),
                                         (333, 10, 3), (4000, 50, 1), (10, 4, 260)])
def test_gather_rows_and_backward(n, num_src, c, dev):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(n)
    idx = torch.randint(0, num_src, (n,), generator=g)
    x = torch.randn(num_src, c, generator=g)
    gw = torch.randn(n, c, generator=g)
    xd = x.to(dev).requires_grad_()
    out = ops.gather_rows(xd, idx.to(dev))
    assert torch.equal(out.cpu(), x[idx])
    (out * gw.to(dev)).sum().backward()
    ref = O.scatter_sum(gw.double(), idx, dim_size=num_src)
    _close(xd.grad, ref)


def test_int64_segment_sum_chain_is_bit_exact_on_demo_nag(dev):
    ops, _ = _ops()
    lv = demo_nag()
    sis = [tl(lv[i]["super_index"]) for i in range(3)]
    ref = O.get_sub_size(sis)
    sizes = None
    for i, si in enumerate(sis):
        n_sup = lv[i + 1]["pos"].shape[0]
        src = torch.ones(si.numel(), dtype=torch.long, device=dev) if sizes is None else sizes
        sizes = ops.segment_sum_i64(src, si.to(dev), n_sup)
        assert torch.equal(sizes.cpu(), ref[i])
    big = torch.full((1000,), 2 ** 40, dtype=torch.long, device=dev)
    out = ops.segment_sum_i64(big, torch.zeros(1000, dtype=torch.long, device=dev), 1)
    assert out.item() == 1000 * 2 ** 40


def test_large_scale_properties(dev):
    """S3DIS-scale sizes: properties instead of an oracle pass."""
    ops, csr = _ops()
    n, ns, c = 3_000_000, 85_000, 128
    g = torch.Generator(device=dev).manual_seed(5)
    idx = torch.randint(0, ns, (n,), generator=g, device=dev)
    v = csr.build_csr(idx, ns)
    sorted_keys = idx[v.perm.long()]
    assert bool((sorted_keys[1:] >= sorted_keys[:-1]).all())          # sortedness
    same = sorted_keys[1:] == sorted_keys[:-1]
    assert bool((v.perm[1:][same] > v.perm[:-1][same]).all())          # stability
    assert torch.equal(v.perm.long().sort().values, torch.arange(n, device=dev))
    assert torch.equal(v.counts().long(), torch.bincount(idx, minlength=ns))
    x = torch.randn(n, c, device=dev, generator=g)
    s = ops.segment_reduce(x, idx, ns, "sum")
    tot = x.double().sum(0)
    assert ((s.double().sum(0) - tot).abs() / tot.abs().clamp(min=1)).max() < 1e-5
    mx, arg = ops.segment_reduce(x, idx, ns, "max", return_arg=True)
    nonempty = v.counts() > 0
    rows = arg[nonempty].long()
    assert torch.equal(x.gather(0, rows), mx[nonempty])               # arg is a witness
    assert torch.equal(idx[rows[:, 0]], torch.nonzero(nonempty)[:, 0])
    assert bool((mx[idx] >= x).all())                                  # upper bound


def _segments(kind, n, gen):
    if kind == "lognormal":                       # the level-0 -> level-1 shape: mean 35, wide
        nseg = n // 35
        w = torch.exp(torch.randn(nseg, generator=gen))
        idx = torch.multinomial(w / w.sum(), n, replacement=True, generator=gen)
        return idx, nseg + 9                      # + trailing empty segments
    if kind == "empties":                         # far more segments than rows can fill
        nseg = n // 2
        return torch.randint(0, nseg, (n,), generator=gen) // 3 * 3, nseg
    if kind == "giant":                           # one segment with a third of the rows
        nseg = 3000
        idx = torch.randint(1, nseg, (n,), generator=gen)
        idx[torch.rand(n, generator=gen) < 0.34] = 1500
        return idx, nseg
    nseg = n // 200                               # "sorted": rows already in CSR order
    return torch.sort(torch.randint(0, nseg, (n,), generator=gen)).values, nseg


@pytest.mark.parametrize("kind", ["lognormal", "empties", "giant", "sorted"])
@pytest.mark.parametrize("n", [65_536, 300_017])
def test_row_streaming_max_equals_lane_group_kernel_and_oracle(kind, n, dev):
    """128-channel max + arg over >= 65 536 rows runs the row-streaming kernel (a wave owns a range
    of CSR positions cut at segment boundaries): values AND arg rows bit-identical to the
    lane-group-per-segment kernel and to the oracle, with forced ties (few distinct values),
    empty segments in front of / between / behind the others, a segment far longer than a wave's
    range, and rows that are already sorted."""
    from superpoint_transformer_amd import _lib
    ops, csr = _ops()
    gen = torch.Generator().manual_seed(n + len(kind))
    idx, nseg = _segments(kind, n, gen)
    x = torch.randint(-4, 5, (n, 128), generator=gen).float()
    x += (torch.rand(n, 128, generator=gen) < 0.3).float() * torch.randn(n, 128, generator=gen)
    res = []
    for on in (1, 0):
        prev = _lib.lib.spt_segcsr_use_stream(on)
        try:
            csr.forget(idx)
            res.append(ops.segment_reduce(x.to(dev), idx.to(dev), nseg, "max", return_arg=True))
        finally:
            _lib.lib.spt_segcsr_use_stream(prev)
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    ref, rarg = O.scatter_max(x.double(), idx, dim_size=nseg)
    assert torch.equal(res[0][0].cpu().double(), ref)
    assert torch.equal(res[0][1].cpu().long(), rarg)


@pytest.mark.parametrize("rows,B", [(150_000, 1), (200_003, 3)])
def test_row_streaming_max_with_the_fused_norm_equals_lane_group_kernel(rows, B, dev, materialised_pool_route):
    """The pool that evaluates the point MLP's last GraphNorm + LeakyReLU on the fly
    (spt_segcsr_max_affine_f32), row-streaming vs lane-group kernel: pooled values, arg rows and
    therefore every gradient bit-identical; several graphs (the coefficient rows change with the
    segment's graph)."""
    from superpoint_transformer_amd import _lib, nn as N
    g = torch.Generator().manual_seed(rows)
    mlp = N.MLP([12, 32, 64, 128], norm=N.GraphNorm).to(dev)
    mlp.FUSE_MIN_ROWS = 0
    x = (torch.randn(rows, 12, generator=g) * 2 + 0.5).to(dev)
    batch = (torch.arange(rows) * B // rows).to(dev) if B > 1 else None
    nseg = rows // 35 // B
    seg = torch.randint(0, nseg, (rows,), generator=g).to(dev)
    if B > 1:
        seg = batch * nseg + seg
    seg_graph = None if B == 1 else (torch.arange(nseg * B, device=dev) // nseg)
    gw = torch.randn(nseg * B, 128, generator=g).to(dev)
    res = []
    for on in (1, 0):
        prev = _lib.lib.spt_segcsr_use_stream(on)
        try:
            xd = x.clone().requires_grad_()
            mlp.zero_grad()
            y = mlp.forward_max_pooled(xd, seg, nseg * B, batch=batch, batch_size=B,
                                       seg_graph=seg_graph)
            (y * gw).sum().backward()
            res.append([y.detach().clone(), xd.grad.clone()] +
                       [p.grad.clone() for p in mlp.parameters()])
        finally:
            _lib.lib.spt_segcsr_use_stream(prev)
    for a, b in zip(*res):
        assert torch.equal(a, b)
