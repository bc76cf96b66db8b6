"""GPU parity: the fused [Linear -> GraphNorm -> LeakyReLU] x L path against the
float64 oracle (src/nn/mlp.py:85-94 + PyG GraphNorm restated) and against the
layer-by-layer HIP path it replaces."""
import copy

import pytest
import torch

from oracle import spt_model as OM

pytestmark = pytest.mark.gpu

CASES = [
    # dims, rows, graphs
    ([12, 32, 64, 128], 50_000, 1),      # point MLP of SPT-64
    ([12, 32, 64, 128], 40_001, 3),
    ([18, 32, 32], 70_000, 2),           # h_edge_mlp
    ([132, 64, 64], 30_000, 4),          # down / up in_mlp
    ([68, 64, 64], 20_000, 1),
    # piecewise-sorted index: the edge MLP of a 4-cloud batch, norm_index[edge_index[0]] sorted
    # inside each third of [i<j | j>i | loops] (src/models/components/spt.py:829-836) = 12 runs
    ([18, 32, 32], 60_000, -4),
    ([12, 32, 64], 33_333, -3),
]


@pytest.fixture(params=[1, 0, 2], ids=["bwd-split-bf16", "f32-pipe", "all-split-bf16"])
def gemm_mode(request):
    """1 = default (backward GEMMs on the bf16 pipe with split operands, forward exact f32),
    0 = f32 matrix pipe everywhere, 2 = forward split too (y then meets 2e-4, not 2e-5)."""
    from superpoint_transformer_amd import _lib
    prev = _lib.lib.spt_fused_linear_use_split_bf16(request.param)
    yield request.param
    _lib.lib.spt_fused_linear_use_split_bf16(prev)


@pytest.mark.parametrize("dims,rows,B", CASES)
def test_fused_mlp_matches_oracle_and_unfused_path(dims, rows, B, gemm_mode, dev):
    from superpoint_transformer_amd import nn as N
    g = torch.Generator().manual_seed(rows + B)
    mlp = N.MLP(dims, norm=N.GraphNorm)
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.randn(rows, dims[0], generator=g) * 2 + 0.5
    if B < 0:                                                            # three sorted thirds
        B = -B
        cuts = [0, rows // 3 + 7, 2 * rows // 3 - 5, rows]
        batch = torch.cat([torch.arange(b - a) * B // (b - a) for a, b in zip(cuts[:-1], cuts[1:])])
    else:
        batch = (torch.arange(rows) * B // rows) if B > 1 else None      # sorted clouds
    gw = torch.randn(rows, dims[-1], generator=g)

    ref = copy.deepcopy(mlp).double()
    OM.KEEP_GRAPH = True
    # LeakyReLU kinks: a pre-activation within the forward's own error of 0 takes one slope or the
    # other depending on the summation order (MFMA vs library vs f64): at these sizes a handful of
    # such elements per run is EXPECTED (4.5 M hidden values, density 0.4 per unit, f32 error
    # 3e-7; more in the all-split mode, whose forward bar is 2e-4).  The comparison judges the
    # arithmetic and not the coin flips - ONE-SIDEDLY, without tolerated outliers:
    #   * `near` = elements within eps of a kink in the f64 oracle (eps = 5 x the mode's forward bar);
    #   * rows within 1e-4 of a kink get a zero upstream gradient (so a flip cannot move the
    #     PARAMETER gradients by a row's whole contribution) - what still reaches them comes
    #     through the GraphNorm statistics (~1 / sqrt(rows) of a normal row's gradient) and does
    #     see the flip; near rows beyond 1e-4 (all-split mode) keep their full gradient;
    #   * the oracle runs three times: kinks as f64 saw them (y, parameter gradients, every row
    #     without a near element) and with every near element on the POSITIVE / NEGATIVE side:
    #     a row with one near element must match one of the two sides, whole row, to the bar.
    eps_kink = 1e-4 if gemm_mode < 2 else 1e-3

    class _LeakyFixedForward(torch.autograd.Function):
        """leaky_relu with the derivative given as a tensor: the three oracles share ONE forward
        (a one-sided slope applied in the forward as well would move later layers' pre-activations
        by up to eps and push OTHER elements across their kinks)."""

        @staticmethod
        def forward(ctx, h, slope, dslope):
            ctx.save_for_backward(dslope)
            return torch.nn.functional.leaky_relu(h, slope)

        @staticmethod
        def backward(ctx, g):
            return g * ctx.saved_tensors[0], None, None

    def oracle(side, gw_):
        """side None: leaky_relu as is; +1 / -1: the DERIVATIVE at near-kink elements is taken as
        slope 1 / negative_slope (the forward values are the same in all three)."""
        m = copy.deepcopy(mlp).double()
        xx = x.double().requires_grad_()
        pre, h = [], xx
        for layer in m.mlp:
            if isinstance(layer, torch.nn.Linear):
                h = h @ layer.weight.t()
            elif isinstance(layer, torch.nn.LeakyReLU):
                pre.append(h.detach())
                f64 = lambda v: torch.tensor(v, dtype=torch.float64)
                sl = torch.where(h.detach() > 0, f64(1.0), f64(layer.negative_slope))
                if side is not None:
                    sl = torch.where(h.detach().abs() <= eps_kink,
                                     f64(1.0 if side > 0 else layer.negative_slope), sl)
                h = _LeakyFixedForward.apply(h, layer.negative_slope, sl)
            else:
                h = OM.graph_norm(layer, h, batch, torch.float64)
        if gw_ is not None:
            (h * gw_.double()).sum().backward()
        return h.detach(), xx.grad, dict(m.named_parameters()), pre

    yr, _, _, pre_acts = oracle(None, None)
    assert torch.equal(yr, OM.mlp(copy.deepcopy(mlp).double(), x.double(), batch, torch.float64).detach())
    near = torch.zeros(rows, dtype=torch.long)
    masked = torch.zeros(rows, dtype=torch.bool)
    for pa in pre_acts:
        near += (pa.abs() <= eps_kink).sum(dim=1)
        masked |= (pa.abs() <= 1e-4).any(dim=1)
    safe, one = near == 0, near == 1
    # zero upstream gradient for the rows within f32 rounding (1e-4) of a kink, as before: a flip
    # there would move the PARAMETER gradients by the row's whole contribution
    gw = gw * (~masked).view(-1, 1).float()
    _, gx_ref, refp, _ = oracle(None, gw)
    _, gx_pos, _, _ = oracle(+1, gw)
    _, gx_neg, _, _ = oracle(-1, gw)
    OM.KEEP_GRAPH = False
    n_multi = int((near > 1).sum())
    print(f"rows with a near-kink element: {int(one.sum())} single, {n_multi} multiple of {rows} "
          f"(eps {eps_kink})")
    assert n_multi <= max(2, rows // (500 if gemm_mode < 2 else 20))      # (left out of the row checks)

    def run(fused):
        m = copy.deepcopy(mlp).to(dev)
        m.FUSE_MIN_ROWS = 0 if fused else 10 ** 12
        xd = x.to(dev).requires_grad_()
        y = m(xd, batch=None if batch is None else batch.to(dev), batch_size=B)
        (y * gw.to(dev)).sum().backward()
        return y.detach().cpu(), xd.grad.cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()}

    yf, gxf, gpf = run(True)
    yu, gxu, gpu_ = run(False)

    def relerr(a, r):
        return (a.double() - r).abs() / r.abs().clamp(min=1)

    def close(a, r, tol, name):
        err = relerr(a, r)
        assert float(err.max()) <= tol, f"{name}: {int((err > tol).sum())} elements above {tol}, max {err.max().item():.3e}"

    # the two one-sided oracles move ALL near elements at once, which shifts the GraphNorm
    # backward's statistics for every row of the graph (thousands of rows x their ~1 / sqrt(rows)
    # gradients in the all-split mode's wide eps): that collective shift is MEASURED on the rows
    # away from every kink (where it is the only difference between the three oracles) and is
    # the slack a near-kink row's one-sided comparison gets on top of the bar - derived, not chosen
    coll = max(float(relerr(gx_pos[safe], gx_ref[safe]).max()), float(relerr(gx_neg[safe], gx_ref[safe]).max()))

    def close_gx(a, name, tol=1e-4):
        """Every row without a near-kink element: the bar, no outliers.  Every row with ONE: the
        whole row matches the oracle with that element on one side of the kink or on the other."""
        close(a[safe], gx_ref[safe], tol, name + " (rows away from every kink)")
        if int(one.sum()):
            e_pos = relerr(a[one], gx_pos[one]).amax(dim=1)
            e_neg = relerr(a[one], gx_neg[one]).amax(dim=1)
            worst = torch.minimum(e_pos, e_neg)
            took_pos = int((e_pos <= e_neg).sum())
            print(f"{name}: {int(one.sum())} single-kink rows, {took_pos} on the positive side, "
                  f"worst one-sided error {worst.max().item():.3e} (bar {tol:.0e} + 1.5 x collective {coll:.2e})")
            # (1.5 x: the shift is measured on the rows AWAY from the kinks - another set of rows)
            assert float(worst.max()) <= tol + 1.5 * coll, (
                f"{name}: a near-kink row matches NEITHER side of its kink: {worst.max().item():.3e}")

    ytol = 2e-5 if gemm_mode < 2 else 2e-4
    close(yf, yr, ytol, "y")
    close_gx(gxf, "gx fused")
    close_gx(gxu, "gx unfused")
    for k in gpf:
        r = refp[k].grad
        scale = r.abs().max().clamp(min=1e-2)
        err = ((gpf[k].double() - r).abs() / scale).max().item()
        erru = ((gpu_[k].double() - r).abs() / scale).max().item()
        assert err <= max(2e-4, 3 * erru), f"{k}: fused {err:.3e} unfused {erru:.3e}"
    # fused and unfused HIP paths agree with each other (where no kink decides)
    close(yf, yu.double(), ytol, "y fused vs unfused")
    close(gxf[safe], gxu[safe].double(), 1e-4, "gx fused vs unfused (rows away from every kink)")


def test_fused_mlp_runs_of_a_piecewise_sorted_batch(dev):
    """The run table the fused kernels are launched with: one run per cloud of a sorted batch,
    the runs of every graph together for a piecewise-sorted one, None (-> layer-by-layer route)
    for an index with more than 16 runs."""
    from superpoint_transformer_amd import nn as N, ops
    rows = 20000
    sorted_b = (torch.arange(rows) * 3 // rows).to(dev)
    r = ops.graph_runs(sorted_b, 3, rows)
    assert r.sorted_batch and r.n == 3 and r.g == [0, 1, 2] and r.r0[0] == 0 and r.r1[-1] == rows
    third = torch.arange(rows // 4) * 3 // (rows // 4)
    piece = torch.cat([third, third, third, third]).to(dev)
    r = ops.graph_runs(piece, 3, rows)
    assert r is not None and not r.sorted_batch and r.n == 12 and r.g == sorted(r.g)
    assert r.rows_per_graph() == [int((piece == g).sum()) for g in range(3)]
    mlp = N.MLP([12, 32, 64], norm=N.GraphNorm).to(dev)
    x = torch.randn(rows, 12, device=dev)
    batch = torch.randint(0, 3, (rows,), device=dev)
    assert ops.graph_runs(batch, 3, rows) is None
    y = mlp(x, batch=batch, batch_size=3)            # layer-by-layer HIP path
    assert y.shape == (rows, 64) and bool(torch.isfinite(y).all())


@pytest.mark.parametrize("rows,nseg,B", [(50_000, 1500, 1), (40_001, 900, 3)])
def test_mlp_folded_into_the_max_pool_equals_mlp_then_pool(rows, nseg, B, dev, materialised_pool_route):
    """MLP.forward_max_pooled (last GraphNorm + LeakyReLU applied inside the pool's read of
    the raw activations) against the same MLP followed by the segment max-pool: pooled
    values bit-identical (same expression per element, hence the same arg rows); gradients
    equal up to the f64 summation order of the top GraphNorm's backward statistics (computed
    from the pool's sparse gradient on the folded route)."""
    from superpoint_transformer_amd import nn as N, ops
    g = torch.Generator().manual_seed(rows)
    mlp = N.MLP([12, 32, 64, 128], norm=N.GraphNorm).to(dev)
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=g).to(dev))
    x = (torch.randn(rows, 12, generator=g) * 2 + 0.5).to(dev)
    batch = (torch.arange(rows) * B // rows).to(dev) if B > 1 else None
    # segments never straddle two graphs; unsorted membership inside a graph
    seg_graph = (torch.arange(nseg) * B // nseg).to(dev)
    idx = torch.empty(rows, dtype=torch.long, device=dev)
    for b in range(B):
        rows_b = torch.where(batch == b)[0] if B > 1 else torch.arange(rows, device=dev)
        segs_b = torch.where(seg_graph == b)[0]
        idx[rows_b] = segs_b[torch.randint(0, segs_b.numel(), (rows_b.numel(),), generator=g).to(dev)]
    gout = torch.randn(nseg, 128, generator=g).to(dev)

    def run(fused):
        m = copy.deepcopy(mlp)
        xd = x.clone().requires_grad_()
        if fused:
            out = m.forward_max_pooled(xd, idx, nseg, batch=batch, batch_size=B,
                                       seg_graph=seg_graph if B > 1 else None)
            assert out is not None
        else:
            out = ops.segment_reduce(m(xd, batch=batch, batch_size=B), idx, nseg, "max")
        (out * gout).sum().backward()
        return out.detach(), xd.grad, [p.grad for p in m.parameters()]

    of, gxf, gpf = run(True)
    ou, gxu, gpu_ = run(False)
    assert torch.equal(of, ou)
    assert torch.allclose(gxf, gxu, rtol=1e-5, atol=1e-6 * float(gxu.abs().max()))
    # the folded route's top layer walks the rows in the pool's CSR order (the pool's gradient
    # is consumed directly): its weight gradient is the same f32 sum over 4e4-5e4 rows in another
    # order
    for a, b in zip(gpf, gpu_):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max()))


@pytest.mark.parametrize("K,N", [(64, 128), (32, 64), (64, 64)])
@pytest.mark.parametrize("rows,nseg", [(10_007, 300), (4_099, 1500), (17, 2)])
@pytest.mark.parametrize("pooled", [True, False])
def test_dma_staged_backward_is_bitwise_the_register_staged_one(K, N, rows, nseg, pooled, dev):
    """csrc/fused_mlp_dma.hip (tiles staged by global_load_lds, swizzled LDS layout, gx leaving as
    whole rows) against csrc/fused_mlp.hip's register-staged kernels through the C ABI: same
    products, same summation order inside a tile and across the per-wave tables - gx, gW and the
    f64 statistics agree BIT FOR BIT.  Covers a short last tile (rows % 16 != 0), tiles over more
    segments than one (gout, arg) DMA holds (nseg 1500: ~3 rows per segment -> the global-load
    path) and fewer tiles than waves."""
    from superpoint_transformer_amd import _lib
    g = torch.Generator().manual_seed(rows + K + N)
    si = torch.randint(0, nseg, (rows,), generator=g)
    si[:nseg] = torch.arange(nseg)                              # every segment has a row
    perm = torch.argsort(si, stable=True)
    pos_seg = si[perm]
    rowptr = torch.zeros(nseg + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(si, minlength=nseg), 0)
    off = (torch.rand(nseg, N, generator=g) * (rowptr[1:] - rowptr[:-1]).view(-1, 1)).long()
    arg = perm[rowptr[:-1].view(-1, 1) + off]
    t = lambda *s: torch.randn(*s, generator=g).to(dev)
    gout, h, x, W, gy = t(nseg, N), t(rows, N), t(rows, K), t(N, K) * 0.1, t(rows, N)
    tabN = [(torch.rand(N, generator=g) + 0.5).to(dev) for _ in range(6)]
    tabK = [(torch.rand(K, generator=g) + 0.5).to(dev) for _ in range(3)]
    perm_d, seg_d, arg_d = perm.int().to(dev), pos_seg.int().to(dev), arg.int().contiguous().to(dev)
    ws = torch.empty(_lib.lib.spt_fused_linear_workspace_bytes(K, N), dtype=torch.uint8, device=dev)
    P = _lib.ptr

    def run(mode):
        gx = torch.full((rows, K), float("nan"), device=dev)
        gW = torch.empty(N, K, device=dev)
        prev = torch.empty(2 * K + 1, dtype=torch.float64, device=dev)
        if pooled:
            st = _lib.lib.spt_fused_linear_bwd_pooled_ex_f32(
                P(gout), P(arg_d), P(perm_d), P(seg_d), P(h), 0, rows, N, P(tabN[0]), P(tabN[1]),
                P(tabN[2]), 0.01, P(tabN[3]), P(tabN[4]), P(tabN[5]), P(x), K, P(tabK[0]), P(tabK[1]),
                P(tabK[2]), 0.2, P(W), P(gx), P(gW), 0, P(prev), mode, P(ws), ws.numel(),
                _lib.stream_ptr(dev))
        else:
            st = _lib.lib.spt_fused_linear_bwd_ex_f32(
                P(gy), P(h), 0, rows, N, P(tabN[0]), P(tabN[1]), P(tabN[2]), 0.01, P(tabN[3]),
                P(tabN[4]), P(tabN[5]), P(x), K, P(tabK[0]), P(tabK[1]), P(tabK[2]), 0.2, P(W), P(gx),
                P(gW), 0, P(prev), mode, P(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(st, "fused backward")
        torch.cuda.synchronize()
        return gx, gW, prev

    for precision in (1, 3):                                    # split-bf16, bf16
        a = run(precision)
        b = run(precision | 4)                                  # SPT_FMLP_BWD_REGISTER_STAGED
        assert bool(torch.isfinite(a[0]).all())                 # every row of gx was written
        for u, v in zip(a, b):
            assert torch.equal(u, v)


@pytest.mark.parametrize("K,N,pooled", [(64, 128, True), (32, 64, False), (32, 64, True), (12, 32, False),
                                        (64, 64, True)])
@pytest.mark.parametrize("rows,nseg,B", [(10_007, 300, 1), (20_011, 700, 3)])
def test_bf16_storage_kernels_are_bitwise_their_f32_storage_siblings(K, N, pooled, rows, nseg, B, dev):
    """The bf16 mode's activation storage (mode-word bits SPT_FMLP_H_BF16 / SPT_FMLP_X_BF16): fed
    bf16-REPRESENTABLE activations, the storage kernels (bf16 rows in HBM: fused forward with the
    packed output tile, DMA-staged backward with in-LDS widening, register-staged backward) and
    the f32-storage kernels of the same matrix mode see the same numbers - every output must agree
    BIT FOR BIT (h: the f32 kernel's output rounded to bf16).  Several graphs per launch (run
    tables), a ragged last tile."""
    import ctypes
    from superpoint_transformer_amd import _lib
    g = torch.Generator().manual_seed(rows + K + N + B)
    bf = lambda t: t.to(torch.bfloat16)
    P = _lib.ptr
    sp = _lib.stream_ptr(dev)
    first = K == 12                                           # the chain's first layer reads f32 input
    x32 = (torch.randn(rows, K, generator=g) * 2).to(dev)
    if not first:
        x32 = bf(x32).float()
    x16 = bf(x32)
    W = (torch.randn(N, K, generator=g) * 0.2).to(dev)
    tabK = [(torch.rand(B, K, generator=g) + 0.5).to(dev) for _ in range(2)] + \
           [(torch.rand(K, generator=g) + 0.5).to(dev)]
    cuts = [rows * b // B for b in range(B + 1)]
    n = B
    r0 = (ctypes.c_int64 * n)(*cuts[:-1]); r1 = (ctypes.c_int64 * n)(*cuts[1:])
    gg = (ctypes.c_int32 * n)(*range(B))
    ws = torch.empty(_lib.lib.spt_fused_linear_workspace_bytes(K, N), dtype=torch.uint8, device=dev)
    H, X = 8, 16

    # ---- forward --------------------------------------------------------------------------------
    def fwd(store):
        h = torch.empty((rows, N), dtype=torch.bfloat16 if store else torch.float32, device=dev)
        tot = torch.empty((B, 2 * N + 1), dtype=torch.float64, device=dev)
        mode = 3 | ((H | (0 if first else X)) if store else 0)
        xin = x16 if (store and not first) else x32
        pre = (None, None, None) if first else tuple(P(t) for t in tabK)
        st = _lib.lib.spt_fused_linear_fwd_runs_f32(
            P(xin), n, r0, r1, gg, B, K, P(W), N, pre[0], pre[1], pre[2], 0.2, P(h), P(tot), mode,
            P(ws), ws.numel(), sp)
        _lib.check(st, "fwd")
        torch.cuda.synchronize()
        return h, tot
    h16, tot16 = fwd(True)
    h32, tot32 = fwd(False)
    assert torch.equal(h16, bf(h32))
    hd = h16.double()
    for b in range(B):                                        # statistics of the ROUNDED values
        seg = hd[cuts[b]:cuts[b + 1]]
        ref = torch.cat([seg.sum(0), (seg * seg).sum(0), torch.tensor([float(seg.shape[0])], device=dev, dtype=torch.float64)])
        assert torch.allclose(tot16[b], ref, rtol=1e-12, atol=1e-9)

    # ---- backward -------------------------------------------------------------------------------
    hq32 = h16.float()                                        # bf16-representable h for both kernels
    tabN = [(torch.rand(B, N, generator=g) + 0.5).to(dev) for _ in range(2)] + \
           [(torch.rand(N, generator=g) + 0.5).to(dev)] + \
           [(torch.rand(B, N, generator=g) + 0.5).to(dev) for _ in range(3)]
    gy = torch.randn(rows, N, generator=g).to(dev)
    # pooled: segments numbered graph by graph, CSR positions of a graph contiguous
    seg_of_row = torch.empty(rows, dtype=torch.long)
    for b in range(B):
        lo, hi = nseg * b // B, nseg * (b + 1) // B
        seg_of_row[cuts[b]:cuts[b + 1]] = torch.randint(lo, hi, (cuts[b + 1] - cuts[b],), generator=g)
        seg_of_row[cuts[b]:cuts[b] + (hi - lo)] = torch.arange(lo, hi)
    perm = torch.argsort(seg_of_row, stable=True)
    pos_seg = seg_of_row[perm]
    rowptr = torch.zeros(nseg + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(seg_of_row, minlength=nseg), 0)
    off = (torch.rand(nseg, N, generator=g) * (rowptr[1:] - rowptr[:-1]).view(-1, 1)).long()
    arg = perm[rowptr[:-1].view(-1, 1) + off].int().contiguous().to(dev)
    gout = torch.randn(nseg, N, generator=g).to(dev)
    perm_d, seg_d = perm.int().to(dev), pos_seg.int().to(dev)

    def bwd(store, extra=0):
        gx = torch.full((rows, K), float("nan"), device=dev)
        gW = torch.empty(N, K, device=dev)
        prev = None if first else torch.empty((B, 2 * K + 1), dtype=torch.float64, device=dev)
        mode = 3 | extra | ((H | (0 if first else X)) if store else 0)
        hin = h16 if store else hq32
        xin = x16 if (store and not first) else x32
        pre = (None, None, None) if first else tuple(P(t) for t in tabK)
        if pooled:
            st = _lib.lib.spt_fused_linear_bwd_pooled_runs_f32(
                P(gout), P(arg), P(perm_d), P(seg_d), P(hin), n, r0, r1, gg, B, N, P(tabN[0]), P(tabN[1]),
                P(tabN[2]), 0.01, P(tabN[3]), P(tabN[4]), P(tabN[5]), P(xin), K, pre[0], pre[1], pre[2],
                0.2, P(W), P(gx), P(gW), P(prev), mode, P(ws), ws.numel(), sp)
        else:
            st = _lib.lib.spt_fused_linear_bwd_runs_f32(
                P(gy), P(hin), n, r0, r1, gg, B, N, P(tabN[0]), P(tabN[1]), P(tabN[2]), 0.01, P(tabN[3]),
                P(tabN[4]), P(tabN[5]), P(xin), K, pre[0], pre[1], pre[2], 0.2, P(W), P(gx), P(gW),
                P(prev), mode, P(ws), ws.numel(), sp)
        _lib.check(st, "bwd")
        torch.cuda.synchronize()
        return gx, gW, prev
    if pooled and first:
        return
    ref = bwd(False)
    for extra in (0, 4):                                      # DMA-staged where built / register-staged
        got = bwd(True, extra)
        assert bool(torch.isfinite(got[0]).all())
        for u, v in zip(got, ref):
            if u is not None:
                assert torch.equal(u, v)


def test_bf16_rows_in_the_streaming_segment_max_and_the_sparse_statistics(dev):
    """spt_segcsr_max_affine_bf16 / spt_graphnorm_bwd_stats_sparse_ex_f32(x_is_bf16) on bf16 rows
    against the f32 entries on the same (bf16-representable) values: bit-identical outputs."""
    from superpoint_transformer_amd import _lib
    from superpoint_transformer_amd.csr import build_csr
    g = torch.Generator().manual_seed(77)
    rows, nseg, C, B = 70_003, 2_100, 128, 2
    h16 = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev)
    h32 = h16.float()
    si = torch.randint(0, nseg, (rows,), generator=g)
    si[:nseg] = torch.arange(nseg)
    si = si.sort().values[torch.randperm(rows, generator=g)]                 # unsorted membership
    seg_graph = (torch.arange(nseg) * B // nseg).to(dev)
    csr = build_csr(si.to(dev), nseg)
    am, sc = ((torch.rand(B, C, generator=g) - 0.3).to(dev) for _ in range(2))
    bs = torch.rand(C, generator=g).to(dev)
    P, sp = _lib.ptr, _lib.stream_ptr(dev)

    def pool(fn, x):
        out = torch.empty((nseg, C), device=dev)
        arg = torch.empty((nseg, C), dtype=torch.int32, device=dev)
        st = fn(P(x), P(csr.perm), P(csr.rowptr), rows, nseg, C, P(am), P(sc), P(bs), 0.01,
                P(seg_graph), P(out), P(arg), sp)
        _lib.check(st, "segmax")
        return out, arg
    assert _lib.lib.spt_segcsr_max_affine_bf16_supported(C, rows) == 1
    o16, a16 = pool(_lib.lib.spt_segcsr_max_affine_bf16, h16)
    o32, a32 = pool(_lib.lib.spt_segcsr_max_affine_f32, h32)
    assert torch.equal(o16, o32) and torch.equal(a16, a32)

    gout = torch.randn(nseg, C, generator=g).to(dev)
    grows = torch.tensor([int((seg_graph[si.to(dev)] == b).sum()) for b in range(B)], device=dev)
    nb = _lib.lib.spt_graphnorm_bwd_stats_sparse_workspace_bytes(nseg, C, B)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)

    def stats(x, is16):
        tot = torch.empty((B, 2 * C + 1), dtype=torch.float64, device=dev)
        st = _lib.lib.spt_graphnorm_bwd_stats_sparse_ex_f32(
            P(x), is16, P(gout), P(a32), P(seg_graph), P(grows), nseg, rows, C, B, P(am), P(sc), P(bs),
            0.01, P(tot), P(ws), nb, sp)
        _lib.check(st, "sparse stats")
        return tot
    assert torch.equal(stats(h16, 1), stats(h32, 0))


def test_raw_output_of_the_pool_feeds_the_sparse_statistics(dev):
    """spt_segcsr_max_affine_raw_f32: same (out, arg) as the two-output entries, raw = x[arg] (0 for
    an empty segment), f32 and bf16 rows; spt_graphnorm_bwd_stats_sparse_raw_f32 on it = the
    gathering entry, bit for bit."""
    from superpoint_transformer_amd import _lib
    from superpoint_transformer_amd.csr import build_csr
    g = torch.Generator().manual_seed(78)
    rows, nseg, C, B = 70_011, 2_300, 128, 2
    h16 = torch.randn(rows, C, generator=g).to(torch.bfloat16).to(dev)
    h32 = h16.float()
    si = torch.randint(0, nseg, (rows,), generator=g)
    si[si % 17 == 3] = 5                                                     # empty segments
    si = si.sort().values[torch.randperm(rows, generator=g)]                 # unsorted membership
    seg_graph = (torch.arange(nseg) * B // nseg).to(dev)
    csr = build_csr(si.to(dev), nseg)
    am, sc = ((torch.rand(B, C, generator=g) - 0.3).to(dev) for _ in range(2))
    bs = torch.rand(C, generator=g).to(dev)
    P, sp = _lib.ptr, _lib.stream_ptr(dev)
    assert _lib.lib.spt_segcsr_max_affine_raw_supported(C, rows) == 1

    def pool(x, is16, want_raw):
        out = torch.empty((nseg, C), device=dev)
        arg = torch.empty((nseg, C), dtype=torch.int32, device=dev)
        raw = torch.full((nseg, C), float("nan"), device=dev)
        if want_raw:
            st = _lib.lib.spt_segcsr_max_affine_raw_f32(
                P(x), is16, P(csr.perm), P(csr.rowptr), rows, nseg, C, P(am), P(sc), P(bs), 0.01,
                P(seg_graph), P(out), P(arg), P(raw), sp)
        else:
            fn = _lib.lib.spt_segcsr_max_affine_bf16 if is16 else _lib.lib.spt_segcsr_max_affine_f32
            st = fn(P(x), P(csr.perm), P(csr.rowptr), rows, nseg, C, P(am), P(sc), P(bs), 0.01,
                    P(seg_graph), P(out), P(arg), sp)
        _lib.check(st, "segmax")
        return out, arg, raw
    o_ref, a_ref, _ = pool(h32, 0, False)
    empty = a_ref[:, 0] == rows
    assert int(empty.sum()) > 50
    expect = torch.gather(h32, 0, a_ref.long().clamp(max=rows - 1))
    expect[empty] = 0.0
    for x, is16 in ((h32, 0), (h16, 1)):
        o, a, raw = pool(x, is16, True)
        assert torch.equal(o, o_ref) and torch.equal(a, a_ref)
        assert torch.equal(raw, expect)

    gout = torch.randn(nseg, C, generator=g).to(dev)
    grows = torch.tensor([int((seg_graph[si.to(dev)] == b).sum()) for b in range(B)], device=dev)
    nb = _lib.lib.spt_graphnorm_bwd_stats_sparse_workspace_bytes(nseg, C, B)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    t_gather = torch.empty((B, 2 * C + 1), dtype=torch.float64, device=dev)
    st = _lib.lib.spt_graphnorm_bwd_stats_sparse_ex_f32(
        P(h32), 0, P(gout), P(a_ref), P(seg_graph), P(grows), nseg, rows, C, B, P(am), P(sc), P(bs),
        0.01, P(t_gather), P(ws), nb, sp)
    _lib.check(st, "sparse stats")
    t_raw = torch.empty_like(t_gather)
    st = _lib.lib.spt_graphnorm_bwd_stats_sparse_raw_f32(
        P(expect), P(gout), P(a_ref), P(seg_graph), P(grows), nseg, rows, C, B, P(am), P(sc), P(bs),
        0.01, P(t_raw), P(ws), nb, sp)
    _lib.check(st, "sparse stats (raw)")
    assert torch.allclose(t_raw, t_gather, rtol=1e-13, atol=1e-13)


def test_fused_mlp_max_pool_with_and_without_the_raw_output(dev, materialised_pool_route):
    """MLP.forward_max_pooled at a size where the pool's raw output is built (128 channels,
    >= 65 536 rows): same pooled values, gradients equal up to the f64 summation order of the top
    statistics, with the raw output feeding them (default) and with the gathering statistics."""
    from superpoint_transformer_amd import nn as N, ops
    g = torch.Generator().manual_seed(79)
    rows, nseg, B = 70_000, 2_000, 2
    mlp = N.MLP([12, 32, 64, 128], norm=N.GraphNorm).to(dev)
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=g).to(dev))
    x = (torch.randn(rows, 12, generator=g) * 2 + 0.5).to(dev)
    batch = (torch.arange(rows) * B // rows).to(dev)
    seg_graph = (torch.arange(nseg) * B // nseg).to(dev)
    idx = torch.empty(rows, dtype=torch.long, device=dev)
    for b in range(B):
        rows_b = torch.where(batch == b)[0]
        segs_b = torch.where(seg_graph == b)[0][3:]                          # a few empty segments
        idx[rows_b] = segs_b[torch.randint(0, segs_b.numel(), (rows_b.numel(),), generator=g).to(dev)]
    gout = torch.randn(nseg, 128, generator=g).to(dev)

    def run(flag):
        old, ops.POOL_RAW_OUTPUT = ops.POOL_RAW_OUTPUT, flag
        try:
            m = copy.deepcopy(mlp)
            xd = x.clone().requires_grad_()
            out = m.forward_max_pooled(xd, idx, nseg, batch=batch, batch_size=B, seg_graph=seg_graph)
            assert out is not None
            (out * gout).sum().backward()
            return out.detach(), xd.grad, [p.grad for p in m.parameters()]
        finally:
            ops.POOL_RAW_OUTPUT = old
    o1, gx1, gp1 = run(True)
    o0, gx0, gp0 = run(False)
    assert torch.equal(o1, o0)
    assert torch.allclose(gx1, gx0, rtol=1e-6, atol=1e-7 * float(gx0.abs().max()))
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))


@pytest.mark.parametrize("dims,rows,B", [([12, 32, 64, 128], 40_001, 3), ([18, 32, 32], 60_000, -4),
                                         ([132, 64, 64], 30_000, 1)])
def test_post_launch_is_bitwise_the_separate_table_kernels(dims, rows, B, dev):
    """Round 6: a fused layer's table sums and the GraphNorm table kernel behind them run as ONE
    post launch (spt_fused_linear_*_gn_f32).  Same column sums in the same order, same formulas:
    outputs, input gradient and every parameter gradient are bitwise those of the separate
    kernels (ops.fuse_post(False)) - for the plain chain and for the pool-fused top layer."""
    from superpoint_transformer_amd import nn as N, ops
    g = torch.Generator().manual_seed(rows)
    mlp = N.MLP(dims, norm=N.GraphNorm)
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.randn(rows, dims[0], generator=g)
    if B < 0:
        B = -B
        cuts = [0, rows // 3 + 7, 2 * rows // 3 - 5, rows]
        batch = torch.cat([torch.arange(b - a) * B // (b - a) for a, b in zip(cuts[:-1], cuts[1:])])
        sorted_batch = False
    else:
        batch = (torch.arange(rows) * B // rows) if B > 1 else None
        sorted_batch = True
    gw = torch.randn(rows, dims[-1], generator=g)
    nseg = 700
    seg_graph = torch.arange(nseg) * B // nseg
    si = torch.empty(rows, dtype=torch.long)
    for b in range(B):
        rmask = (batch == b) if batch is not None else torch.ones(rows, dtype=torch.bool)
        segs = torch.nonzero(seg_graph == b).flatten()
        si[rmask] = segs[torch.randint(0, segs.numel(), (int(rmask.sum()),), generator=g)]
    gout = torch.randn(nseg, dims[-1], generator=g)

    def run(fuse, pooled):
        old = ops.fuse_post(fuse)
        try:
            m = copy.deepcopy(mlp).to(dev)
            m.FUSE_MIN_ROWS = 0
            xd = x.to(dev).requires_grad_()
            bd = None if batch is None else batch.to(dev)
            if pooled:
                y = m.forward_max_pooled(xd, si.to(dev), nseg, batch=bd, batch_size=B,
                                         seg_graph=seg_graph.to(dev) if B > 1 else None)
                if y is None:
                    return None
                (y * gout.to(dev)).sum().backward()
            else:
                y = m(xd, batch=bd, batch_size=B)
                (y * gw.to(dev)).sum().backward()
            return [y.detach(), xd.grad] + [p.grad for p in m.parameters()]
        finally:
            ops.fuse_post(old)

    for pooled in ([False, True] if (sorted_batch and dims[-1] in (64, 128)) else [False]):
        a, b = run(True, pooled), run(False, pooled)
        if a is None:
            continue
        for i, (u, v) in enumerate(zip(a, b)):
            assert torch.equal(u, v), f"pooled={pooled}: tensor {i} differs by {float((u - v).abs().max()):.3e}"
