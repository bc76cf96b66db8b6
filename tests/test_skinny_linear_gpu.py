"""GPU parity of the tall-skinny Linear kernel (qkv / out_proj of the attention
block and their input gradients) against torch in float64.  f32 in / f32 accumulate:
the error is that of an fmaf chain over K terms - bar 2e-6 * K * max|x| * max|w|."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", [1, 15, 16, 17, 4099, 70001])
@pytest.mark.parametrize("K,N", [(64, 192), (64, 64), (192, 64), (32, 64), (128, 128), (192, 192),
                                 # the KITTI-360 width's node MLPs ([132, 128, 128], [260, 128, 128],
                                 # kitti360.yaml:22-27): wide inputs, and their dX (132 / 260 OUTPUT columns)
                                 (132, 128), (260, 128), (128, 132), (128, 260), (256, 128)])
@pytest.mark.parametrize("bias", [True, False])
def test_forward_matches_float64(rows, K, N, bias, dev):
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.2
    b = torch.randn(N, generator=g) if bias else None
    y = ops._skinny_launch(x.to(dev), w.to(dev), None if b is None else b.to(dev)).cpu().double()
    ref = torch.nn.functional.linear(x.double(), w.double(), None if b is None else b.double())
    tol = 2e-6 * K * float(x.abs().max()) * float(w.abs().max())
    assert (y - ref).abs().max() < tol


def test_linear_autograd_matches_torch(dev):
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(0)
    for rows, K, N in ((5000, 64, 192), (140000, 64, 64), (9000, 48, 64),   # (48, 64): library path
                       (35000, 132, 128), (14500, 260, 128)):               # the SPT-128 node MLPs
        x = torch.randn(rows, K, generator=g).to(dev).requires_grad_()
        w = (torch.randn(N, K, generator=g) * 0.2).to(dev).requires_grad_()
        b = torch.randn(N, generator=g).to(dev).requires_grad_()
        go = torch.randn(rows, N, generator=g).to(dev)
        y = ops.linear(x, w, b)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
        xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
        yr = torch.nn.functional.linear(xd, wd, bd)
        rx, rw, rb = torch.autograd.grad(yr, (xd, wd, bd), go.double())
        for a, r in ((y, yr), (gx, rx), (gw, rw), (gb, rb)):
            assert (a.double() - r).abs().max() <= 1e-5 * max(1.0, float(r.abs().max())) * (K + N) ** 0.5


def test_unsupported_shapes_and_small_inputs_use_the_library(dev):
    from superpoint_transformer_amd import ops, _lib
    assert not _lib.lib.spt_skinny_linear_supported(48, 64)
    assert not _lib.lib.spt_skinny_linear_supported(64, 40)
    assert _lib.lib.spt_skinny_linear_supported(132, 128) and _lib.lib.spt_skinny_linear_supported(128, 260)
    assert _lib.lib.spt_skinny_dw_supported(260, 128) and not _lib.lib.spt_skinny_dw_supported(260, 132)
    assert _lib.lib.spt_skinny_linear_supported(64, 13)        # narrow heads are built
    x = torch.randn(100, 64, device=dev)
    w = torch.randn(192, 64, device=dev)
    assert torch.equal(ops.linear(x, w), torch.nn.functional.linear(x, w))
    with pytest.raises(RuntimeError):
        ops._skinny_launch(torch.randn(5000, 48, device=dev), torch.randn(64, 48, device=dev), None)


@pytest.mark.parametrize("rows", [1, 15, 16, 17, 4099, 70001, 428_571])
@pytest.mark.parametrize("K,N", [(64, 192), (64, 64), (32, 64), (32, 128), (128, 128), (128, 256),
                                 (132, 128), (260, 128)])
@pytest.mark.parametrize("mode", [0, 1], ids=["f32-pipe", "split-bf16"])
def test_weight_gradient_kernel_matches_float64(rows, K, N, mode, dev):
    """dW = G^T X on the skinny dW kernel (per-wave partials, fixed-order sum) against float64, on
    the f32 pipe (the f32-exact mode: the error is that of an f32 sum over `rows` terms - bar
    2e-6 * sqrt(rows) of the largest |g||x| product scale) and on the bf16 pipe with split operands
    (the default mode, round 6: hi*hi + lo*hi + hi*lo drops ~2^-17 of every product on top of that
    sum - bar 4e-6 * sqrt(rows); measured 2.0e-6 .. 2.1e-6 of the same scale); incl. last tiles that
    are not full and fewer tiles than waves; deterministic (two runs agree bit for bit)."""
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g)
    go = torch.randn(rows, N, generator=g)
    gw, gb = ops._skinny_dw(go.to(dev), x.to(dev), want_bias=True, mode=mode)
    gw2 = ops._skinny_dw(go.to(dev), x.to(dev), mode=mode)
    assert torch.equal(gw, gw2)
    rb = go.double().sum(0)
    assert (gb.cpu().double() - rb).abs().max() < 2e-6 * max(rows, 16) ** 0.5 * float(go.abs().max()) + 1e-6 * float(rb.abs().max())
    ref = go.double().t() @ x.double()
    bar = 2e-6 if mode == 0 else 4e-6
    tol = bar * max(rows, 16) ** 0.5 * float(go.abs().max()) * float(x.abs().max()) + 1e-6 * float(ref.abs().max())
    assert (gw.cpu().double() - ref).abs().max() < tol


@pytest.mark.parametrize("rows", [4096, 4099, 70001, 428_571])
@pytest.mark.parametrize("K,N", [(64, 13), (64, 16), (64, 5), (32, 13), (128, 7)])
def test_narrow_head_linear_autograd_matches_float64(rows, K, N, dev):
    """The classifier heads (64 -> 13): forward on the skinny kernel's narrow variant; backward
    (dX, dW, db) in one pass on the narrow kernel where K = 64, else on the library.  Against
    float64 torch with the bars of the wide kernels."""
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, K, generator=g).to(dev).requires_grad_()
    w = (torch.randn(N, K, generator=g) * 0.2).to(dev).requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    go = torch.randn(rows, N, generator=g).to(dev)
    y = ops.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    yr = torch.nn.functional.linear(xd, wd, bd)
    rx, rw, rb = torch.autograd.grad(yr, (xd, wd, bd), go.double())
    assert (y.double() - yr).abs().max() < 2e-6 * K * float(x.abs().max()) * float(w.abs().max())
    assert (gx.double() - rx).abs().max() < 2e-6 * N * float(go.abs().max()) * float(w.abs().max()) + 1e-6
    tol = 2e-6 * rows ** 0.5 * float(go.abs().max()) * float(x.abs().max())
    assert (gw.double() - rw).abs().max() < tol + 1e-6 * float(rw.abs().max())
    assert (gb.double() - rb).abs().max() < tol + 1e-6 * float(rb.abs().max())


@pytest.mark.parametrize("rows", [1, 17, 4099, 70001])
@pytest.mark.parametrize("n_out,n_in", [(192, 64), (64, 64), (128, 128), (256, 128), (64, 132), (128, 132),
                                        (128, 260), (32, 64), (192, 128), (128, 256)])
def test_input_gradient_reads_the_weight_transposed_bit_for_bit(rows, n_out, n_in, dev):
    """dX = G W with the layer's own weight [n_out, n_in] (spt_skinny_linear_wt_m_f32: the slab is
    transposed while it is staged) equals the same kernel on a transposed COPY bit for bit - the
    copy was one torch launch per Linear and backward call.  Bit for bit in the modes whose forward
    and backward use the same products (f32 pipe, bf16); in the default mode (forward six products,
    backward three) against float64 at the split scheme's bar."""
    from superpoint_transformer_amd import _lib, ops
    g = torch.Generator().manual_seed(rows + n_out + 7 * n_in)
    go = torch.randn(rows, n_out, generator=g).to(dev)
    w = (torch.randn(n_out, n_in, generator=g) * 0.2).to(dev)
    if not _lib.lib.spt_skinny_linear_supported(n_out, n_in):
        pytest.skip("shape not built")

    def wt_entry(mode):
        y = torch.full((rows, n_in), float("nan"), device=dev)
        st = _lib.lib.spt_skinny_linear_wt_m_f32(_lib.ptr(go), rows, n_out, _lib.ptr(w), n_in, _lib.ptr(y),
                                                 mode, _lib.stream_ptr(dev))
        _lib.check(st, "spt_skinny_linear_wt_m_f32")
        return y

    for mode in (0, 3):
        assert torch.equal(wt_entry(mode), ops._skinny_launch(go, w.t().contiguous(), None, mode))
    ref = go.double().cpu() @ w.double().cpu()
    y1 = wt_entry(1).cpu().double()
    tol = 4e-6 * n_out ** 0.5 * float(go.abs().max()) * float(w.abs().max()) + 1e-6 * float(ref.abs().max())
    assert (y1 - ref).abs().max() < tol
    old = ops._SKINNY_MIN_ROWS
    ops._SKINNY_MIN_ROWS = 1
    try:
        assert torch.equal(ops._input_grad(go, w, 0), wt_entry(0))
    finally:
        ops._SKINNY_MIN_ROWS = old


@pytest.mark.parametrize("K,N", [(64, 192), (64, 64), (32, 64)])
def test_matrix_modes_of_the_skinny_linears(K, N, dev):
    """The three modes of the forward (spt_skinny_linear_pre_m_f32) against float64: f32 pipe and the
    six-product split at the f32 bar (the split is f32-exact), bf16 operands at 2^-8 of the product
    scale; the f32-pipe mode is the kernel every mode ran before round 6."""
    from superpoint_transformer_amd import ops
    g = torch.Generator().manual_seed(K * 7 + N)
    rows = 50_000
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.2
    b = torch.randn(N, generator=g)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    scale = float(x.abs().max()) * float(w.abs().max())
    for mode, bar in ((0, 2e-6 * K), (1, 2e-6 * K), (3, 2.0 ** -8 * K ** 0.5)):
        y = ops._skinny_launch(x.to(dev), w.to(dev), b.to(dev), mode).cpu().double()
        assert (y - ref).abs().max() < bar * scale, (mode, float((y - ref).abs().max()), bar * scale)
