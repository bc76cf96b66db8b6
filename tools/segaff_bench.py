"""L0 -> L1 segment-max with the fused GraphNorm + LeakyReLU map (spt_segcsr_max_affine_f32) alone:
row-streaming vs lane-group kernel, with and without a seg_graph vector."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import _lib
from superpoint_transformer_amd.csr import csr_of

dev = torch.device("cuda:0")
n, nseg, c = 15_000_000, 428_571, 128
g = torch.Generator(device=dev).manual_seed(0)
w = torch.exp(torch.randn(nseg, device=dev, generator=g))
idx = torch.multinomial(w / w.sum(), n, replacement=True, generator=g)
x = torch.randn(n, c, device=dev, generator=g)
csr = csr_of(idx, nseg)
am = torch.randn(1, c, device=dev); sc = torch.rand(1, c, device=dev) + 0.5; bs = torch.randn(c, device=dev)
out = torch.empty(nseg, c, device=dev); arg = torch.empty(nseg, c, dtype=torch.int32, device=dev)
sg0 = torch.zeros(nseg, dtype=torch.int64, device=dev)
for name, sg in (("seg_graph=None", None), ("seg_graph=zeros", sg0)):
    for on in (1, 0):
        _lib.lib.spt_segcsr_use_stream(on)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = _lib.lib.spt_segcsr_max_affine_f32(_lib.ptr(x), _lib.ptr(csr.perm), _lib.ptr(csr.rowptr), n, nseg, c,
                                                   _lib.ptr(am), _lib.ptr(sc), _lib.ptr(bs), 0.01, _lib.ptr(sg),
                                                   _lib.ptr(out), _lib.ptr(arg), _lib.stream_ptr(dev))
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name} stream={on}: {1e3 * min(ts[1:]):.3f} ms")
