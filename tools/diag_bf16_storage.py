import copy, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from superpoint_transformer_amd import nn as N, precision
from oracle import spt_oracle as O, spt_model as OM2
dev=torch.device('cuda:0')
for B in (1,3):
    g = torch.Generator().manual_seed(31 + B)
    rows, nseg = 70_001, 2_300
    mlp = N.MLP([12, 32, 64, 128], norm=N.GraphNorm).to(dev)
    with torch.no_grad():
        for p in mlp.parameters(): p.add_(0.1 * torch.randn(p.shape, generator=g).to(dev))
    x = torch.randn(rows, 12, generator=g) * 2 + 0.5
    batch = (torch.arange(rows) * B // rows) if B > 1 else None
    seg_of_graph = torch.arange(nseg) * B // nseg
    si = torch.empty(rows, dtype=torch.long)
    for b in range(B):
        rmask = (batch == b) if batch is not None else torch.ones(rows, dtype=torch.bool)
        segs = torch.nonzero(seg_of_graph == b).flatten()
        si[rmask] = segs[torch.randint(0, segs.numel(), (int(rmask.sum()),), generator=g)]
    gw = torch.randn(nseg, 128, generator=g)
    def run(storage, mode="bf16"):
        prev = precision.set_bf16_activation_storage(storage)
        try:
            with precision.matrix_precision(mode):
                m = copy.deepcopy(mlp); xd = x.to(dev).requires_grad_()
                out = m.forward_max_pooled(xd, si.to(dev), nseg, batch=None if batch is None else batch.to(dev), batch_size=B, seg_graph=seg_of_graph.to(dev))
                (out * gw.to(dev)).sum().backward()
                return out.detach().cpu(), xd.grad.cpu(), [p.grad.cpu() for p in m.parameters()]
        finally: precision.set_bf16_activation_storage(prev)
    o1,gx1,gp1=run(True); o0,gx0,gp0=run(False); of,gxf,gpf=run(False,"f32")
    ref = copy.deepcopy(mlp).double().cpu(); x64 = x.double().requires_grad_()
    OM2.KEEP_GRAPH=True; y64 = OM2.mlp(ref, x64, batch, torch.float64); OM2.KEEP_GRAPH=False
    p64,_ = O.scatter_max(y64, si, dim_size=nseg); (p64*gw.double()).sum().backward()
    rel=lambda a,r: float((a.double()-r).abs().max()/r.abs().max().clamp_min(1e-30))
    print('B',B,'out', rel(o1,p64.detach()), rel(o0,p64.detach()), rel(of,p64.detach()))
    for (n,p),a,b,c in zip(ref.named_parameters(),gp1,gp0,gpf):
        print(f"  {n:22s} st1 {rel(a,p.grad):.4f} st0 {rel(b,p.grad):.4f} f32 {rel(c,p.grad):.6f}  |ref|max {float(p.grad.abs().max()):.3e}")
