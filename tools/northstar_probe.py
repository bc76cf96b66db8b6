import torch, sys, os
sys.path.insert(0, os.getcwd())
from superpoint_transformer_amd import ops
from superpoint_transformer_amd.synthetic import make_nag
dev = torch.device("cuda:0")
nag = make_nag("S", seed=0, device=dev)
si = nag.levels[0]["super_index"]; n0 = si.numel(); n1 = nag.levels[1]["pos"].shape[0]
x = torch.randn(n0, 128, device=dev)
for _ in range(3): ops.segment_reduce(x, si, n1, "max", return_arg=True)
torch.cuda.synchronize()
for trial in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.segment_reduce(x, si, n1, "max", return_arg=True)
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 20
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.segment_reduce(x, si, n1, "max", return_arg=True); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) for a, b in evs]
    import time
    t0 = time.perf_counter()
    for _ in range(20): ops.segment_reduce(x, si, n1, "max", return_arg=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"bracket {tot:.4f} ms/launch; per-launch events mean {sum(per)/20:.4f} min {min(per):.4f} max {max(per):.4f}; host enqueue {1e3*(t1-t0)/20:.3f} ms/launch, wall {1e3*(t2-t0)/20:.3f}")
