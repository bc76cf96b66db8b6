"""Which Python call sites still reach the vendor GEMM in one train step (shapes + a short stack)?
    python tools/diag_vendor_gemm.py [--model spt128] [--scene T]"""
import argparse
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import hotpath  # noqa: E402
from superpoint_transformer_amd.synthetic import make_nag  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="spt128")
ap.add_argument("--scene", default="T")
a = ap.parse_args()
dev = torch.device("cuda:0")
seen = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "superpoint_transformer_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*args, **kw):
        shapes = tuple(tuple(t.shape) for t in args if torch.is_tensor(t))
        if any(len(s) >= 2 for s in shapes):
            seen[(name, shapes, where())] += 1
        return orig(*args, **kw)
    setattr(mod, name, f)


for mod, name in ((torch.nn.functional, "linear"), (torch, "matmul"), (torch, "bmm"), (torch, "mm"), (torch, "addmm"),
                  (torch.Tensor, "__matmul__"), (torch.Tensor, "matmul"), (torch.Tensor, "mm")):
    wrap(mod, name)
nag = make_nag(a.scene, seed=1234, device=dev)
path = hotpath.build(nag, dev, model=a.model)
path.step()
seen.clear()
path.step()
torch.cuda.synchronize()
for (name, shapes, w), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, name, shapes, w)
