"""cProfile of the host side of one train step at a launch-bound scene (R):
where the fixed per-step milliseconds go.  python tools/host_profile.py [scene]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import hotpath, synthetic  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "R"
dev = torch.device("cuda:0")
nag = synthetic.make_nag(scene, device=dev)
step = hotpath.build(nag, dev, 1, "all")
for _ in range(5):
    step.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
