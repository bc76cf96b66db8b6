"""Registers / scratch / LDS of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiled for gfx950: no GPU needed).  usage: python tools/kernel_resources.py csrc/fused_pool.hip [filter]"""
import re
import subprocess
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superpoint_transformer_amd import build as B

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
import os
extra = B.PER_FILE_FLAGS.get(os.path.basename(src), [])
cmd = [B.HIPCC] + B.FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for name, (k, v) in zip(names, rows.items()):
    name = re.sub(r"\(.*", "", name).replace("spt::", "")
    if flt in name:
        print(f"{name:70s} vgpr {v.get('vgpr')} agpr {v.get('agpr')} scratch {v.get('scratch')} lds {v.get('lds')} occ {v.get('occ')}")
