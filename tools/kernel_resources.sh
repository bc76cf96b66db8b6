#!/bin/bash
# per-kernel register / LDS / spill table of one .hip file:  tools/kernel_resources.sh <file.hip> [flags...]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -Rpass-analysis=kernel-resource-usage \
  -c "$f" -o /dev/null 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except Exception: pass
        cur = {"name": name}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    print("%-90s vgpr %4s agpr %4s spill %3s scratch %4s occ %2s lds %7s" % (r["name"][-90:], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
