"""Launches and GPU time PER STEP from a kernel-stats CSV of a train-step capture (rocpd_summary.py):
the step count comes from the cross-entropy forward's calls (2 per step).
    python tools/launch_count.py profiles/<tag>_spt64_trainstep_sceneT_kernel_stats.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = None
for r in rows:                                  # two cross-entropy forwards per step (levels 1, 2)
    if "ce_fwd_kernel" in r["Name"]:
        steps = int(r["Calls"]) / 2
        break
if not steps:
    sys.exit("no cross-entropy kernel in the capture")
calls = sum(int(r["Calls"]) for r in rows) / steps
us = sum(float(r["TotalDurationUs"]) for r in rows) / steps
small = [r for r in rows if float(r["AverageUs"]) < 30]
print(f"{sys.argv[1]}: {steps:.0f} steps, {calls:.1f} launches / step, {us / 1e3:.3f} ms GPU / step; "
      f"launches under 30 us: {sum(int(r['Calls']) for r in small) / steps:.1f} / step = "
      f"{sum(float(r['TotalDurationUs']) for r in small) / steps / 1e3:.3f} ms")
torch_rows = [r for r in rows if r["Name"].startswith(("void at::", "at::", "__amd_rocclr", "void (anonymous"))]
print(f"  torch / runtime kernels: {sum(int(r['Calls']) for r in torch_rows) / steps:.1f} launches / step")
