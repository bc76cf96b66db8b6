"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) capture as CSV.

    python tools/rocpd_summary.py gpurun_out/<dir>/<name>_results.db > profiles/<name>_kernel_stats.csv

Equivalent to rocprofv3's `--stats` kernel table (name, calls, total/avg/min/max
duration in microseconds, share of GPU time) for captures written in the
default rocpd format.
"""
import csv
import glob
import os
import sqlite3
import sys


def main(path):
    if os.path.isdir(path):                     # a rocprofv3 -d directory: take its database
        dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
        if not dbs:
            sys.exit(f"no .db under {path}")
        path = dbs[0]
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, "
        "min(end-start)/1e3, max(end-start)/1e3 from kernels group by name "
        "order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
    for name, n, t, a, mn, mx in rows:
        short = name if len(name) < 200 else name[:197] + "..."
        w.writerow([short, n, f"{t:.3f}", f"{a:.3f}", f"{mn:.3f}", f"{mx:.3f}", f"{100 * t / tot:.2f}"])


if __name__ == "__main__":
    main(sys.argv[1])
