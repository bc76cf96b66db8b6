"""Top kernels of a rocprofv3 rocpd capture, names shortened (quick look on the GPU box)."""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(path)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1.0
for name, n, t, a, mn in rows[:top]:
    short = name.split("(")[0][-70:]
    print(f"{short:70s} calls {n:5d} total {t:10.1f} us  avg {a:9.1f}  min {mn:9.1f}  {100 * t / tot:5.1f} %")
