"""Average PMC counter values per kernel from a rocprofv3 rocpd database.

    python tools/pmc_query.py <dir with *.db> "%kernel pattern%"

Prints, per (kernel, counter): the mean over all launches, the maximum, and the mean over the
UPPER GROUP of launches (value above the midpoint of min and max) - a kernel launched on two
graph levels per step (the attention kernels: level 1 and level 2) separates into its level-1
calls (`hi`) and the rest that way; for a kernel with one launch shape the two means agree."""
import glob
import sqlite3
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "%"
files = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not files:
    print("no .db under", sys.argv[1])
for f in files:
    c = sqlite3.connect(f)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kname = next((x for x in ("kernel_name", "name", "kernel") if x in cols), None)
    cname = next((x for x in ("counter_name", "pmc_name", "counter") if x in cols), None)
    vname = next((x for x in ("value", "counter_value") if x in cols), None)
    if not (kname and cname and vname):
        print("counters_collection columns:", cols)
        continue
    q = (f"select substr({kname},1,60), {cname}, {vname} from counters_collection "
         f"where {kname} like ?")
    groups = {}
    for k, cn, v in c.execute(q, (pat,)):
        groups.setdefault((k, cn), []).append(float(v))
    for (k, cn), vals in sorted(groups.items()):
        lo, hi = min(vals), max(vals)
        top = [v for v in vals if v >= 0.5 * (lo + hi)]
        print(f"{k[:48]:48s} {cn:28s} avg {sum(vals) / len(vals):16.1f}  max {hi:16.1f}  n={len(vals)}"
              f"  hi {sum(top) / len(top):16.1f}  nhi={len(top)}")
