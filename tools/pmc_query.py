"""Average PMC counter values per kernel from a rocprofv3 rocpd database."""
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
    q = (f"select {kname}, {cname}, avg({vname}), count(*), max({vname}) from counters_collection "
         f"where {kname} like ? group by substr({kname},1,60), {cname}")
    for r in c.execute(q, (pat,)):
        print(f"{r[0][:48]:48s} {r[1]:28s} avg {r[2]:16.1f}  max {r[4]:16.1f}  n={r[3]}")
