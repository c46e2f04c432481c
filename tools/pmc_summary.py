#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd database.  usage: pmc_summary.py results.db [name-filter]"""
import sqlite3
import sys

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
name_col = "kernel_name" if "kernel_name" in cols else "name"
q = f"select {name_col}, counter_name, sum(value), count(*) from counters_collection where {name_col} like ? group by {name_col}, counter_name"
out = {}
for n, cn, v, k in c.execute(q, (f"%{flt}%",)):
    out.setdefault(n, {})[cn] = (v, k)
for n, d in sorted(out.items(), key=lambda kv: -sum(v for v, _ in kv[1].values())):
    print(n[:110])
    for cn, (v, k) in sorted(d.items()):
        print(f"    {cn:32s} {v:18.0f}   ({k} samples)")
