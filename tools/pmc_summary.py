#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (run on the GPU box)."""
import sqlite3
import sys

db, label = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
name = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
q = (f"select {name}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
     f"group by {name}, counter_name order by sum(value) desc limit 16")
for r in c.execute(q):
    print(f"{label} | {r[1]} | n={r[2]} avg={r[3]:.1f} min={r[4]:.1f} max={r[5]:.1f} | {r[0][:90]}")
