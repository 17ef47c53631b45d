#!/usr/bin/env python3
"""Append per-kernel averages of every counter in a rocprofv3 rocpd database to a JSON-lines file (run on the GPU box).

    pmc_dump.py results.db "<label>" out.jsonl

One record per (kernel, counter): per-DISPATCH value (a counter reported per XCC / SE instance is summed over its instances
first), averaged over the dispatches of that kernel, plus the kernel's average duration from the same database."""
import json
import sqlite3
import sys

db, label, out = sys.argv[1], sys.argv[2], sys.argv[3]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
name = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x and "counter" not in x][0]
disp = "dispatch_id" if "dispatch_id" in cols else None
dur = {}
try:
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    kname = "name" if "name" in kcols else "kernel_name"
    for n, k, t in c.execute(f"select {kname}, count(*), avg(end - start) from kernels group by {kname}"):
        dur[n] = (k, t / 1e3)
except sqlite3.Error:
    pass
if disp:
    q = (f"select {name}, counter_name, count(*), avg(v), min(v), max(v) from (select {name}, counter_name, {disp}, sum(value) as v "
         f"from counters_collection group by {name}, counter_name, {disp}) group by {name}, counter_name")
else:
    q = f"select {name}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection group by {name}, counter_name"
with open(out, "a") as f:
    for kn, cn, n, avg, mn, mx in c.execute(q):
        rec = {"label": label, "kernel": kn, "counter": cn, "n": n, "avg": avg, "min": mn, "max": mx,
               "calls": dur.get(kn, (None, None))[0], "avg_us": dur.get(kn, (None, None))[1], "per_dispatch": bool(disp)}
        f.write(json.dumps(rec) + "\n")
print(f"{label}: {db} columns={cols}", file=sys.stderr)
