#!/usr/bin/env python3
"""Dump a window of a rocprofv3 (rocpd sqlite) kernel trace as a timeline: start / end (us since the window's first kernel),
queue, name.  Usage: rocpd_timeline.py results.db [skip=2000] [count=60] [substring]"""
import sqlite3
import sys

db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
qcol = next((x for x in ("queue_id", "queue", "stream_id", "stream") if x in cols), None)
rows = c.execute(f"select {name_col}, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
rows = rows[skip:skip + count]
t0 = rows[0][1]
print(f"# columns of `kernels`: {cols}")
for r in rows:
    n, s, e = r[0], r[1], r[2]
    q = r[3] if qcol else "-"
    short = n.split("(")[0].replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  q={q}  {short}")
