#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max,
plus the inter-kernel gap statistics on the busiest stream.  Usage: rocpd_stats.py results.db [...]"""
import sqlite3
import sys


def summarise(db):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for n, s, e in rows:
        d = e - s
        st = stats.setdefault(n, [0, 0, 1 << 62, 0])
        st[0] += 1; st[1] += d; st[2] = min(st[2], d); st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    print(f"# {db}: {len(rows)} dispatches, {total/1e6:.3f} ms of kernel time")
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for n, (k, t, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:8d} {t/1e6:10.3f} {t/k/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*t/total:6.1f}  {n[:110]}")
    # gaps between consecutive dispatches of the two step kernels
    for key in ("pi_fwd", "pi_bwd", "pi_adj", "pi_wgrad", "s1_fwd", "s1_adj"):
        seq = [(s, e) for n, s, e in rows if key in n]
        if len(seq) > 10:
            gaps = sorted(seq[i + 1][0] - seq[i][1] for i in range(len(seq) - 1))
            print(f"  gap between consecutive {key}* dispatches: median {gaps[len(gaps)//2]/1e3:.2f} us, "
                  f"p10 {gaps[len(gaps)//10]/1e3:.2f} us, p90 {gaps[9*len(gaps)//10]/1e3:.2f} us")


if __name__ == "__main__":
    for db in sys.argv[1:]:
        summarise(db)
        print()
