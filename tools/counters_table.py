#!/usr/bin/env python3
"""profiles/r05_counters_summary.txt from tools/gpu_counters_r05.sh's records: per (workload, kernel) the raw per-launch counter
averages and the derived shares DESIGN.md section 4's "bound" column cites.

SQ_* cycle counters count quad-cycles summed over waves (MI355X_MICROARCH.md): shares are taken against SQ_WAVE_CYCLES, so they
read "fraction of a resident wave's life": ACTIVE_INST_ANY (issuing), WAIT_INST_ANY (issue stalled: dependency / pipe), WAIT_ANY
(parked in s_waitcnt / s_barrier).  VALU-busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES per SIMD is approximated by
4 * SQ_ACTIVE_INST_VALU / (SQ_BUSY_CYCLES * simds_per_se...) -- not attempted: the instance count of SQ_BUSY_CYCLES is
implementation-specific; the per-wave shares and the instruction mix are reported instead."""
import collections
import json
import re
import sys

recs = [json.loads(l) for l in open(sys.argv[1])]
by = collections.defaultdict(dict)
meta = {}
for r in recs:
    k = (r["label"], r["kernel"])
    by[k][r["counter"]] = r["avg"]
    if r.get("avg_us"):
        m = meta.setdefault(k, {"us": [], "calls": r["calls"]})
        m["us"].append(r["avg_us"])


def short(kn):
    m = re.search(r"pi::(\w+)(<[^(]*>)?", kn)
    if m:
        return m.group(1) + (m.group(2) or "")
    m = re.search(r"(elementwise_kernel|vectorized_elementwise_kernel)", kn)
    return (m.group(1) + " (torch copy)") if m else kn[:60]


def g(d, k):
    return d.get(k)


def ratio(a, b):
    return None if a is None or not b else a / b


def fmt(x, p="{:.3f}"):
    return "   n/a" if x is None else p.format(x)


want = ("elementwise", "vectorized", "pi_fwd2d_persist", "pi_adj2d_persist", "pi_fwd3d_brick", "pi_adj3d_brick", "pi_stream3d", "pi_fwd2d_tile", "pi_adj2d_tile",
        "pi_moments", "pi_bwd_kernel", "pi_fwd_kernel")
print("# per-launch averages; SQ cycle counters in quad-cycles summed over waves; shares are of SQ_WAVE_CYCLES")
print(f"{'workload':<16} {'kernel':<66} {'us':>8} | {'issue':>6} {'stall':>6} {'parked':>6} | {'VALU/wv-cyc':>11} {'LDSact':>6} {'VMEMact':>7} | "
      f"{'VALU':>9} {'SALU':>9} {'LDS':>9} {'VMEMrd':>8} {'VMEMwr':>8} {'SMEM':>8} | {'bankconf/LDSact':>15} | {'L1acc':>9} {'L1->L2rd':>9} "
      f"{'L1->L2wr':>9} | {'L2req':>9} {'L2hit%':>6} {'EArd':>9} {'EAwr':>9} | {'FETCH MB':>9} {'WRITE MB':>9}")
for (wl, kn), d in sorted(by.items()):
    s = short(kn)
    if not s.startswith(want):
        continue
    us = sorted(meta.get((wl, kn), {}).get("us", [0]))[len(meta.get((wl, kn), {}).get("us", [0])) // 2]
    wc = g(d, "SQ_WAVE_CYCLES")
    hit, miss = g(d, "TCC_HIT_sum"), g(d, "TCC_MISS_sum")
    hr = None if hit is None or miss is None or hit + miss == 0 else 100.0 * hit / (hit + miss)
    fs, ws = g(d, "FETCH_SIZE"), g(d, "WRITE_SIZE")
    print(f"{wl:<16} {s[:66]:<66} {us:8.2f} | {fmt(ratio(g(d, 'SQ_ACTIVE_INST_ANY'), wc)):>6} {fmt(ratio(g(d, 'SQ_WAIT_INST_ANY'), wc)):>6} "
          f"{fmt(ratio(g(d, 'SQ_WAIT_ANY'), wc)):>6} | {fmt(ratio(g(d, 'SQ_ACTIVE_INST_VALU'), wc)):>11} {fmt(ratio(g(d, 'SQ_ACTIVE_INST_LDS'), wc)):>6} "
          f"{fmt(ratio(g(d, 'SQ_ACTIVE_INST_VMEM'), wc)):>7} | {fmt(g(d, 'SQ_INSTS_VALU'), '{:.0f}'):>9} {fmt(g(d, 'SQ_INSTS_SALU'), '{:.0f}'):>9} "
          f"{fmt(g(d, 'SQ_INSTS_LDS'), '{:.0f}'):>9} {fmt(g(d, 'SQ_INSTS_VMEM_RD'), '{:.0f}'):>8} {fmt(g(d, 'SQ_INSTS_VMEM_WR'), '{:.0f}'):>8} "
          f"{fmt(g(d, 'SQ_INSTS_SMEM'), '{:.0f}'):>8} | {fmt(ratio(g(d, 'SQ_LDS_BANK_CONFLICT'), g(d, 'SQ_ACTIVE_INST_LDS'))):>15} | "
          f"{fmt(g(d, 'TCP_TOTAL_CACHE_ACCESSES_sum'), '{:.0f}'):>9} {fmt(g(d, 'TCP_TCC_READ_REQ_sum'), '{:.0f}'):>9} "
          f"{fmt(g(d, 'TCP_TCC_WRITE_REQ_sum'), '{:.0f}'):>9} | {fmt(g(d, 'TCC_REQ_sum'), '{:.0f}'):>9} {fmt(hr, '{:.1f}'):>6} "
          f"{fmt(g(d, 'TCC_EA0_RDREQ_sum'), '{:.0f}'):>9} {fmt(g(d, 'TCC_EA0_WRREQ_sum'), '{:.0f}'):>9} | "
          f"{fmt(None if fs is None else fs * 1024 / 1e6, '{:.2f}'):>9} {fmt(None if ws is None else ws * 1024 / 1e6, '{:.2f}'):>9}")
print()
print("# every counter collected (per-launch average), for the record")
for (wl, kn), d in sorted(by.items()):
    s = short(kn)
    if not s.startswith(want):
        continue
    print(f"{wl} | {s}")
    for k in sorted(d):
        print(f"    {k:<36} {d[k]:18.1f}")
