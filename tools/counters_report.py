#!/usr/bin/env python3
"""profiles/r05_counters_summary.txt: the round-5 counter campaign (tools/gpu_counters_r05.sh) in one readable file.
usage: counters_report.py sq_tcp_tcc.jsonl ea_sizes.jsonl after_fix.jsonl > profiles/r05_counters_summary.txt"""
import collections
import json
import re
import subprocess
import sys


def load(fn):
    by, us = collections.defaultdict(dict), {}
    for l in open(fn):
        r = json.loads(l)
        k = (r["label"], r["kernel"])
        by[k][r["counter"]] = r["avg"]
        if r.get("avg_us"):
            us[k] = r["avg_us"]
    return by, us


def short(kn):
    m = re.search(r"pi::(\w+)(<[^(]*>)?", kn)
    if m:
        return m.group(1) + (m.group(2) or "")
    return "torch elementwise (x = randn * c: 1 read, 1 write per element)" if "vectorized_elementwise" in kn else None


def ea_table(fn, title):
    by, us = load(fn)
    print(title)
    print(f"{'workload':<16} {'kernel':<58} {'us':>8} | {'EA reads':>9} {'128 B':>9} {'64 B':>7} {'read MB':>8} | {'EA writes':>9} {'write MB':>8} | "
          f"{'total MB':>8} {'TB/s':>5} | {'rd latency':>10} | {'L2 hit %':>8}")
    for (wl, kn), d in sorted(by.items()):
        s = short(kn)
        if not s or (not s.startswith(("pi_fwd3d", "pi_adj3d", "pi_stream3d", "torch")) or us.get((wl, kn), 0) < 5):
            continue
        g = lambda k: d.get(k, 0.0) or 0.0
        rd, r128, r64 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_128B_sum"), g("TCC_EA0_RDREQ_64B_sum")
        if not rd:
            continue
        rb = (r128 * 128 + r64 * 64 + max(0.0, rd - r128 - r64) * 32) / 1e6
        wr = g("TCC_EA0_WRREQ_sum")
        wb = wr * 64 / 1e6
        lat = g("TCC_EA0_RDREQ_LEVEL_sum") / rd if g("TCC_EA0_RDREQ_LEVEL_sum") else float("nan")
        hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        hr = 100.0 * hit / (hit + miss) if hit + miss else float("nan")
        u = us[(wl, kn)]
        print(f"{wl:<16} {s[:58]:<58} {u:8.2f} | {rd:9.0f} {r128:9.0f} {r64:7.0f} {rb:8.1f} | {wr:9.0f} {wb:8.1f} | {rb + wb:8.1f} {(rb + wb) / u:5.2f} | "
              f"{lat:8.0f} cy | {hr:8.1f}")
    print()


print("""# Round-5 counter campaign (VERDICT r4 next #3) -- rocprofv3 --kernel-trace --pmc, separate passes per counter group, names filtered
# against `rocprofv3 -L`; tools/gpu_counters_r05.sh + tools/pmc_dump.py; one MI355X, ROCm 7.2.  Per-LAUNCH averages.
# Workloads: gs2d_512 / lo2d_512 = bench.py --workload ... --T 100 (25 groups of 4 steps per resident launch);
#            gs3d_* = tools/opt_sweep.py --family gs3d --shape ... (one launch per time step).
# Raw records: profiles/r05_counters_records_*.jsonl.
#
# READING IT
#  * SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves (MI355X_MICROARCH.md); the three shares
#    issue = SQ_ACTIVE_INST_ANY, stall = SQ_WAIT_INST_ANY (issue stalled: dependency / pipe busy), parked = SQ_WAIT_ANY
#    (s_waitcnt / s_barrier) are fractions of a resident wave's life and add up to ~1.
#  * L1acc = TCP_TOTAL_CACHE_ACCESSES (64-byte units), L1->L2 = TCP_TCC_READ/WRITE_REQ, EA = memory-side (fabric) requests of
#    the L2s.  The second table resolves their SIZE (TCC_EA0_RDREQ_128B / _64B): streaming 16-byte-per-lane reads go out as
#    128-byte requests, which is why FETCH_SIZE (= RDREQ x 64 B) reports half of the bytes on gfx950.
#
# WHAT IT SAYS (DESIGN.md section 4 cites these lines)
#  1. No kernel on the metric's configs is VALU-bound: the vector ALU is active in 10-21 % of a wave's life.  The 2D resident
#     kernels sit PARKED 44 % (sweep) / 64 % (forward) of the time -- barriers and granule waits, one to two waves per SIMD; the
#     forward's stall share is small (0.11): it waits, it does not compete for issue.  The 3D brick kernels split evenly between
#     issue (0.24-0.28), issue stalls (0.36) and waits (0.36-0.42).
#  2. The 3D brick kernels execute MORE scalar than vector instructions (forward 128^3: 1.83 M SALU vs 1.53 M VALU per launch,
#     448 vs 373 per wave): plane addresses, periodic wraps and halo tasks are wave-uniform integer work.  At one instruction per
#     wave and four cycles that is ~0.9 us of serial prologue before the first load of a wave is requested.
#  3. LDS bank conflicts: 2D tile kernels 3-4.4 conflict cycles per active LDS cycle (the 4-point strips read 16-byte rows at a
#     132-byte pitch), 3D bricks 0.2-0.5.  LDS is active 1.5-3.5 % of a wave's life: not the limiter anywhere.
#  4. 256^3 (HBM regime): the memory-side READS were 1.44x (adjoint bricks, 579 MB per launch for 403 MB algorithmic) and 1.48x
#     (plane-streaming forward, 199 MB for 134 MB) the algorithmic ones, the writes exactly algorithmic.  All of the adjoint's
#     excess is on the stencil-read field (2.3x): each XCD's region streamed 2 MB of new lines per plane group through its 4 MB L2,
#     evicting the plane neighbours the next group needs.  Fixed in round 5 (XCD regions sized by the L2, pi_abi.hip
#     make_brick_geom): third table -- 431 MB (1.07x), 140 -> 131 us.  Memory-side traffic then: 566 MB in 126 us = 4.5 TB/s
#     against 5.7-6.5 TB/s for a 1-read-1-write elementwise kernel on the same buffers: the sweep is no longer
#     bandwidth-bound but latency x occupancy-bound (216 registers, 2 waves per SIMD; profiles/r05_brick_skeleton_vs_product.txt).
#  5. 128^3 / 32 x 256^2: reads 1.07-1.19x algorithmic, L2 hit rate 43-46 %; the launch is one resident round of waves in
#     phase lock (address prologue -> 14-16 loads per lane through a 64 B/clk L1 -> LDS -> barrier -> arithmetic -> stores).
""")
by, us = load(sys.argv[1])
print("## 1. SQ / TCP / TCC counters at the BASELINE sizes (before the round-5 changes)")
sys.stdout.flush()
subprocess.run([sys.executable, __file__.replace("counters_report.py", "counters_table.py"), sys.argv[1]])
print()
ea_table(sys.argv[2], "## 2. Memory-side request sizes, 3D kernels, BEFORE the L2-sized XCD regions (us under the profiler)")
ea_table(sys.argv[3], "## 3. 256^3 AFTER the L2-sized XCD regions")
