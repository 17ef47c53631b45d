#!/usr/bin/env python3
"""profiles/rocprof_kernel_avg_<workload>.json (read by bench.py: roofline.rocprofv3_avg_launch_us / frac_by_rocprofv3) from the
rocprofv3 --kernel-trace --stats summaries tools/gpu_final_profiles.sh wrote (tools/rocpd_stats.py output of
`rocprofv3 --kernel-trace --stats -- python bench.py --workload <wl> --steps 3 --warmup 1`) and the bench line printed under the
profiler (its T).  Usage: make_rocprof_avg_json.py [round prefix, default: newest under profiles/]"""
import datetime, glob, json, os, re, sys

pre = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("profiles/r*_final_rocprofv3_kernel_stats_*.txt"))[-1].split("/")[-1].split("_final_")[0]
for f in sorted(glob.glob(f"profiles/{pre}_final_rocprofv3_kernel_stats_*.txt")):
    wl = f.split("_kernel_stats_")[1][:-4]
    T = None
    bj = f"profiles/{pre}_final_bench_under_rocprof_{wl}.json"
    if os.path.exists(bj):
        try:
            T = json.loads(open(bj).read().strip().splitlines()[-1])["config"]["T"]
        except Exception:
            pass
    kernels = {}
    for line in open(f):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(?:void )?pi::(?:r3d::|s1::)?(\w+)", line)
        if m:
            calls, tot, avg, mn, mx, pct, kern = m.groups()
            if kern not in kernels:                          # (first = the flavour with the most time)
                kernels[kern] = {"calls": int(calls), "avg_us": float(avg), "min_us": float(mn), "max_us": float(mx)}
    out = {"workload": wl, "T": T, "date": datetime.date.fromtimestamp(os.path.getmtime(f)).isoformat(),
           "source": f"{f} (rocprofv3 --kernel-trace --stats -- python bench.py --workload {wl} --no-cpu-baseline --no-extras --no-also "
                     "--steps 3 --warmup 1; average over the launches of that run)", "kernels": kernels}
    json.dump(out, open(f"profiles/rocprof_kernel_avg_{wl}.json", "w"), indent=1)
    print(wl, T, {k: v["avg_us"] for k, v in list(kernels.items())[:4]})
