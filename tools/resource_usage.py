"""Per-kernel register / scratch / LDS table from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: python tools/resource_usage.py remarks.txt [substring ...]   (demangles with c++filt)"""
import re, subprocess, sys

def main():
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    rows = []
    for blk in txt.split("Function Name: ")[1:]:
        name = blk.split(" ", 1)[0].split("\n", 1)[0].rstrip("[-Rpass-analysis=kernel-resource-usage]").strip()
        get = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
        rows.append((name, get("VGPRs"), get("AGPRs"), get("SGPRs"), get(r"ScratchSize \[bytes/lane\]"), get("Occupancy \[waves/SIMD\]"),
                     get(r"LDS Size \[bytes/block\]")))
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True,
                         text=True).stdout.splitlines()
    print("vgpr agpr sgpr scratch occ lds  kernel")
    for r, d in zip(rows, dem):
        d = re.sub(r"^void pi::", "", d)
        if pats and not any(p in d for p in pats):
            continue
        print("%4s %4s %4s %5s %3s %5s  %s" % (r[1], r[2], r[3], r[4], r[5], r[6], d[:150]))

if __name__ == "__main__":
    main()
