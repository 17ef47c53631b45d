"""Effective bandwidth of the three rollout kernels (forward, adjoint sweep, gradient reduction) as a
function of grid size -- separates "kernel quality" from "the 512^2 problem is too small to hide latency".
Run on the GPU box:  python tools/size_sweep.py [--out gpurun_out/size_sweep.json]
"""
import argparse, json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import percnn_amd as pa                                   # noqa: E402
from bench import load_params, make_cell                  # noqa: E402

CASES = [  # family, golden, dtype, shapes
    ("gs2d", "gs2d_big_512x512.npz", torch.float32, [(128, 128), (256, 256), (512, 512), (1024, 1024), (2048, 2048), (4096, 4096)]),
    ("lo2d", "lo2d_big_512x512.npz", torch.float64, [(512, 512), (2048, 2048)]),
    ("gs3d", "gs3d_big_128x128x128.npz", torch.float32, [(64, 64, 64), (128, 128, 128), (256, 256, 256), (384, 384, 384)]),
]


def measure(family, golden, dtype, shape, reaction, dev, budget_bytes=24e9, reps=3):
    cell = make_cell(family, load_params(golden), dev, reaction)
    with torch.no_grad():
        P = cell.param_block().contiguous()
    npts = int(np.prod(shape)); esz = dtype.itemsize
    frame = 2 * npts * esz
    T = int(max(8, min(400, budget_bytes / (3.2 * frame))) // 8 * 8)
    traj = torch.empty((T + 1, 2) + shape, dtype=dtype, device=dev)
    traj[0] = torch.rand((2,) + shape, dtype=dtype, device=dev) * 0.1 + 0.45
    g = torch.randn_like(traj) * 1e-6
    pa.rollout_fwd_(traj, P); pa.rollout_bwd(traj, g, P); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for _ in range(reps): pa.rollout_fwd_(traj, P)
    ev[1].record()
    for _ in range(reps): pa.rollout_bwd(traj, g, P)
    ev[2].record()
    pa.set_option("skip_wgrad", 1)
    for _ in range(reps): pa.rollout_bwd(traj, g, P)
    ev[3].record(); torch.cuda.synchronize()
    pa.set_option("skip_wgrad", 0)
    assert torch.isfinite(traj[-1]).all()
    fwd = ev[0].elapsed_time(ev[1]) / reps / T * 1e-3
    bwd = ev[1].elapsed_time(ev[2]) / reps / T * 1e-3
    swp = ev[2].elapsed_time(ev[3]) / reps / T * 1e-3
    # float32 poly mode on the direct / plane-streaming kernels reduces the gradients INSIDE the sweep launches
    # (fuse_wgrad): there the sweep-only time (skip_wgrad diagnostic) is only a lower bound and no separate pass exists
    fused = (bwd - swp) < 0.25 * swp and reaction == "poly" and dtype == torch.float32 and (len(shape) == 3 or npts >= (1 << 20))
    red = max(bwd - swp, 1e-12)
    Cs = 2 * esz
    hc = cell.hidden_channels
    red_b = 2 * Cs if (reaction == "poly" or hc <= 4) else 3 * Cs
    r = {"family": family, "shape": list(shape), "dtype": str(dtype)[6:], "reaction": reaction, "T": T,
         "fwd_us_step": fwd * 1e6, "sweep_us_step": swp * 1e6, "reduce_us_step": red * 1e6,
         "fwd_GBs": 2 * Cs * npts / fwd / 1e9, "sweep_GBs": 4 * Cs * npts / swp / 1e9,
         "reduce_GBs": None if fused else red_b * npts / red / 1e9, "fused_reduction": bool(fused),
         "bwd_us_step": bwd * 1e6, "steps_per_s": 1.0 / (fwd + bwd),
         "Mpts_steps_per_s": npts / (fwd + bwd) / 1e6}
    del traj, g
    torch.cuda.empty_cache()
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/size_sweep.json")
    ap.add_argument("--reactions", default="poly,factored")
    ap.add_argument("--family", default="")
    ap.add_argument("--min-points", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    for kv in a.opt:
        k, v = kv.split("=")
        pa.set_option(k, int(v))
    dev = torch.device("cuda:0")
    rows = []
    for family, golden, dtype, shapes in CASES:
        for reaction in a.reactions.split(","):
            for shape in shapes:
                if (a.family and family != a.family) or int(np.prod(shape)) < a.min_points:
                    continue
                r = measure(family, golden, dtype, shape, reaction, dev)
                rows.append(r)
                tail = ("sweep+reduction fused %8.2f us" % r["bwd_us_step"]) if r["fused_reduction"] else \
                    ("sweep %8.2f us %5.0f GB/s | reduce %7.2f us %5.0f GB/s" % (r["sweep_us_step"], r["sweep_GBs"],
                                                                               r["reduce_us_step"], r["reduce_GBs"]))
                print("%-5s %-9s %-14s T=%3d  fwd %8.2f us %5.0f GB/s | %s | %8.1f Mpt-steps/s"
                      % (family, reaction, "x".join(map(str, shape)), r["T"], r["fwd_us_step"], r["fwd_GBs"], tail,
                         r["Mpts_steps_per_s"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
