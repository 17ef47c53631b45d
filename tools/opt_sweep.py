#!/usr/bin/env python3
"""Time the rollout kernels of one grid under several per-call option sets (MI355X tuning aid).

    python tools/opt_sweep.py --family gs3d --shape 384 384 384 --T 8 --opts "" "l2_tile_kb=0" "l2_tile_kb=64"

Prints forward / backward microseconds per time step and the effective bandwidth on algorithmic bytes (16 / 32 B per
point and step in float32) for every option string; all variants compute the same values (bit-identical states)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="gs3d", choices=["gs2d", "gs3d", "lo2d"])
    ap.add_argument("--shape", type=int, nargs="+", required=True)
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=1, help="measure every option set once per round, report the median over rounds "
                                                          "(interleaving takes clock / order drifts out of A-B comparisons)")
    ap.add_argument("--reaction", default="poly")
    ap.add_argument("--opts", nargs="*", default=[""])
    ap.add_argument("--check", action="store_true", help="assert every variant reproduces the first one bit for bit")
    a = ap.parse_args()
    import percnn_amd as pa
    import bench
    dev = torch.device("cuda:0")
    wl = {"gs2d": "gs2d_512", "gs3d": "gs3d_128", "lo2d": "lo2d_512"}[a.family]
    family, _, hc, dtype, _, golden = bench.WORKLOADS[wl]
    cell = bench.make_cell(family, bench.load_params(golden), dev, a.reaction)
    with torch.no_grad():
        P = cell.param_block().contiguous()
    shape = tuple(a.shape)
    npts = int(np.prod(shape))
    traj = torch.empty((a.T + 1, 2) + shape, dtype=dtype, device=dev)
    traj[0] = bench.initial_state(family, shape)[0].to(dev)
    g = torch.randn(traj.shape, dtype=dtype, device=dev) * (2.0 / traj.numel())
    esz = dtype.itemsize
    ref = None
    res = {o: ([], []) for o in a.opts}
    for rnd in range(a.rounds):
        for o in a.opts:
            opt = o or None
            for _ in range(2 if rnd == 0 else 1):
                pa.rollout_fwd_(traj, P, options=opt)
                g0, pg = pa.rollout_bwd(traj, g, P, options=opt)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(a.reps):
                pa.rollout_fwd_(traj, P, options=opt)
            ev[1].record()
            for _ in range(a.reps):
                g0, pg = pa.rollout_bwd(traj, g, P, options=opt)
            ev[2].record()
            torch.cuda.synchronize()
            res[o][0].append(ev[0].elapsed_time(ev[1]) * 1e3 / (a.reps * a.T))
            res[o][1].append(ev[1].elapsed_time(ev[2]) * 1e3 / (a.reps * a.T))
            if a.check and rnd == 0:
                cur = (traj[-1].clone(), g0.clone())
                if ref is None:
                    ref = cur
                else:
                    assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), f"variant '{o}' differs"
    for o in a.opts:
        tf, tb = float(np.median(res[o][0])), float(np.median(res[o][1]))
        spread = f" (min {min(res[o][0]):.2f}/{min(res[o][1]):.2f})" if a.rounds > 1 else ""
        print(f"{family} {'x'.join(map(str, shape)):>13s} T={a.T:<4d} {o or '(defaults)':<34s} fwd {tf:9.2f} us {4 * esz * npts / tf / 1e3:7.0f} GB/s"
              f" | bwd {tb:9.2f} us {8 * esz * npts / tb / 1e3:7.0f} GB/s | {a.T and 1e6 / (tf + tb):9.0f} steps/s{spread}", flush=True)


if __name__ == "__main__":
    main()
