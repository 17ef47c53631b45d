#!/usr/bin/env python3
"""backward-only timing of the small-tile resident sweep vs the pause between publish and first ring request"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, percnn_amd as pa
dev = torch.device("cuda:0")
for shape, T in (((100, 100), 200), ((256, 256), 200), ((300, 320), 200)):
    sd = bench.load_params(bench.WORKLOADS["gs2d_512"][5])
    cell = bench.make_cell("gs2d", sd, dev, "poly")
    with torch.no_grad():
        P = cell.param_block().contiguous()
    traj = torch.empty((T + 1, 2) + shape, device=dev)
    traj[0] = bench.initial_state("gs2d", shape)[0].to(dev)
    pa.rollout_fwd_(traj, P)
    g = torch.randn(traj.shape, device=dev) / traj.numel()
    res = {}
    opts = [{"persist_small": 0}] + [{"adj_small_pause": v} for v in (0, 4, 8, 12, 16, 24, 32)]
    for rnd in range(5):
        for o in opts:
            for _ in range(2):
                pa.rollout_bwd(traj, g, P, options=o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                pa.rollout_bwd(traj, g, P, options=o)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(str(o), []).append(e0.elapsed_time(e1) * 1e3 / (10 * T))
    print(shape, " | ".join(f"{k}: {np.median(v):.3f}" for k, v in res.items()), flush=True)
