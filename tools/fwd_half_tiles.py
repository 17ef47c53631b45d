#!/usr/bin/env python3
"""512^2 forward: the 32 x 32 resident kernel vs several small-tile resident workgroups per CU (phase overlap by co-residency)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, percnn_amd as pa
from percnn_amd import _lib
dev = torch.device("cuda:0")
for shape, T in (((512, 512), 400), ((384, 384), 400), ((640, 640), 200)):
    sd = bench.load_params(bench.WORKLOADS["gs2d_512"][5])
    cell = bench.make_cell("gs2d", sd, dev, "poly")
    with torch.no_grad():
        P = cell.param_block().contiguous()
    traj = torch.empty((T + 1, 2) + shape, device=dev)
    traj[0] = bench.initial_state("gs2d", shape)[0].to(dev)
    ref = None
    res = {}
    opts = [{}, {"fwd_persist": 0}, {"tile_by": 16, "persist_small": 2, "fwd_persist_per_cu": 1}, {"tile_by": 16, "persist_small": 2, "fwd_persist_per_cu": 2},
            {"tile_by": 8, "persist_small": 2, "fwd_persist_per_cu": 2}, {"tile_by": 16, "fwd_persist": 0}, {"tile_by": 8, "fwd_persist": 0}]
    for rnd in range(3):
        for o in opts:
            for _ in range(2):
                pa.rollout_fwd_(traj, P, options=o)
            if rnd == 0:
                cur = traj[-1].clone()
                if ref is None:
                    ref = cur
                assert torch.equal(cur, ref), o
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                pa.rollout_fwd_(traj, P, options=o)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(str(o), []).append(e0.elapsed_time(e1) * 1e3 / (5 * T))
    for k, v in res.items():
        print(shape, f"{np.median(v):.3f} us/step", k, _lib.rollout_plan(0, shape, 4, ",".join(f"{a}={b}" for a, b in eval(k).items()) or None)["fwd_persistent"], flush=True)
    print(_lib.persist_status(), flush=True)
