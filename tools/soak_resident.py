#!/usr/bin/env python3
"""Soak of the resident launches added in round 5 (float64 forward / sweep, small-tile forward, small sweep with its pause): random
grids, horizons, frame masks and parameter blocks; every trajectory and dL/dh0 must equal the launch-per-group path bit for bit,
parameter gradients to summation round-off, and no launch may abort.  usage: soak_resident.py [iterations] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import percnn_amd as pa
from percnn_amd import _lib
from util import random_block

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
big = [(512, 512), (384, 384), (384, 512), (512, 256), (448, 448), (320, 512), (288, 512), (512, 288)]
small = [(100, 100), (128, 96), (200, 40), (64, 64), (256, 256), (96, 160), (40, 104), (72, 64), (64, 32), (248, 252)]
a0 = _lib.persist_status()
t0 = time.time()
counts = {}
for it in range(n_it):
    kind = rs.choice(["f64_big", "f32_small", "f32_big"], p=[0.45, 0.4, 0.15])
    shape = (big if kind != "f32_small" else small)[rs.randint(len(big if kind != "f32_small" else small))]
    dtype = np.float64 if kind == "f64_big" else np.float32
    tdt = torch.float64 if kind == "f64_big" else torch.float32
    T = int(rs.randint(32, 90))
    P = torch.tensor(random_block(0, 2, dtype, int(rs.randint(10000)), scale=0.1), device=dev)
    traj = torch.full((T + 1, 2) + shape, float("nan"), dtype=tdt, device=dev)
    traj[0] = torch.tensor(rs.uniform(0, 1, (2,) + shape).astype(dtype), device=dev)
    ref = traj.clone()
    pa.rollout_fwd_(traj, P)
    pa.rollout_fwd_(ref, P, options={"fwd_persist": 0})
    assert torch.equal(traj.view(torch.int64 if tdt == torch.float64 else torch.int32), ref.view(torch.int64 if tdt == torch.float64 else torch.int32)), (it, kind, shape, T)
    g = torch.randn(traj.shape, dtype=tdt, device=dev, generator=torch.Generator(device=dev).manual_seed(it)) / traj[0].numel()
    mk = rs.randint(3)
    mask = None if mk == 0 else [bool(rs.rand() < 0.5) for _ in range(T + 1)] if mk == 1 else [t == T or t % int(rs.randint(2, 7)) == 0 for t in range(T + 1)]
    a, ag = pa.rollout_bwd(traj, g, P, frame_mask=mask)
    b, bg = pa.rollout_bwd(traj, g, P, frame_mask=mask, options={"tile_persist": 0, "persist_small": 0})
    assert torch.equal(a.view(torch.int64 if tdt == torch.float64 else torch.int32), b.view(torch.int64 if tdt == torch.float64 else torch.int32)), (it, kind, shape, T, mk)
    # (blown-up trajectories -- random blocks do that -- overflow the cubic moments in EVERY path: compare what is finite)
    if bool(torch.isfinite(traj[-1]).all()) and float(traj.abs().max()) < 1e6 and bool(torch.isfinite(bg).all()) and float(bg.norm()) > 0:
        err = float((ag - bg).norm() / bg.norm())
        assert err < (1e-11 if tdt == torch.float64 else 5e-6), (it, kind, shape, T, mk, err)
    counts[kind] = counts.get(kind, 0) + 1
    del traj, ref, g
a1 = _lib.persist_status()
print(f"{n_it} iterations in {time.time() - t0:.0f} s: {counts}; resident launches {a1['launches'] - a0['launches']}, aborts {a1['aborts'] - a0['aborts']}")
assert a1["aborts"] == a0["aborts"]
