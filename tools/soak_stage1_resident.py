#!/usr/bin/env python3
"""Soak of the resident Stage-1 rollouts (round 5): random grids of whole 4x4 patches, horizons, frame masks, parameter blocks
and both families; trajectory, dL/dh0 and the parameter gradients must equal the launch-per-step path bit for bit, and no launch
may abort.  In between, 2D resident rollouts on the same stream (they share the residency guard and the outbox scratch).
usage: soak_stage1_resident.py [iterations] [seed]"""
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
a0 = _lib.persist_status()
t0 = time.time()
n_res = 0
n_finite = 0
for it in range(n_it):
    fam = ["burgers", "lo"][rs.randint(2)]
    H, W = 4 * int(rs.randint(2, 33)), 4 * int(rs.randint(2, 33))
    T = int(rs.randint(8, 70))
    torch.manual_seed(it)
    cell = pa.Stage1Cell(fam).to(dev)
    with torch.no_grad():
        for p in cell.parameters():
            if p.requires_grad and p.dim() > 0:
                p.mul_(float(rs.uniform(0.3, 0.8)))
        P = cell.param_block().contiguous()
    traj = torch.full((T + 1, 2, H, W), float("nan"), device=dev)
    traj[0] = torch.tensor(rs.uniform(-0.5, 0.5, (2, H, W)).astype(np.float32), device=dev)
    ref = traj.clone()
    n0 = _lib.persist_status()["launches"]
    pa.stage1.rollout_fwd_(traj, P)
    n_res += _lib.persist_status()["launches"] - n0
    pa.stage1.set_option("persist", 0)
    pa.stage1.rollout_fwd_(ref, P)
    pa.stage1.set_option("persist", 1)
    assert torch.equal(traj.view(torch.int32), ref.view(torch.int32)), (it, fam, H, W, T)
    n_finite += int(bool(torch.isfinite(traj).all()) and float((traj[-1] - traj[0]).abs().max()) > 0)
    g = torch.randn(traj.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(it)) / traj[0].numel()
    mk = rs.randint(3)
    mask = None if mk == 0 else [bool(rs.rand() < 0.5) for _ in range(T + 1)] if mk == 1 else [t == T or t % int(rs.randint(2, 7)) == 0 for t in range(T + 1)]
    n0 = _lib.persist_status()["launches"]
    a, ag = pa.stage1.rollout_bwd(traj, g, P, frame_mask=mask)
    n_res += _lib.persist_status()["launches"] - n0
    pa.stage1.set_option("persist", 0)
    b, bg = pa.stage1.rollout_bwd(traj, g, P, frame_mask=mask)
    pa.stage1.set_option("persist", 1)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (it, fam, H, W, T, mk)
    assert torch.equal(ag.view(torch.int64), bg.view(torch.int64)), (it, fam, H, W, T, mk)
    if it % 5 == 0:                                         # a 2D resident rollout in between (same guard, same scratch)
        shape = [(100, 100), (384, 384), (64, 96)][rs.randint(3)]
        P2 = torch.tensor(random_block(0, 2, np.float32, int(rs.randint(10000)), scale=0.1), device=dev)
        t2 = torch.empty((41, 2) + shape, device=dev)
        t2[0] = torch.rand((2,) + shape, device=dev)
        r2 = t2.clone()
        pa.rollout_fwd_(t2, P2)
        pa.rollout_fwd_(r2, P2, options={"fwd_persist": 0})
        assert torch.equal(t2.view(torch.int32), r2.view(torch.int32)), (it, shape)
    del traj, ref, g
a1 = _lib.persist_status()
print(f"{n_it} iterations in {time.time() - t0:.0f} s; Stage-1 resident launches {n_res} of {2 * n_it} rollouts "
      f"(the rest: more tasks than the device holds), {n_finite} finite and moving trajectories, all launches {a1['launches'] - a0['launches']}, aborts {a1['aborts'] - a0['aborts']}")
assert a1["aborts"] == a0["aborts"]
