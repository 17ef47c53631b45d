"""rocprofv3 target: slab rollouts of one 32 x 256^2 slab through the peer-mailbox transport (self) -- per-kernel times of
peer_put_kernel / peer_take_kernel next to the step kernels.   rocprofv3 --kernel-trace --stats -- python tools/slab_peer_profile.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import percnn_amd as pa
from percnn_amd import slab, synthetic
import bench

dev = torch.device("cuda:0")
sd = bench.load_params(bench.WORKLOADS["gs3d_128"][5])
cell = bench.make_cell("gs3d", sd, dev)
with torch.no_grad():
    P = cell.param_block().contiguous()
transport = sys.argv[1] if len(sys.argv) > 1 else "peer"
ex = slab.PeerHaloExchanger(force_p2p=True) if transport == "peer" else slab.HaloExchanger()
planes, hw, T, halo = 32, 256, 40, 4
local = torch.zeros((2, planes + 2 * halo, hw, hw), device=dev)
local[:, halo:halo + planes] = synthetic.gs_initial_state((planes, hw, hw), seed=0)[0].to(dev)
traj = torch.zeros((T + 1,) + tuple(local.shape), device=dev)
traj[0] = local
gtraj = torch.randn(traj.shape, device=dev) / traj.numel()
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slab.slab_rollout_fwd_(traj, P, ex, halo)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    slab.slab_rollout_bwd(traj, gtraj, P, ex, halo)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"{transport}: fwd enqueue {1e6 * (t1 - t0) / T:.1f} us/step, fwd done {1e6 * (t2 - t0) / T:.1f}; "
          f"bwd enqueue {1e6 * (t3 - t2) / T:.1f}, bwd done {1e6 * (t4 - t2) / T:.1f}")
if hasattr(ex, "close"):
    ex.close()
