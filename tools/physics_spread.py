import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
import percnn_amd as pa
from percnn_amd import physics
from util import Golden, small_cases, rel_l2
from oracle import restatement as R
dev=torch.device("cuda:0")
for fn in small_cases():
    g=Golden(fn)
    cell=g.product_cell(dev)
    traj=R.rollout(g.oracle_cell(), torch.tensor(g.h0), g.steps).detach()
    Q = {"gs2d": lambda: physics.gray_scott_block(cell, 2e-5, 2e-5 / 4, 1 / 25, 3 / 50),
         "gs3d": lambda: physics.gray_scott_block(cell, 0.2, 0.1, 0.025, 0.055),
         "lo2d": lambda: physics.lambda_omega_block(cell, 0.1)}[g.family]()
    out=traj.to(dev).requires_grad_(True); loss=physics.physics_loss(out,Q); loss.backward()
    a=traj.clone().requires_grad_(True); l32=R.physics_loss_reference(a,g.family,g.dx,g.dt); l32.backward()
    b=traj.double().requires_grad_(True); l64=R.physics_loss_reference(b,g.family,g.dx,g.dt); l64.backward()
    print(fn.split('/')[-1], "value: ours-vs-64 %.2e ref32-vs-64 %.2e ours-vs-ref32 %.2e | grad: ours-vs-64 %.2e ref32-vs-64 %.2e"%(
        abs(loss.item()-l64.item())/abs(l64.item()), abs(l32.item()-l64.item())/abs(l64.item()), abs(loss.item()-l32.item())/abs(l32.item()),
        rel_l2(out.grad.cpu().numpy(), b.grad.numpy()), rel_l2(a.grad.numpy(), b.grad.numpy())))
