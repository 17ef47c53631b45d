import sys, time, torch
sys.path.insert(0, "/root/repo")
import percnn_amd as pa
from percnn_amd import functional as F_pi, synthetic
dev = torch.device("cuda:0")
n, T = 100, 200
cell = pa.gs2d_cell(8).to(dev)
for f in cell.filter_list: f.weight.data.mul_(12.0)
h0 = synthetic.gs_initial_state((n, n), seed=0).to(dev)
def timeit(fn, k=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / k
def via_ops():
    for p in cell.parameters(): p.grad = None
    P = cell.param_block()
    traj = F_pi.pi_rollout(h0, P, T)
    traj.backward(torch.ones_like(traj) * 1e-6)
g = torch.ones((T + 1, 2, n, n), device=dev) * 1e-6
def direct():
    with torch.no_grad():
        P = cell.param_block()
    traj = torch.empty((T + 1, 2, n, n), device=dev); traj[0] = h0[0]
    pa.rollout_fwd_(traj, P)
    g0, pg = pa.rollout_bwd(traj, g, P)
def obs():
    for p in cell.parameters(): p.grad = None
    m = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h0)
    pred = m.observe(slice(0, -1, 20), 4)
    pred.sum().backward()
print(f"registered ops (param_block + pi_rollout + backward): {timeit(via_ops):.0f} us per iteration")
print(f"direct C-ABI wrappers, no autograd:                    {timeit(direct):.0f} us")
print(f"RCNN.observe + backward:                               {timeit(obs):.0f} us")
