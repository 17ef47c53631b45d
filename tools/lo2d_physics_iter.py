"""lambda-omega training iteration through the modules (bench.py's lo2d_physics_path_extra), fused loss node vs the
residual-tensor expression, and the loss kernels alone.  MI355X:  python tools/lo2d_physics_iter.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import percnn_amd as pa
import bench
from percnn_amd import physics


def main():
    dev = torch.device("cuda:0")
    family, shape, hc, dtype, T, golden = bench.WORKLOADS["lo2d_512"]
    cell = bench.make_cell(family, bench.load_params(golden), dev, "poly")
    h0 = bench.initial_state(family, shape).to(dev).requires_grad_(True)
    model = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h0)
    params = [p for p in cell.parameters() if p.requires_grad]
    Q = physics.lambda_omega_block(cell, 0.1)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for fused in (True, False):
        def it():
            loss = physics.physics_loss(model.trajectory(), Q, fused=fused)
            torch.autograd.grad(loss, params + [h0])
        print(f"iteration fused={fused}: {timed(it):.2f} ms", flush=True)
    with torch.no_grad():
        traj = model.trajectory()
    for fused in (True, False):
        t = traj.detach().clone().requires_grad_(True)
        print(f"loss forward fused={fused}: {timed(lambda: physics.physics_loss(t, Q, fused=fused)):.3f} ms")
        def fb():
            physics.physics_loss(t, Q, fused=fused).backward(); t.grad = None
        print(f"loss fwd+bwd fused={fused}: {timed(fb):.3f} ms")
    def roll():
        torch.autograd.grad(model.trajectory()[-1].sum(), params + [h0])
    print(f"rollout fwd+bwd (last-frame loss): {timed(roll):.2f} ms   bytes/frame {traj[0].numel() * traj.element_size() / 1e6:.1f} MB x {traj.shape[0]}")


if __name__ == "__main__":
    main()
