#!/usr/bin/env python3
"""The reference's 2D Gray-Scott training iteration (DataDrivenModeling/2d_gs_rd/train_2drd.py:373-410) on percnn_amd.

Everything outside the hot path is the reference's own recipe -- Adam + StepLR (train_2drd.py:383-384), loss =
40 * data MSE on output[0:-1:20, :, ::4, ::4] + 0.25 * IC loss (:397-406), physics loss for monitoring (:405) -- only
the model construction and the loss plumbing use this package.  There is no dataset in this environment, so the
"truth" is a rollout of a teacher cell with perturbed weights (synthetic); the point is the wiring, not the science.

    python examples/train_2dgs_synthetic.py --iters 20 --size 100 --steps 200
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import percnn_amd as pa                                     # noqa: E402
from percnn_amd import physics, synthetic                   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--size", type=int, default=100)          # the reference trains on 100 x 100 (train_2drd.py:331)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--reference-lines", action="store_true",
                    help="build the data-loss operand with the reference's own lines (train_2drd.py:393-397: model(), torch.cat, strided "
                         "slice) instead of RCNN.observe(): since round 5 that torch.cat returns the trajectory buffer (functional.Frame)")
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, T = a.size, a.steps

    # synthetic ground truth: a teacher Pi-block rolled out from a Gray-Scott-like initial state
    teacher = pa.gs2d_cell(8).to(dev)
    for f in teacher.filter_list:
        f.weight.data.mul_(12.0)
    h0_true = synthetic.gs_initial_state((n, n), seed=0).to(dev)
    with torch.no_grad():
        truth = pa.pi_rollout(h0_true, teacher.param_block(), T)              # [T+1, 2, n, n]
        low = F.avg_pool2d(h0_true, 4)                                       # "low-resolution measurement" of the IC
    assert torch.isfinite(truth).all(), "teacher rollout diverged"

    # student: fresh cell + IC generator, wired exactly like the reference's RCNN (train_2drd.py:630-636)
    cell = pa.gs2d_cell(8).to(dev)
    for f in cell.filter_list:
        f.weight.data.mul_(12.0).add_(torch.randn_like(f.weight) * 0.02)
    model = pa.RCNN(cell, step=T, effective_step=list(range(T)), upscaler=pa.Upscaler(2), init_state_low=low).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=200, gamma=0.985)
    Q = physics.gray_scott_block(cell, Du=2e-5, Dv=5e-6, f=1 / 25, k=3 / 50)     # the true equation (train_2drd.py:321-327)
    ic_target = F.interpolate(low, (n, n), mode="bicubic")
    gt = truth[0:-1:20, :, ::4, ::4]
    losses, t0 = [], None
    for it in range(a.iters):
        if it == min(3, a.iters - 1):                                        # steady state: skip first-use setup (MIOpen, allocator)
            torch.cuda.synchronize()
            t0, n0 = time.perf_counter(), it
        opt.zero_grad()
        if a.reference_lines:
            output, second_last_state = model()                              # train_2drd.py:393-397, verbatim
            output = torch.cat(tuple(output), dim=0)
            pred = output[0:-1:20, :, ::4, ::4]
        else:
            pred = model.observe(slice(0, -1, 20), 4)                        # the same operand from ONE autograd node (masked sweep)
        idx = int(pred.shape[0] * 0.9)
        loss_data = F.mse_loss(pred[:idx], gt[:idx])
        loss_val = F.mse_loss(pred[idx:], gt[idx:])
        loss_ic = F.mse_loss(model.UpconvBlock(low), ic_target)
        with torch.no_grad():                                                # monitoring only, as in the reference
            loss_phy = physics.physics_loss(output.detach() if a.reference_lines else model.last_trajectory, Q)
        loss = 40 * loss_data + 0.25 * loss_ic
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
        print(f"[{it + 1:3d}] loss {loss.item():.6e}  data {loss_data.item():.3e}  val {loss_val.item():.3e}  "
              f"ic {loss_ic.item():.3e}  phy {loss_phy.item():.3e}")
    torch.cuda.synchronize()
    print(f"{T}-step {n}x{n} rollout: {(time.perf_counter() - t0) / max(a.iters - n0, 1) * 1e3:.2f} ms per training iteration "
          f"(steady state, {a.iters - n0} iterations)")
    return losses


if __name__ == "__main__":
    main()
