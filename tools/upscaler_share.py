"""how much of a training iteration is the stock-MIOpen IC generator once the T-step rollout is fused?
Times modules.RCNN forward + loss.backward() with and without the upscaler in front (2D GS 512^2 T=1000, 3D GS 128^3 T=500)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import percnn_amd as pa
from bench import load_params, make_cell
dev = torch.device("cuda:0")

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for fam, gold, low, T, ndim in (("gs2d", "gs2d_big_512x512.npz", (128, 128), 1000, 2), ("gs3d", "gs3d_big_128x128x128.npz", (64, 64, 64), 500, 3)):
    cell = make_cell(fam, load_params(gold), dev)
    up = pa.Upscaler(ndim).to(dev)
    low_t = torch.rand((1, 2) + low, device=dev) * 0.2 + 0.4
    with torch.no_grad():
        h0 = up(low_t)
    shape = tuple(h0.shape[2:])
    def full():
        m = pa.RCNN(cell, step=T, effective_step=list(range(T)), upscaler=up, init_state_low=low_t)
        outs, _ = m()
        loss = (torch.cat(tuple(outs), 0) ** 2).mean()
        loss.backward()
    def rollout_only():
        h = h0.clone().requires_grad_(True)
        m = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h)
        outs, _ = m()
        loss = (torch.cat(tuple(outs), 0) ** 2).mean()
        loss.backward()
    def up_only():
        x = up(low_t)
        x.backward(torch.ones_like(x))
    a, b, c = timed(full), timed(rollout_only), timed(up_only)
    print(f"{fam} {shape} T={T}: iteration with upscaler {a:.2f} ms, without {b:.2f} ms, upscaler fwd+bwd alone {c:.3f} ms "
          f"({100 * c / a:.1f} % of the iteration)", flush=True)
