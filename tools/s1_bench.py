"""Stage-1 Pi-block timing on the GPU box: forward, adjoint sweep, weight-gradient kernel; MFMA roofline fraction.
    python tools/s1_bench.py [--cpu]   (--cpu also times the torch restatement on the host cores)
"""
import argparse, json, os, sys, time
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import percnn_amd as pa                                   # noqa: E402

MFMA_F32_PEAK = 157.3e12
FLOP_BRANCH = 2 * 2 * 3 * 16 * 50        # per point: 2 species x 3 branches x 16 channels x 50 taps, 2 flop / MAC


def load_cell(case, dev):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{case}_stage1_32x32.npz"))
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    cell = pa.Stage1Cell({"bur1": "burgers", "lo1": "lo"}[case]).to(dev)
    cell.load_state_dict(sd)
    return cell, sd


def measure(case, shape, T, dev, reps=5):
    cell, _ = load_cell(case, dev)
    with torch.no_grad():
        P = cell.param_block().contiguous()
    n = shape[0] * shape[1]
    ys, xs = torch.meshgrid(torch.arange(shape[0]) / shape[0], torch.arange(shape[1]) / shape[1], indexing="ij")
    traj = torch.empty((T + 1, 2) + shape, device=dev)
    traj[0, 0] = (0.6 * torch.sin(2 * np.pi * xs) * torch.cos(2 * np.pi * ys)).to(dev)
    traj[0, 1] = (0.6 * torch.cos(2 * np.pi * xs) * torch.sin(2 * np.pi * ys)).to(dev)
    g = torch.randn_like(traj) * 1e-4
    pa.stage1.rollout_fwd_(traj, P); pa.stage1.rollout_bwd(traj, g, P); torch.cuda.synchronize()
    assert torch.isfinite(traj[-1]).all()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(reps): pa.stage1.rollout_fwd_(traj, P)
    ev[1].record()
    for _ in range(reps): pa.stage1.rollout_bwd(traj, g, P)
    ev[2].record(); torch.cuda.synchronize()
    fwd = ev[0].elapsed_time(ev[1]) / reps * 1e-3
    bwd = ev[1].elapsed_time(ev[2]) / reps * 1e-3
    r = {"case": case, "shape": list(shape), "T": T, "fwd_us_step": fwd / T * 1e6, "bwd_us_step": bwd / T * 1e6,
         "steps_per_s": T / (fwd + bwd),
         "fwd_TFLOPs": FLOP_BRANCH * n * T / fwd / 1e12, "fwd_frac_mfma_f32": FLOP_BRANCH * n * T / fwd / MFMA_F32_PEAK,
         # backward = branch recompute (sweep) + input-gradient GEMM + branch recompute (wgrad) + weight-gradient GEMM
         "bwd_TFLOPs": 4 * FLOP_BRANCH * n * T / bwd / 1e12, "bwd_frac_mfma_f32": 4 * FLOP_BRANCH * n * T / bwd / MFMA_F32_PEAK}
    return r


def cpu_baseline(case, shape, T):
    from oracle import restatement as R
    _, sd = load_cell(case, torch.device("cuda:0"))
    cell = R.OracleStage1Cell({"bur1": "burgers", "lo1": "lo"}[case])
    cell.load_state_dict(sd)
    best = None
    for nt in (8, 16, 32, 64):
        torch.set_num_threads(nt)
        h = (torch.rand(1, 2, *shape) * 0.2).requires_grad_(True)
        t0 = time.perf_counter()
        outs = [h]; x = h
        for _ in range(T):
            x, _ = cell(x); outs.append(x)
        (torch.cat(outs) ** 2).mean().backward()
        el = time.perf_counter() - t0
        if best is None or el < best[0]:
            best = (el, nt)
    return {"steps_per_s": T / best[0], "threads": best[1], "sample": f"{T} steps fwd+bwd at {shape}"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--out", default="gpurun_out/s1_bench.json")
    ap.add_argument("--only", type=int, default=0, help="grid edge: run only the cases of that size")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for case, shape, T in (("bur1", (100, 100), 200), ("lo1", (100, 100), 200), ("bur1", (256, 256), 100),
                           ("bur1", (512, 512), 50), ("bur1", (1024, 1024), 20)):
        if a.only and shape[0] != a.only:
            continue
        r = measure(case, shape, T, dev)
        rows.append(r)
        print("%s %4dx%-4d T=%3d fwd %7.2f us/step (%5.1f TF, %4.1f%% of f32 MFMA peak)  bwd %7.2f us/step (%5.1f TF, %4.1f%%)  %8.0f steps/s"
              % (case, shape[0], shape[1], T, r["fwd_us_step"], r["fwd_TFLOPs"], 100 * r["fwd_frac_mfma_f32"], r["bwd_us_step"],
                 r["bwd_TFLOPs"], 100 * r["bwd_frac_mfma_f32"], r["steps_per_s"]), flush=True)
    if a.cpu:
        c = cpu_baseline("bur1", (100, 100), 40)
        print("cpu", c, flush=True)
        rows.append({"cpu_baseline": c})
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)
