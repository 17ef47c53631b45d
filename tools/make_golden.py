#!/usr/bin/env python3
"""Generate golden vectors by importing the REAL reference scripts (build container only).

    python tools/make_golden.py            # all small cases  -> tests/golden/*.npz
    python tools/make_golden.py --big      # + 512^2 x1000 and 128^3 x500 statistics (minutes)
    python tools/make_golden.py --biggrad  # + reference GRADIENTS at 512^2 (T=100), 128^3 (T=20), lambda-omega 512^2 (T=100)

The reference lives read-only at /root/reference and never travels to the GPU box; only
the data written here (inputs + expected outputs) is committed.  Every case also asserts
that ``oracle/restatement.py`` reproduces the reference BIT-FOR-BIT on the same inputs and
weights, which is what pins the oracle.

Each reference script is imported in its own subprocess: importing ``lo`` flips the global
default dtype to float64 (lo:12) and every script sets CUDA_VISIBLE_DEVICES at import.
The scripts hard-call ``.cuda()`` (2dgs:150,261,268): the harness shims ``.cuda`` to a no-op
before import -- the reference files themselves are not modified or copied.
"""
import argparse
import importlib.util
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
SCRIPTS = {
    "gs2d": "DataDrivenModeling/2d_gs_rd/train_2drd.py",
    "gs3d": "DataDrivenModeling/3d_gs_rd/train_3drd.py",
    "lo2d": "ForwardSimulationOfPDEs/2d_lambda_omega/percnn_LO_eqn.py",
    "lo3": "DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-3/fine_tuning_LO_[10%noise,41x51x51].py",
    "bur3": "DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-3/fine_tuning_[5%noise,41x51x51].py",
    "bur1": "DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/rcnn_Burgers_[resnet,GT41x51x51,LAPLACE,5%noise].py",
    "lo1": "DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-1/rcnn_LO_[resnet,GT41x51x51,LAPLACE,5%noise].py",
}
CKPT = {
    "gs2d": "DataDrivenModeling/2d_gs_rd/model/checkpoint.pt",
    "gs3d": "DataDrivenModeling/3d_gs_rd/model/checkpoint.pt",
    "lo2d": "ForwardSimulationOfPDEs/2d_lambda_omega/model/rcnn_pde.pt",
    "bur1": "DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/model/checkpoint.pt",
    "lo1": "DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-1/model/checkpoint.pt",
}
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference(case):
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    import matplotlib
    matplotlib.use("Agg")
    path = os.path.join(REF, SCRIPTS[case])
    spec = importlib.util.spec_from_file_location("ref_" + case, path)
    mod = importlib.util.module_from_spec(spec)
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    try:
        spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
    return mod


def ref_cell(mod, case):
    if case == "gs2d":
        return mod.RCNNCell(2, 8, 5)
    if case == "gs3d":
        return mod.RCNNCell(2, 2, 5)
    return mod.RCNNCell(input_kernel_size=1, input_stride=1, input_padding=0)


def oracle_cell(case):
    from oracle import restatement as R
    return {"gs2d": R.gs2d_cell, "gs3d": R.gs3d_cell, "lo2d": R.lo2d_cell}[case]()


def ckpt_cell_state(case):
    sd = torch.load(os.path.join(REF, CKPT[case]), map_location="cpu", weights_only=False)
    if "model_state_dict" in sd:
        sd = sd["model_state_dict"]
    out = {}
    for k, v in sd.items():
        for pre in ("crnn_cell.", "rcnn_cell."):
            if k.startswith(pre):
                out[k[len(pre):]] = v
    return out, sd


def initial_state(case, shape):
    from oracle import restatement as R
    if case == "lo2d":
        assert shape[0] == shape[1]
        return R.lo_initial_state(shape[0])
    return R.gs_initial_state(shape, seed=0)


def run_traj(cell, h0, steps):
    outs = [h0]
    h = h0
    for _ in range(steps):
        h, _ = cell(h)
        outs.append(h)
    return torch.cat(tuple(outs), dim=0)


def data_loss(traj, stride_t, ndim):
    sl = (slice(0, -1, stride_t), slice(None)) + (slice(None, None, 4),) * ndim   # cf. 2dgs:397
    return ((traj[sl] - 0.5) ** 2).mean()


def grads_of(loss, cell, h0):
    names = [n for n, p in cell.named_parameters() if p.requires_grad]
    params = [p for n, p in cell.named_parameters() if p.requires_grad]
    g = torch.autograd.grad(loss, params + [h0], retain_graph=True, allow_unused=True)
    return {n: gi for n, gi in zip(names, g[:-1])}, g[-1]


def small_case(case, mod, tag, state, shape, steps, keep_t, stride_t):
    """One fixture: reference vs restatement (bit-equal) + saved vectors."""
    ndim = len(shape)
    rc, oc = ref_cell(mod, case), oracle_cell(case)
    if state is not None:
        rc.load_state_dict(state)
    oc.load_state_dict(rc.state_dict())
    assert sorted(rc.state_dict().keys()) == sorted(oc.state_dict().keys())
    h0r = initial_state(case, shape).requires_grad_(True)
    h0o = h0r.detach().clone().requires_grad_(True)
    tr, to = run_traj(rc, h0r, steps), run_traj(oc, h0o, steps)
    assert torch.equal(tr, to), f"{case}/{tag}: restatement forward differs from reference"
    rec = {"h0": h0r.detach().numpy(), "steps": steps, "keep_t": np.array(keep_t),
           "stride_t": stride_t, "dx": rc.dx, "dt": rc.dt,
           "mu_up": getattr(rc, "mu_up", np.nan)}
    for k, v in rc.state_dict().items():
        rec["param/" + k] = v.numpy()
    for t in keep_t:
        rec[f"traj/{t}"] = tr[t].detach().numpy()
    for lname, lf in (("meansq", lambda x: (x ** 2).mean()),
                      ("data", lambda x: data_loss(x, stride_t, ndim))):
        lr, lo = lf(tr), lf(to)
        gr, ghr = grads_of(lr, rc, h0r)
        go, gho = grads_of(lo, oc, h0o)
        assert torch.equal(lr, lo)
        for n in gr:
            assert torch.equal(gr[n], go[n]), f"{case}/{tag}: grad {n} differs"
            rec[f"grad_{lname}/{n}"] = gr[n].numpy()
        assert torch.equal(ghr, gho)
        rec[f"loss_{lname}"] = lr.item()
        rec[f"grad_{lname}_h0"] = ghr.numpy()
    # known-answer scalar: the reference's own physics residual of this trajectory
    # (2dgs:340-353 loss_gen / 3dgs:334-346 loss_func / lo:343-357 loss_gen)
    lg = mod.loss_generator(rc.dt, rc.dx)
    phy = mod.loss_func if case == "gs3d" else mod.loss_gen
    rec["phy_loss"] = float(phy(tr.detach(), lg))
    fn = os.path.join(OUT, f"{case}_{tag}_{'x'.join(map(str, shape))}.npz")
    np.savez_compressed(fn, **rec)
    print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)  "
          f"loss_meansq={rec['loss_meansq']:.9g}")


def rcnn_harness_case(case, mod):
    """Pins a9: RCNN.forward incl. upscaler / effective_step / second_last_state."""
    from oracle import restatement as R
    _, full = ckpt_cell_state(case)
    g = torch.Generator().manual_seed(1)
    if case == "gs2d":
        low = 0.5 + 0.1 * torch.randn((1, 2, 8, 8), generator=g)
        steps = 12
        eff = list(range(0, steps, 1))
        m = mod.RCNN(input_channels=2, hidden_channels=8, init_state_low=low, input_kernel_size=5,
                     step=steps, effective_step=eff)
        m.load_state_dict(full)
        o = R.OracleRCNN(R.gs2d_cell(), step=steps, effective_step=eff, upscaler=R.OracleUpscaler(2),
                         init_state_low=low)
    elif case == "gs3d":
        low = 0.5 + 0.1 * torch.randn((1, 2, 6, 6, 6), generator=g)
        steps = 6
        eff = [0, 2, 3, 5]            # sparse effective_step: membership must be honoured
        m = mod.RCNN(input_channels=2, hidden_channels=2, init_state_low=low, input_kernel_size=5,
                     step=steps, effective_step=eff)
        m.load_state_dict(full)
        o = R.OracleRCNN(R.gs3d_cell(), step=steps, effective_step=eff, upscaler=R.OracleUpscaler(3),
                         init_state_low=low)
    else:
        ini = R.lo_initial_state(20).numpy()
        steps = 8
        eff = list(range(steps))
        m = mod.RCNN(input_kernel_size=1, ini_state=ini, input_stride=1, input_padding=0, step=steps,
                     effective_step=eff)
        m.load_state_dict({k.replace("crnn_cell.", "rcnn_cell."): v for k, v in full.items()})
        o = R.OracleRCNN(R.lo2d_cell(), step=steps, effective_step=eff,
                         init_state=torch.tensor(ini, dtype=torch.float64), cell_name="rcnn_cell")
    o.load_state_dict(m.state_dict())
    assert list(o.state_dict().keys()) == list(m.state_dict().keys())
    with torch.no_grad():
        outs_r, sl_r = m()
        outs_o, sl_o = o()
    assert len(outs_r) == len(outs_o)
    for a, b in zip(outs_r, outs_o):
        assert torch.equal(a, b)
    assert torch.equal(sl_r, sl_o)
    rec = {"steps": steps, "effective_step": np.array(eff), "second_last_state": sl_r.numpy(),
           "outputs": torch.cat(tuple(outs_r), 0).numpy()}
    if case != "lo2d":
        rec["init_state_low"] = low.numpy()
    else:
        rec["init_state"] = ini
    for k, v in m.state_dict().items():
        rec["state/" + k] = v.numpy()
    fn = os.path.join(OUT, f"{case}_rcnn_harness.npz")
    np.savez_compressed(fn, **rec)
    print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)")


def big_case(case, mod, shape, checkpoints):
    """Full-size reference run on CPU; keeps an every-8th-point subsample + statistics."""
    state, _ = ckpt_cell_state(case)
    rc = ref_cell(mod, case)
    rc.load_state_dict(state)
    h = initial_state(case, shape)
    ndim = len(shape)
    sub = (slice(None), slice(None)) + (slice(None, None, 8),) * ndim
    rec = {"h0_seed": 0, "checkpoints": np.array(checkpoints)}
    for k, v in rc.state_dict().items():
        rec["param/" + k] = v.numpy()
    import time
    t0 = time.time()
    with torch.no_grad():
        for t in range(1, max(checkpoints) + 1):
            h, _ = rc(h)
            if t in checkpoints:
                rec[f"sub/{t}"] = h[sub].numpy()
                rec[f"l2/{t}"] = float(torch.linalg.vector_norm(h.double()))
                rec[f"minmax/{t}"] = np.array([h[:, 0].min(), h[:, 0].max(), h[:, 1].min(), h[:, 1].max()])
                print(f"   t={t} ({time.time()-t0:.0f}s) |h|={rec[f'l2/{t}']:.9g}")
    fn = os.path.join(OUT, f"{case}_big_{'x'.join(map(str, shape))}.npz")
    np.savez_compressed(fn, **rec)
    print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)")


def biggrad_case(case, mod, shape, steps, stride_t, twin=True, tag="biggrad"):
    """Reference forward + autograd backward at the BASELINE grid size (shorter horizon: the reference's tape is
    ~80 MB per 512^2 step): both losses, every parameter gradient, an every-8th-point subsample of dL/dh0 and of the
    last frame.  This is the backward the reference triggers at 2dgs:407 / 3dgs:408 / lo:373, at full spatial size."""
    import time
    state, _ = ckpt_cell_state(case)
    rc = ref_cell(mod, case)
    rc.load_state_dict(state)
    ndim = len(shape)
    sub = (slice(None), slice(None)) + (slice(None, None, 8),) * ndim
    rec = {"h0_seed": 0, "steps": steps, "stride_t": stride_t}
    for k, v in rc.state_dict().items():
        rec["param/" + k] = v.numpy()
    t0 = time.time()
    h0 = initial_state(case, shape).requires_grad_(True)
    tr = run_traj(rc, h0, steps)
    print(f"   forward {steps} steps ({time.time()-t0:.0f}s)")
    rec["sub_last"] = tr[-1:].detach()[sub].numpy()
    rec["l2_last"] = float(torch.linalg.vector_norm(tr[-1].detach().double()))
    for lname, lf in (("meansq", lambda x: (x ** 2).mean()), ("data", lambda x: data_loss(x, stride_t, ndim))):
        loss = lf(tr)
        g, gh = grads_of(loss, rc, h0)
        rec[f"loss_{lname}"] = loss.item()
        for n in g:
            rec[f"grad_{lname}/{n}"] = g[n].numpy()
        rec[f"grad_{lname}_h0_sub"] = gh[sub].numpy()
        rec[f"grad_{lname}_h0_l2"] = float(torch.linalg.vector_norm(gh.double()))
        print(f"   loss {lname} = {loss.item():.9g}, backward done ({time.time()-t0:.0f}s)")
    del tr, loss, g, gh
    if h0.dtype == torch.float32 and twin:
        # float64 twin of the same cell (same float32-rounded weights and stencil taps, upcast): the yardstick that says
        # how far the float32 reference's OWN gradients are from the exact ones (its per-step bias / weight reductions
        # sum 262 144+ terms in float32)
        import copy
        oc = oracle_cell(case)
        oc.load_state_dict(rc.state_dict())
        oc = copy.deepcopy(oc).double()
        h64 = initial_state(case, shape).double().requires_grad_(True)
        t64 = run_traj(oc, h64, steps)
        for lname, lf in (("meansq", lambda x: (x ** 2).mean()), ("data", lambda x: data_loss(x, stride_t, ndim))):
            g64, gh64 = grads_of(lf(t64), oc, h64)
            worst = 0.0
            for n in g64:
                rec[f"grad64_{lname}/{n}"] = g64[n].numpy()
                ref = torch.tensor(rec[f"grad_{lname}/{n}"]).double()
                worst = max(worst, ((ref - g64[n]).norm() / g64[n].norm()).item())
            rec[f"grad64_{lname}_h0_sub"] = gh64[sub].numpy()
            print(f"   float64 twin, loss {lname}: worst per-tensor rel-L2 of the float32 reference's parameter "
                  f"gradients = {worst:.2e} ({time.time()-t0:.0f}s)")
    fn = os.path.join(OUT, f"{case}_{tag}_{'x'.join(map(str, shape))}.npz")
    np.savez_compressed(fn, **rec)
    print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)")


def stage3_rk4_vectors(rc, oc, h0, rec, steps=5):
    """forward_rk4 of the reference cell (defined there, never called by its scripts): `steps` RK4 steps from h0, the
    restatement asserted bit-equal, frames 1 and `steps` + the gradients of mean(h_steps^2) saved."""
    def roll(cell, h):
        outs = []
        for _ in range(steps):
            h, _ = cell.forward_rk4(h)
            outs.append(h)
        return outs
    h0r, h0o = h0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
    tr, to = roll(rc, h0r), roll(oc, h0o)
    for a, b in zip(tr, to):
        assert torch.equal(a, b), "stage-3 forward_rk4 restatement differs from the reference"
    rec["rk4_steps"] = steps
    rec["rk4/1"], rec[f"rk4/{steps}"] = tr[0].detach().numpy(), tr[-1].detach().numpy()
    lr, lo = (tr[-1] ** 2).mean(), (to[-1] ** 2).mean()
    gr, ghr = grads_of(lr, rc, h0r)
    go, gho = grads_of(lo, oc, h0o)
    for n in gr:
        assert torch.equal(gr[n], go[n]), n
        rec["rk4_grad_meansq/" + n] = gr[n].numpy()
    assert torch.equal(ghr, gho)
    rec["rk4_loss_meansq"] = lr.item()
    rec["rk4_grad_meansq_h0"] = ghr.numpy()


def stage3_lo_case(mod):
    """the Stage-3 physics-based lambda-omega cell (13 trainable scalars, Euler)."""
    from oracle import restatement as R
    rc = mod.RCNNCell(input_channels=2, hidden_channels=16, output_channels=2, input_kernel_size=5,
                      input_stride=1, input_padding=2)
    oc = R.OracleStage3LOCell()
    assert list(rc.state_dict().keys()) == list(oc.state_dict().keys())
    oc.load_state_dict(rc.state_dict())
    for shape, steps, keep in (((32, 32), 40, [1, 2, 10, 40]), ((24, 40), 10, [1, 10])):
        x = [(torch.arange(n, dtype=torch.float64) - n / 2) * 0.2 for n in shape]
        yy, xx = torch.meshgrid(*x, indexing="ij")
        r, th = torch.sqrt(xx ** 2 + yy ** 2), torch.atan2(yy, xx)
        h0 = torch.stack((torch.tanh(r) * torch.cos(th - r), torch.tanh(r) * torch.sin(th - r)))[None]
        h0r, h0o = h0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        tr, to = run_traj(rc, h0r, steps), run_traj(oc, h0o, steps)
        assert torch.equal(tr, to), "stage-3 restatement differs from the reference"
        rec = {"h0": h0.numpy(), "steps": steps, "keep_t": np.array(keep), "dx": rc.dx, "dt": rc.dt}
        for k, v in rc.state_dict().items():
            rec["param/" + k] = v.numpy()
        for t in keep:
            rec[f"traj/{t}"] = tr[t].detach().numpy()
        lr, lo = (tr ** 2).mean(), (to ** 2).mean()
        gr, ghr = grads_of(lr, rc, h0r)
        go, gho = grads_of(lo, oc, h0o)
        for n in gr:
            assert torch.equal(gr[n], go[n]), n
            rec["grad_meansq/" + n] = gr[n].numpy()
        assert torch.equal(ghr, gho)
        rec["loss_meansq"] = lr.item()
        rec["grad_meansq_h0"] = ghr.numpy()
        stage3_rk4_vectors(rc, oc, h0, rec)
        fn = os.path.join(OUT, f"lo3_stage3_{'x'.join(map(str, shape))}.npz")
        np.savez_compressed(fn, **rec)
        print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)")


def stage3_burgers_case(mod):
    """the Stage-3 physics-based 2D Burgers cell (6 trainable scalars, Euler)."""
    from oracle import restatement as R
    rc = mod.RCNNCell(input_channels=2, hidden_channels=16, output_channels=2, input_kernel_size=5,
                      input_stride=1, input_padding=2)
    oc = R.OracleStage3BurgersCell()
    assert list(rc.state_dict().keys()) == list(oc.state_dict().keys())
    oc.load_state_dict(rc.state_dict())
    for shape, steps, keep in (((32, 32), 40, [1, 2, 10, 40]), ((24, 40), 10, [1, 10])):
        ys = torch.arange(shape[0], dtype=torch.float64) / shape[0]
        xs = torch.arange(shape[1], dtype=torch.float64) / shape[1]
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        two_pi = 2 * np.pi
        u = torch.sin(two_pi * xx) * torch.cos(two_pi * yy) + 0.3 * torch.cos(2 * two_pi * xx + 0.5)
        v = torch.cos(two_pi * xx) * torch.sin(two_pi * yy) - 0.2 * torch.sin(two_pi * (xx + 2 * yy))
        h0 = torch.stack((u, v))[None]
        h0r, h0o = h0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        tr, to = run_traj(rc, h0r, steps), run_traj(oc, h0o, steps)
        assert torch.equal(tr, to), "stage-3 Burgers restatement differs from the reference"
        rec = {"h0": h0.numpy(), "steps": steps, "keep_t": np.array(keep), "dx": rc.dx, "dt": rc.dt}
        for k, v_ in rc.state_dict().items():
            rec["param/" + k] = v_.numpy()
        for t in keep:
            rec[f"traj/{t}"] = tr[t].detach().numpy()
        lr, lo = (tr ** 2).mean(), (to ** 2).mean()
        gr, ghr = grads_of(lr, rc, h0r)
        go, gho = grads_of(lo, oc, h0o)
        for n in gr:
            assert torch.equal(gr[n], go[n]), n
            rec["grad_meansq/" + n] = gr[n].numpy()
        assert torch.equal(ghr, gho)
        rec["loss_meansq"] = lr.item()
        rec["grad_meansq_h0"] = ghr.numpy()
        stage3_rk4_vectors(rc, oc, h0, rec)
        fn = os.path.join(OUT, f"bur3_stage3_{'x'.join(map(str, shape))}.npz")
        np.savez_compressed(fn, **rec)
        print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)")


def stage1_case(mod, case):
    """the Stage-1 Pi-block (three 5x5 conv branches 2 -> 16 per species), float32, with the
    reference's own trained weights (Stage-1 checkpoint)."""
    from oracle import restatement as R
    fam = {"bur1": "burgers", "lo1": "lo"}[case]
    state, _ = ckpt_cell_state(case)
    rc = mod.RCNNCell(input_channels=2, hidden_channels=16, output_channels=2, input_kernel_size=5,
                      input_stride=1, input_padding=2)
    oc = R.OracleStage1Cell(fam)
    assert list(rc.state_dict().keys()) == list(oc.state_dict().keys())
    assert all(rc.state_dict()[k].shape == oc.state_dict()[k].shape and rc.state_dict()[k].dtype == oc.state_dict()[k].dtype
               for k in rc.state_dict())
    assert (rc.dx, rc.dt, rc.nu_up) == (oc.dx, oc.dt, oc.nu_up)
    rc.load_state_dict(state); oc.load_state_dict(state)
    o64 = R.OracleStage1Cell(fam, dtype=torch.float64)
    o64.load_state_dict({k: v.double() for k, v in state.items()})
    for shape, steps, keep in (((32, 32), 40, [1, 2, 10, 40]), ((24, 40), 10, [1, 10]), ((22, 26), 6, [1, 6]),
                               ((100, 100), 200, [1, 20, 200])):
        ys = torch.arange(shape[0], dtype=torch.float64) / shape[0]
        xs = torch.arange(shape[1], dtype=torch.float64) / shape[1]
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        two_pi = 2 * np.pi
        u = 0.6 * torch.sin(two_pi * xx) * torch.cos(two_pi * yy) + 0.3 * torch.cos(2 * two_pi * xx + 0.5)
        v = 0.6 * torch.cos(two_pi * xx) * torch.sin(two_pi * yy) - 0.2 * torch.sin(two_pi * (xx + 2 * yy))
        h0 = torch.stack((u, v))[None].float()
        h0r, h0o = h0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
        tr, to = run_traj(rc, h0r, steps), run_traj(oc, h0o, steps)
        assert torch.equal(tr, to), "stage-1 restatement differs from the reference"
        assert torch.isfinite(tr).all()
        rec = {"h0": h0.numpy(), "steps": steps, "keep_t": np.array(keep), "dx": rc.dx, "dt": rc.dt, "nu_up": rc.nu_up}
        for k, v_ in rc.state_dict().items():
            rec["param/" + k] = v_.numpy()
        for t in keep:
            rec[f"traj/{t}"] = tr[t].detach().numpy()
        lr, lo = (tr ** 2).mean(), (to ** 2).mean()
        gr, ghr = grads_of(lr, rc, h0r)
        go, gho = grads_of(lo, oc, h0o)
        h64 = h0.double().requires_grad_(True)
        t64 = run_traj(o64, h64, steps)
        g64, gh64 = grads_of((t64 ** 2).mean(), o64, h64)
        worst = 0.0
        for n in gr:
            assert torch.equal(gr[n], go[n]), n
            rec["grad_meansq/" + n] = gr[n].numpy()
            rec["grad64_meansq/" + n] = g64[n].numpy()
            worst = max(worst, ((gr[n].double() - g64[n]).norm() / g64[n].norm()).item())
        assert torch.equal(ghr, gho)
        rec["loss_meansq"] = lr.item()
        rec["grad_meansq_h0"] = ghr.numpy()
        rec["grad64_meansq_h0"] = gh64.numpy()
        rec["traj64_last"] = t64[-1].detach().numpy()
        print(f"  {case} {shape}: fp32 reference vs fp64: traj rel {((tr[-1].double()-t64[-1]).norm()/t64[-1].norm()).item():.2e}, "
              f"worst param-grad rel {worst:.2e}, |h| max {tr.abs().max().item():.3f}")
        fn = os.path.join(OUT, f"{case}_stage1_{'x'.join(map(str, shape))}.npz")
        np.savez_compressed(fn, **rec)
        print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)")


def train_iter_case(case, mod):
    """One TRAINING iteration of the reference script, composed exactly as its loop does (2dgs:393-407, 3dgs:399-408):
    ``model()`` -> ``torch.cat`` -> strided data loss against a truth tensor + ``get_ic_loss(model)`` (2dgs:331-338,
    3dgs:325-332: upscaler output vs the bicubic / trilinear interpolation of the low-resolution measurement) -> weighted sum
    -> ``backward()``.  Captured: the three losses, the physics-residual scalar and EVERY gradient the optimizer would see --
    the cell's and the IC generator's (2 034 / 20 k upscaler parameters).  Pins the upscaler's backward, ``get_ic_loss`` and
    the data-loss route of ``RCNN.observe`` / ``RCNN.loss_mse`` on something the reference holds."""
    from oracle import restatement as R
    _, full = ckpt_cell_state(case)
    g = torch.Generator().manual_seed(7)
    if case == "gs2d":
        n_low, n, steps, st, ss, w_data, w_ic = 25, 100, 60, 20, 4, 40.0, 0.25        # get_ic_loss hard-codes (100, 100)
        low = torch.cat((0.9 + 0.1 * torch.rand((1, 1, n_low, n_low), generator=g),
                         0.1 + 0.1 * torch.rand((1, 1, n_low, n_low), generator=g)), 1)
        eff = list(range(steps))
        m = mod.RCNN(input_channels=2, hidden_channels=8, init_state_low=low, input_kernel_size=5, step=steps,
                     effective_step=eff)
        o = R.OracleRCNN(R.gs2d_cell(), step=steps, effective_step=eff, upscaler=R.OracleUpscaler(2), init_state_low=low)
        ndim = 2
    else:
        n_low, n, steps, st, ss, w_data, w_ic = 24, 48, 30, 15, 2, 10.0, 5.0          # get_ic_loss hard-codes (48, 48, 48)
        low = torch.cat((0.9 + 0.1 * torch.rand((1, 1, n_low, n_low, n_low), generator=g),
                         0.1 + 0.1 * torch.rand((1, 1, n_low, n_low, n_low), generator=g)), 1)
        eff = list(range(steps))
        m = mod.RCNN(input_channels=2, hidden_channels=2, init_state_low=low, input_kernel_size=5, step=steps,
                     effective_step=eff)
        o = R.OracleRCNN(R.gs3d_cell(), step=steps, effective_step=eff, upscaler=R.OracleUpscaler(3), init_state_low=low)
        ndim = 3
    m.load_state_dict(full)
    o.load_state_dict(m.state_dict())
    sub = (slice(None), slice(None)) + (slice(None, None, ss),) * ndim
    # the truth tensor `truth[::st][sub]` the script compares with: a smooth seeded field in the state's range
    nt = len(range(0, steps, st))
    gt = torch.cat((0.8 + 0.2 * torch.rand((nt, 1) + (n // ss,) * ndim, generator=g),
                    0.2 * torch.rand((nt, 1) + (n // ss,) * ndim, generator=g)), 1)
    mse = torch.nn.MSELoss()
    lg = mod.loss_generator(m.crnn_cell.dt, m.crnn_cell.dx)
    phy = mod.loss_func if case == "gs3d" else mod.loss_gen

    def iteration(model, ic_loss_fn):
        for p in model.parameters():
            p.grad = None
        output, _ = model()
        output = torch.cat(tuple(output), dim=0)
        pred = output[0:-1:st][sub]
        if case == "gs2d":                                       # 2dgs:397-401: 90 % of the observed frames train
            idx = int(pred.shape[0] * 0.9)
            loss_data, loss_valid = mse(pred[:idx], gt[:idx]), mse(pred[idx:], gt[idx:])
        else:                                                    # 3dgs:403
            idx = pred.shape[0]
            loss_data, loss_valid = mse(pred, gt), torch.zeros(())
        loss_ic = ic_loss_fn(model)
        loss_phy = phy(output, lg)
        loss = w_data * loss_data + w_ic * loss_ic
        loss.backward(retain_graph=True)
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        return dict(loss=loss.detach(), loss_data=loss_data.detach(), loss_valid=loss_valid.detach(), loss_ic=loss_ic.detach(),
                    loss_phy=loss_phy.detach(), init_state=model.init_state.detach().clone(), idx=idx), grads

    def oracle_ic_loss(model):                                   # get_ic_loss restated (2dgs:331-338 / 3dgs:325-332)
        target = torch.nn.functional.interpolate(model.init_state_low, (n,) * ndim,
                                                 mode="bicubic" if ndim == 2 else "trilinear")
        return mse(model.UpconvBlock(model.init_state_low), target)

    vr, gr = iteration(m, mod.get_ic_loss)
    vo, go = iteration(o, oracle_ic_loss)
    for k in ("loss", "loss_data", "loss_ic", "init_state"):
        assert torch.equal(vr[k], vo[k]), f"{case}/train_iter: restatement {k} differs from the reference"
    assert sorted(gr) == sorted(go)
    for k in gr:
        assert torch.equal(gr[k], go[k]), f"{case}/train_iter: restatement gradient {k} differs from the reference"
    rec = {"steps": steps, "stride_t": st, "stride_x": ss, "w_data": w_data, "w_ic": w_ic, "idx": vr["idx"],
           "init_state_low": low.numpy(), "gt": gt.numpy(), "init_state": vr["init_state"].numpy(), "n": n}
    for k in ("loss", "loss_data", "loss_valid", "loss_ic", "loss_phy"):
        rec[k] = float(vr[k])
    for k, v in m.state_dict().items():
        rec["state/" + k] = v.numpy()
    for k, v in gr.items():
        rec["grad/" + k] = v.numpy()
    fn = os.path.join(OUT, f"{case}_train_iter.npz")
    np.savez_compressed(fn, **rec)
    print(f"  wrote {os.path.relpath(fn, ROOT)}  ({os.path.getsize(fn)/1024:.0f} KiB)  loss={rec['loss']:.9g} "
          f"data={rec['loss_data']:.6g} ic={rec['loss_ic']:.6g} phy={rec['loss_phy']:.6g}; {len(gr)} gradient tensors")


def run_case(case, big, biggrad=False, longgrad=False):
    mod = import_reference(case)
    torch.set_num_threads(8)
    if longgrad:
        # the reference's own autograd over a LONG horizon at the headline grid (512^2 x 300 steps: a ~25 GB tape,
        # what this container holds; the float64 twin would need twice that and is left to the T = 100 case) -- pins the
        # long-horizon backward to the reference itself, not only to the C oracle
        if case == "gs2d":
            biggrad_case(case, mod, (512, 512), 300, 20, twin=False, tag="longgrad300")
        return
    if biggrad:
        if case == "gs2d":
            biggrad_case(case, mod, (512, 512), 100, 20)
        elif case == "gs3d":
            biggrad_case(case, mod, (128, 128, 128), 20, 5)
        elif case == "lo2d":
            biggrad_case(case, mod, (512, 512), 100, 20)
        return
    if case in ("bur1", "lo1"):
        if not big:
            stage1_case(mod, case)
        return
    if case == "bur3":
        if not big:
            stage3_burgers_case(mod)
        return
    if case == "lo3":
        if not big:
            stage3_lo_case(mod)
        return
    state, _ = ckpt_cell_state(case)
    print(f"[{case}] reference imported; default dtype {torch.get_default_dtype()}")
    if big:
        if case == "gs2d":
            big_case(case, mod, (512, 512), [1, 100, 1000])
        elif case == "gs3d":
            big_case(case, mod, (128, 128, 128), [1, 50, 500])
        else:
            big_case(case, mod, (512, 512), [1, 100, 400])
        return
    if case in ("gs2d", "lo2d"):
        small_case(case, mod, "ckpt", state, (32, 32), 50, [1, 2, 10, 50], 5)
        small_case(case, mod, "fresh", None, (32, 32), 10, [1, 2, 10], 5)
        small_case(case, mod, "ckpt", state, (64, 64), 200, [1, 2, 10, 50, 200], 20)
        if case == "gs2d":
            small_case(case, mod, "ckpt", state, (24, 40), 10, [1, 2, 10], 5)   # non-square
            # BASELINE configs[0]: the reference's own training configuration (2dgs:597-636: 100^2 grid, 200 steps)
            small_case(case, mod, "ckpt", state, (100, 100), 200, [1, 20, 200], 20)
    else:
        small_case(case, mod, "ckpt", state, (16, 16, 16), 50, [1, 2, 10, 50], 5)
        small_case(case, mod, "fresh", None, (16, 16, 16), 10, [1, 2, 10], 5)
        small_case(case, mod, "ckpt", state, (24, 24, 24), 50, [1, 2, 10, 50], 5)
        small_case(case, mod, "ckpt", state, (8, 12, 20), 10, [1, 2, 10], 5)    # non-cubic
    rcnn_harness_case(case, mod)
    if case in ("gs2d", "gs3d"):
        train_iter_case(case, mod)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", choices=list(SCRIPTS))
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--train-iter", action="store_true", help="only the training-iteration fixtures (gs2d, gs3d)")
    ap.add_argument("--longgrad", action="store_true", help="reference gradients at 512^2 x 300 steps (gs2d; ~25 GB of autograd tape)")
    ap.add_argument("--biggrad", action="store_true", help="full-size reference gradients (512^2 x100, 128^3 x20, lo 512^2 x100)")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.train_iter:
        if a.case:
            train_iter_case(a.case, import_reference(a.case))
        else:
            for c in ("gs2d", "gs3d"):
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--case", c, "--train-iter"])
    elif a.longgrad:
        run_case("gs2d", False, longgrad=True)
    elif a.case:
        run_case(a.case, a.big, a.biggrad)
    else:
        for c in (("gs2d", "gs3d", "lo2d") if a.biggrad else SCRIPTS):
            cmd = [sys.executable, os.path.abspath(__file__), "--case", c] + (["--big"] if a.big else []) + \
                  (["--biggrad"] if a.biggrad else [])
            subprocess.check_call(cmd)
