"""Stage-1 Pi-block: 5x5-branch cell on the matrix cores.

CPU part: the C oracle's k-ordered fmaf chain and the torch restatement against the golden vectors captured from the
reference's own Stage-1 scripts + checkpoints (tools/make_golden.py --case bur1 / lo1).
GPU part (through the C-ABI): forward bit-identical to the C oracle, trajectory vs the reference's golden frames,
gradients vs the reference's float32 autograd and vs a float64 evaluation (the float32 reference is itself
1e-6..5e-6 away from float64, so the tolerance for gradients is 2e-5 relative L2).
"""
import glob
import os

import numpy as np
import pytest
import torch

from util import GOLDEN as GOLDEN_DIR, rel_l2

CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*_stage1_*.npz")))
FAMILY = {"bur1": "burgers", "lo1": "lo"}


def _load(fn):
    z = np.load(fn)
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    return z, sd, FAMILY[os.path.basename(fn).split("_")[0]]


def _oracle_block(z, sd):
    from oracle import pi_oracle as O
    nu = float(z["nu_up"])
    cu, cv = float(nu * torch.sigmoid(sd["CA"])), float(nu * torch.sigmoid(sd["CB"]))
    return O.s1_pack({k: v.numpy() for k, v in sd.items()}, float(z["dt"]), cu, cv)


def _rollout(cell, h, T):
    outs = [h]
    for _ in range(T):
        h, _o = cell(h)
        outs.append(h)
    return torch.cat(tuple(outs), dim=0)


def _ids(fn):
    return os.path.basename(fn)[:-4]


def test_golden_present():
    assert len(CASES) == 8


@pytest.mark.parametrize("fn", [c for c in CASES if "100x100" not in c], ids=_ids)
def test_oracles_vs_reference_golden(fn):
    """C oracle (MFMA summation order) and torch restatement reproduce the reference's trajectory; the restatement
    also reproduces the reference's float32 gradients bit for bit (asserted at generation time, re-checked here)."""
    from oracle import pi_oracle as O, restatement as R
    z, sd, fam = _load(fn)
    T = int(z["steps"])
    traj = O.s1_rollout_fwd(z["h0"][0], _oracle_block(z, sd), T)
    cell = R.OracleStage1Cell(fam)
    cell.load_state_dict(sd)
    h = torch.tensor(z["h0"], requires_grad=True)
    tr = _rollout(cell, h, T)
    for t in z["keep_t"]:
        assert rel_l2(traj[t], z[f"traj/{t}"]) < 2e-7
        assert np.array_equal(tr[t].detach().numpy(), np.squeeze(z[f"traj/{t}"]))
    loss = (tr ** 2).mean()
    names = [n for n, p in cell.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, [p for _, p in cell.named_parameters() if p.requires_grad] + [h])
    for n, g in zip(names, grads[:-1]):
        assert np.array_equal(g.numpy(), z["grad_meansq/" + n]), n
    assert np.array_equal(grads[-1].numpy(), z["grad_meansq_h0"])


def test_state_dict_schema_and_block_layout():
    """Host module: reference schema; its differentiable packing equals the oracle's independent packing."""
    from oracle import pi_oracle as O, restatement as R
    import percnn_amd as pa
    z, sd, fam = _load(CASES[0])
    cell, ref = pa.Stage1Cell(fam), R.OracleStage1Cell(fam)
    assert list(cell.state_dict().keys()) == list(ref.state_dict().keys())
    assert all(cell.state_dict()[k].shape == ref.state_dict()[k].shape for k in ref.state_dict())
    assert (cell.dx, cell.dt, cell.nu_up) == (ref.dx, ref.dt, ref.nu_up)
    cell.load_state_dict(sd)
    with torch.no_grad():
        P = cell.param_block()
    assert P.numel() == pa.stage1.NP == O.S1_NP
    assert np.array_equal(P.numpy(), _oracle_block(z, sd))
    with pytest.raises(RuntimeError):
        cell(torch.zeros(1, 2, 16, 16))                 # CPU tensors are refused: no fallback path


def test_whole_model_schema_matches_the_stage1_checkpoints():
    """pa.RCNN(Stage1Cell, upscaler=stage1.Upscaler()) has exactly the key set / shapes of the reference's Stage-1
    checkpoint['model_state_dict'] (both scripts; listed from the shipped checkpoint.pt files)."""
    import percnn_amd as pa
    m = pa.RCNN(pa.Stage1Cell("lo"), step=4, effective_step=[0, 1, 2, 3], upscaler=pa.stage1.Upscaler(),
                init_state_low=torch.zeros(1, 2, 8, 8))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    exp = {"UpconvBlock.up0.weight": (2, 16, 5, 5), "UpconvBlock.up0.bias": (16,), "UpconvBlock.out.weight": (2, 16, 1, 1),
           "UpconvBlock.out.bias": (2,), "UpconvBlock.convnet.0.weight": (2, 16, 5, 5), "UpconvBlock.convnet.0.bias": (16,),
           "UpconvBlock.convnet.2.weight": (2, 16, 1, 1), "UpconvBlock.convnet.2.bias": (2,),
           "crnn_cell.CA": (), "crnn_cell.CB": (), "crnn_cell.W_laplace.weight": (1, 1, 5, 5)}
    for s in "uv":
        for k in (1, 2, 3):
            exp[f"crnn_cell.Wh{k}_{s}.weight"] = (16, 2, 5, 5)
            exp[f"crnn_cell.Wh{k}_{s}.bias"] = (16,)
        exp[f"crnn_cell.Wh4_{s}.weight"] = (1, 16, 1, 1)
        exp[f"crnn_cell.Wh4_{s}.bias"] = (1,)
    assert got == exp


# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("fn", CASES, ids=_ids)
def test_forward_bitwise_vs_c_oracle_and_golden(fn, dev):
    from oracle import pi_oracle as O
    import percnn_amd as pa
    z, sd, fam = _load(fn)
    T = min(int(z["steps"]), 20)
    P = _oracle_block(z, sd)
    traj_o = O.s1_rollout_fwd(z["h0"][0], P, T)
    traj = torch.empty((T + 1,) + z["h0"].shape[1:], dtype=torch.float32, device=dev)
    traj[0] = torch.tensor(z["h0"][0], device=dev)
    pa.stage1.rollout_fwd_(traj, torch.tensor(P, device=dev))
    assert np.array_equal(traj.cpu().numpy(), traj_o)
    # single steps through the step entry point, same bits
    out = pa.stage1.step_fwd(traj[3].contiguous(), torch.tensor(P, device=dev))
    assert np.array_equal(out.cpu().numpy(), traj_o[4])


@pytest.mark.gpu
@pytest.mark.parametrize("fn", CASES, ids=_ids)
def test_module_rollout_and_gradients_vs_reference_golden(fn, dev):
    import percnn_amd as pa
    z, sd, fam = _load(fn)
    T = int(z["steps"])
    cell = pa.Stage1Cell(fam).to(dev)
    cell.load_state_dict(sd)
    h0 = torch.tensor(z["h0"], device=dev, requires_grad=True)
    traj = cell.rollout(h0, T)
    for t in z["keep_t"]:
        assert rel_l2(traj[t].detach().cpu().numpy(), np.squeeze(z[f"traj/{t}"])) < 5e-7
    assert rel_l2(traj[-1].detach().cpu().numpy(), np.squeeze(z["traj64_last"])) < 5e-7
    loss = (traj ** 2).mean()
    assert abs(loss.item() - float(z["loss_meansq"])) < 1e-6 * abs(float(z["loss_meansq"]))
    loss.backward()
    worst = 0.0
    for n, p in cell.named_parameters():
        if not p.requires_grad:
            continue
        g = p.grad.detach().cpu().numpy()
        e32, e64 = rel_l2(g, z["grad_meansq/" + n]), rel_l2(g, z["grad64_meansq/" + n])
        worst = max(worst, e64)
        assert e32 < 2e-5 and e64 < 2e-5, (n, e32, e64)
    assert rel_l2(h0.grad.cpu().numpy(), z["grad64_meansq_h0"]) < 2e-5
    print(f"{_ids(fn)}: worst parameter-gradient error vs float64 {worst:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 16), (22, 26), (9, 13), (64, 48), (8, 8), (8, 40), (36, 12), (28, 100), (11, 8)])
def test_step_backward_random_vs_float64_autograd(shape, dev):
    """One step with random weights / state / upstream gradient, sparse frame mask included."""
    from oracle import restatement as R
    import percnn_amd as pa
    torch.manual_seed(3)
    ref = R.OracleStage1Cell("lo", dtype=torch.float64)
    for p in ref.parameters():
        if p.requires_grad and p.dim() > 0:
            p.data = torch.randn_like(p) * 0.3
    cell = pa.Stage1Cell("lo").to(dev)
    cell.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    h = torch.randn(1, 2, *shape, dtype=torch.float64)
    T = 3
    hr = h.clone().requires_grad_(True)
    tr = _rollout(ref, hr, T)
    w = torch.randn_like(tr)
    w[1] = 0                                              # a frame without gradient
    (tr * w).sum().backward()
    with torch.no_grad():
        P = cell.param_block().contiguous()
    traj = torch.empty((T + 1, 2) + shape, dtype=torch.float32, device=dev)
    traj[0] = h[0].float().to(dev)
    pa.stage1.rollout_fwd_(traj, P)
    assert rel_l2(traj.cpu().numpy(), tr.detach().numpy()) < 1e-6
    # both hand-over formats of the sweep (footprint tiles need H, W % 4 == 0; per-tap planes work for any shape)
    for mask, etile in ((None, 1), ([True, False, True, True], 1), (None, 0)):
        pa.stage1.set_option("etile", etile)
        try:
            g0, pg = pa.stage1.rollout_bwd(traj, w.float().to(dev).contiguous(), P, frame_mask=mask)
        finally:
            pa.stage1.set_option("etile", 1)
        assert rel_l2(g0.cpu().numpy(), hr.grad[0].numpy()) < 1e-5
        # map the block gradient back through the packing with stock autograd and compare per tensor
        Pg = cell.param_block()
        Pg.backward(pg.float())
        for (n, p), (_, q) in zip(cell.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                assert rel_l2(p.grad.cpu().numpy(), q.grad.numpy()) < 2e-5, n
        cell.zero_grad()


@pytest.mark.gpu
def test_edge_cases(dev):
    import percnn_amd as pa
    cell = pa.Stage1Cell("burgers").to(dev)
    with torch.no_grad():
        P = cell.param_block().contiguous()
    # T = 0: trajectory is the initial state, gradient passes through
    traj = torch.rand(1, 2, 16, 16, device=dev)
    pa.stage1.rollout_fwd_(traj, P)
    g = torch.rand_like(traj)
    g0, pg = pa.stage1.rollout_bwd(traj, g, P)
    assert torch.equal(g0, g[0]) and float(pg.abs().max()) == 0.0
    # invalid shapes / wrong block length are refused
    with pytest.raises(RuntimeError):
        pa.stage1.step_fwd(torch.rand(2, 4, 4, device=dev), P)
    with pytest.raises(RuntimeError):
        pa.stage1.step_fwd(torch.rand(2, 16, 16, device=dev), P[:-1].contiguous())
    with pytest.raises(RuntimeError):
        pa.stage1.step_fwd(torch.rand(2, 16, 16, device=dev, dtype=torch.float64), P)
    # translation equivariance of the periodic wrap
    h = torch.rand(2, 24, 20, device=dev)
    out = pa.stage1.step_fwd(h, P)
    out_s = pa.stage1.step_fwd(torch.roll(h, (5, 7), (1, 2)).contiguous(), P)
    assert torch.equal(out_s, torch.roll(out, (5, 7), (1, 2)))


@pytest.mark.gpu
def test_rcnn_wrapper_drives_the_stage1_cell(dev):
    """pa.RCNN + Stage1Cell: the reference's RCNN.forward contract (list of frames, second_last_state) on the fused
    Stage-1 rollout; same gradients as the one-tensor path."""
    import percnn_amd as pa
    torch.manual_seed(2)
    cell = pa.Stage1Cell("burgers").to(dev)
    up = pa.stage1.Upscaler().to(dev)
    low = torch.rand(1, 2, 12, 12, device=dev)
    steps = 9
    m = pa.RCNN(cell, step=steps, effective_step=list(range(steps)), upscaler=up, init_state_low=low)
    outs, sl = m()
    assert len(outs) == steps + 1 and outs[0].shape == (1, 2, 24, 24)
    traj = m.trajectory()
    assert torch.equal(torch.cat(tuple(outs), 0), traj) and torch.equal(sl, traj[steps - 1:steps])
    # round 5: the reference's own torch.cat(tuple(output), dim=0) (bur1:607) returns the trajectory buffer, no copy
    assert torch.cat(tuple(outs), dim=0) is outs.stacked and outs.stacked.data_ptr() == outs[0].data_ptr()
    assert torch.cat(tuple(f.as_subclass(torch.Tensor) for f in outs), 0).data_ptr() != outs.stacked.data_ptr()
    w = torch.randn_like(traj)
    params = [p for p in m.parameters() if p.requires_grad]          # W_laplace is frozen
    g1 = torch.autograd.grad((torch.cat(tuple(outs), 0) * w).sum() + (sl ** 2).sum(), params, allow_unused=True)
    g2 = torch.autograd.grad((traj * w).sum() + (traj[steps - 1] ** 2).sum(), params, allow_unused=True)
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None)
        if a is not None:
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
    # observation operator (the data loss's operand) from one autograd node == cat + slice
    ref = torch.cat(tuple(m()[0]), 0)[0:-1:2][:, :, ::2, ::2]
    pred = m.observe(slice(0, -1, 2), 2)
    assert torch.equal(pred, ref)
    truth = torch.rand_like(ref)
    g3 = torch.autograd.grad(torch.nn.functional.mse_loss(pred, truth), params, allow_unused=True)
    g4 = torch.autograd.grad(torch.nn.functional.mse_loss(ref, truth), params, allow_unused=True)
    for a, b in zip(g3, g4):
        if a is not None:
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-5



# ---- resident rollouts (round 5) ------------------------------------------------------------------------------------
def _s1_problem(dev, shape, T, seed=0, family="burgers"):
    import percnn_amd as pa
    torch.manual_seed(seed)
    cell = pa.Stage1Cell(family).to(dev)
    with torch.no_grad():
        for p in cell.parameters():
            if p.requires_grad and p.dim() > 0:
                p.mul_(0.5)
        P = cell.param_block().contiguous()
    traj = torch.empty((T + 1, 2) + shape, dtype=torch.float32, device=dev)
    traj[0] = torch.rand((2,) + shape, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 1)) * 0.5
    return traj, P


@pytest.mark.gpu
@pytest.mark.parametrize("shape,T", [((100, 100), 40), ((32, 32), 9), ((64, 48), 23), ((8, 8), 12), ((128, 100), 17)])
def test_resident_rollouts_equal_launch_per_step(shape, T, dev):
    """s1_fwd_persist_kernel / s1_adj_persist_kernel (whole rollouts as ONE launch of resident waves, hand-over through data-tagged
    granules): trajectory, dL/dh0 and every adjoint-dependent parameter gradient BIT for bit those of one launch per step (same
    device functions in the same order); dense dL/dtraj and frame masks; the C oracle on the trajectory; twice in a row."""
    from oracle import pi_oracle as O
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    traj, P = _s1_problem(dev, shape, T)
    ref = traj.clone()
    s0 = _lib.persist_status()
    n0 = s0["launches"]
    try:
        pa.stage1.set_option("persist", 0)
        pa.stage1.rollout_fwd_(ref, P)
        assert _lib.persist_status()["launches"] == n0
    finally:
        pa.stage1.set_option("persist", 1)
    pa.stage1.rollout_fwd_(traj, P)
    s1 = _lib.persist_status()
    assert s1["launches"] == n0 + 1 and s1["aborts"] == s0["aborts"]
    assert torch.isfinite(ref).all() and torch.equal(traj, ref)
    if shape[0] * shape[1] <= 64 * 48:
        assert np.array_equal(traj.cpu().numpy(), O.s1_rollout_fwd(traj[0].cpu().numpy(), P.cpu().numpy(), T))
    again = traj.clone()
    again[1:] = float("nan")
    pa.stage1.rollout_fwd_(again, P)
    assert torch.equal(again, ref)
    g = torch.randn(traj.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) / traj[0].numel()
    for mask in (None, [t % 3 != 1 for t in range(T + 1)], [t == T for t in range(T + 1)]):
        try:
            pa.stage1.set_option("persist", 0)
            r0, rg = pa.stage1.rollout_bwd(traj, g, P, frame_mask=mask)
        finally:
            pa.stage1.set_option("persist", 1)
        n1 = _lib.persist_status()["launches"]
        a0, ag = pa.stage1.rollout_bwd(traj, g, P, frame_mask=mask)
        assert _lib.persist_status()["launches"] == n1 + 1 and _lib.persist_status()["aborts"] == s1["aborts"]
        assert torch.isfinite(a0).all() and torch.equal(a0, r0)
        assert torch.equal(ag, rg)                          # the weight-gradient kernel reads the same adjoint frames
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_resident_rollouts_keep_to_their_shapes(dev):
    """ragged grids (H or W not a multiple of 4), short rollouts and grids with more tasks than the device holds at once keep
    one launch per step -- and give the same results as ever."""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    for shape, T in (((30, 32), 12), ((32, 32), 5), ((512, 512), 9)):
        traj, P = _s1_problem(dev, shape, T)
        n0 = _lib.persist_status()["launches"]
        pa.stage1.rollout_fwd_(traj, P)
        g = torch.randn(traj.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) / traj[0].numel()
        g0, pg = pa.stage1.rollout_bwd(traj, g, P)
        assert _lib.persist_status()["launches"] == n0, (shape, T)
        assert torch.isfinite(traj).all() and torch.isfinite(g0).all() and torch.isfinite(pg).all()


@pytest.mark.gpu
def test_resident_rollouts_abort_and_fall_back(dev):
    """CUs held by another kernel: not every task of the resident launch is on the device, the waves that are give up at their
    first hand-over (persist_first_timeout_ms), and the SAME call recomputes launch by launch -- bit-identical results; the
    device keeps the launch-per-step path until persist_reset (state shared with the 2D resident kernels)."""
    import percnn_amd as pa
    from percnn_amd import _lib
    from test_safety_gpu import _hog
    pa.set_option("persist_reset", 1)
    shape, T = (100, 100), 24
    traj, P = _s1_problem(dev, shape, T)
    ref = traj.clone()
    pa.stage1.rollout_fwd_(ref, P)
    g = torch.randn(traj.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) / traj[0].numel()
    r0, rg = pa.stage1.rollout_bwd(ref, g, P)
    torch.cuda.synchronize()
    assert not _lib.persist_status()["disabled_on_current_device"]
    pa.set_option("persist_first_timeout_ms", 20)
    try:
        for which in ("forward", "sweep"):
            _hog(216, 150 * 1024, 1500, dev)                # 40 free CUs: < 314 workgroups of 12 / 26 KB
            s1 = _lib.persist_status()
            if which == "forward":
                out = traj.clone()
                out[1:] = float("nan")
                pa.stage1.rollout_fwd_(out, P)
                torch.cuda.synchronize()
                assert torch.equal(out, ref)
            else:
                a0, ag = pa.stage1.rollout_bwd(ref, g, P)
                torch.cuda.synchronize()
                assert torch.equal(a0, r0) and torch.equal(ag, rg)
            s2 = _lib.persist_status()
            assert s2["launches"] == s1["launches"] + 1 and s2["aborts"] == s1["aborts"] + 1 and s2["disabled_on_current_device"]
            n = s2["launches"]
            again = traj.clone()
            pa.stage1.rollout_fwd_(again, P)                # disabled: launch per step, no new resident launch
            assert _lib.persist_status()["launches"] == n and torch.equal(again, ref)
            torch.cuda.synchronize()
            pa.set_option("persist_reset", 1)
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_first_timeout_ms", 100)
        pa.set_option("persist_reset", 1)
    out = traj.clone()
    n = _lib.persist_status()["launches"]
    pa.stage1.rollout_fwd_(out, P)
    assert _lib.persist_status()["launches"] == n + 1 and torch.equal(out, ref)
