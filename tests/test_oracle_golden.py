"""CPU: the oracle (torch restatement + plain-C restatement) against the golden vectors that
tools/make_golden.py captured from the imported reference."""
import numpy as np
import pytest
import torch

from util import Golden, TOL_GRAD, case_id, data_loss, rel_l2, small_cases, GOLDEN
import os


@pytest.mark.parametrize("fn", small_cases(), ids=case_id)
def test_restatement_reproduces_reference_bitwise(fn):
    g = Golden(fn)
    cell = g.oracle_cell()
    h0 = torch.tensor(g.h0).requires_grad_(True)
    outs = [h0]
    h = h0
    for _ in range(g.steps):
        h, _ = cell(h)
        outs.append(h)
    traj = torch.cat(outs, 0)
    for t in g.keep_t:
        assert np.array_equal(traj[t].detach().numpy(), g.traj(t)), f"frame {t}"
    for name, loss in (("meansq", (traj ** 2).mean()), ("data", data_loss(traj, g.stride_t, g.ndim))):
        assert loss.item() == float(g.z[f"loss_{name}"])
        params = {n: p for n, p in cell.named_parameters() if p.requires_grad}
        grads = torch.autograd.grad(loss, list(params.values()) + [h0], retain_graph=True)
        for (n, _), gr in zip(params.items(), grads[:-1]):
            assert np.array_equal(gr.numpy(), g.grads(name)[n]), n
        assert np.array_equal(grads[-1].numpy(), g.z[f"grad_{name}_h0"])


@pytest.mark.parametrize("fn", small_cases(), ids=case_id)
def test_c_oracle_matches_reference(fn):
    from oracle import pi_oracle as O
    g = Golden(fn)
    P = g.packed()
    traj = O.rollout_fwd(g.h0[0], P, g.hc, g.steps)
    tol = 5e-7 if g.dtype == np.float32 else 1e-14
    for t in g.keep_t:
        assert rel_l2(traj[t], g.traj(t)) < tol, f"frame {t}"
    # dense loss L = mean(traj^2): dL/dtraj = 2 traj / N
    gt = (2.0 * traj / traj.size).astype(g.dtype)
    g0, pg = O.rollout_bwd(traj, gt, P, g.hc)
    assert rel_l2(g0, g.z["grad_meansq_h0"][0]) < TOL_GRAD[g.dtype]
    named = g.named_grads_from_packed(pg)
    ref = g.grads("meansq")
    for n in ref:
        assert rel_l2(named[n], ref[n]) < TOL_GRAD[g.dtype], n
    # sparse strided data loss (cf. train_2drd.py:397-402)
    tt = torch.tensor(traj, requires_grad=True)
    data_loss(tt, g.stride_t, g.ndim).backward()
    g0, pg = O.rollout_bwd(traj, tt.grad.numpy(), P, g.hc)
    assert rel_l2(g0, g.z["grad_data_h0"][0]) < TOL_GRAD[g.dtype]
    named = g.named_grads_from_packed(pg)
    ref = g.grads("data")
    allr = np.concatenate([np.ravel(ref[n]) for n in sorted(ref)])
    allm = np.concatenate([np.ravel(named[n]) for n in sorted(ref)])
    assert rel_l2(allm, allr) < TOL_GRAD[g.dtype]


@pytest.mark.parametrize("fam", ["gs2d", "gs3d", "lo2d"])
def test_rcnn_harness_restatement(fam):
    """a9: upscaler / fixed IC, effective_step membership, second_last_state."""
    from oracle import restatement as R
    z = np.load(os.path.join(GOLDEN, f"{fam}_rcnn_harness.npz"))
    steps, eff = int(z["steps"]), [int(e) for e in z["effective_step"]]
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("state/")}
    if fam == "lo2d":
        m = R.OracleRCNN(R.lo2d_cell(), step=steps, effective_step=eff,
                         init_state=torch.tensor(z["init_state"], dtype=torch.float64), cell_name="rcnn_cell")
    else:
        nd = 2 if fam == "gs2d" else 3
        cell = R.gs2d_cell() if nd == 2 else R.gs3d_cell()
        m = R.OracleRCNN(cell, step=steps, effective_step=eff, upscaler=R.OracleUpscaler(nd),
                         init_state_low=torch.tensor(z["init_state_low"]))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    with torch.no_grad():
        outs, sl = m()
    assert len(outs) == z["outputs"].shape[0]
    assert np.array_equal(torch.cat(outs, 0).numpy(), z["outputs"])
    assert np.array_equal(sl.numpy(), z["second_last_state"])


@pytest.mark.parametrize("fn", small_cases(), ids=case_id)
def test_physics_loss_restatement_equals_reference(fn):
    """The reference's own physics-residual scalar of its trajectory (captured by make_golden.py)."""
    from oracle import restatement as R
    g = Golden(fn)
    traj = R.rollout(g.oracle_cell(), torch.tensor(g.h0), g.steps).detach()
    loss = R.physics_loss_reference(traj, g.family, g.dx, g.dt)
    ref = float(g.z["phy_loss"])
    assert abs(loss.item() - ref) <= 1e-6 * abs(ref), (loss.item(), ref)
