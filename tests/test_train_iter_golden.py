"""One TRAINING iteration of the reference scripts -- ``model()`` + ``torch.cat`` + strided data loss + ``get_ic_loss`` +
``backward()`` (train_2drd.py:393-407, train_3drd.py:399-408) -- captured from the imported reference by
``tools/make_golden.py --train-iter`` (``tests/golden/gs{2,3}d_train_iter.npz``: losses and EVERY gradient the optimizer would
see, the cell's 18 tensors and the IC generator's 6).

Pins on something the reference holds: the upscaler's backward (2D stock layers, 3D the hand-written
contraction + ``percnn_pi_conv3d_k5c8_wgrad_f32``), ``get_ic_loss``, and the data-loss routes of the package --
``RCNN.forward()`` + ``torch.cat``, ``RCNN.observe()``, ``RCNN.loss_mse()`` -- end to end, no route validated only against
another route of this package.
"""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, TOL_GRAD, rel_l2

F32 = np.dtype("float32")


def _load(fam):
    z = np.load(os.path.join(GOLDEN, f"{fam}_train_iter.npz"))
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("state/")}
    grads = {k[5:]: z[k] for k in z.files if k.startswith("grad/")}
    return z, sd, grads


def _iteration(model, z, route, ic_loss):
    """The reference loop's loss composition on `model` (any of: reference restatement, this package's RCNN)."""
    st, ss, ndim = int(z["stride_t"]), int(z["stride_x"]), z["init_state_low"].ndim - 2
    gt = torch.tensor(z["gt"], device=model.init_state_low.device)
    idx = int(z["idx"])
    sub = (slice(None), slice(None)) + (slice(None, None, ss),) * ndim
    mse = torch.nn.functional.mse_loss
    if route == "cat":                                   # the reference's own call pattern (2dgs:393-401)
        output, _ = model()
        output = torch.cat(tuple(output), dim=0)
        pred = output[0:-1:st][sub]
        loss_data = mse(pred[:idx], gt[:idx])
    elif route == "observe":                             # same tensor from ONE autograd node without the dense dL/dtraj
        pred = model.observe(slice(0, -1, st), ss)
        loss_data = mse(pred[:idx], gt[:idx])
    else:                                                # loss_mse: the first idx observed frames
        loss_data = model.loss_mse(gt[:idx], slice(0, idx * st, st), ss)
    loss_ic = ic_loss(model)
    loss = float(z["w_data"]) * loss_data + float(z["w_ic"]) * loss_ic
    loss.backward()
    return loss.detach(), loss_data.detach(), loss_ic.detach()


@pytest.mark.parametrize("fam", ["gs2d", "gs3d"])
def test_restatement_training_iteration_equals_reference(fam):
    """CPU: the oracle's restatement reproduces the captured iteration (bit for bit on the build image; 1e-6 here so that a
    host with another thread count / oneDNN blocking does not fail on summation order)."""
    from oracle import restatement as R
    z, sd, grads = _load(fam)
    nd = 2 if fam == "gs2d" else 3
    steps = int(z["steps"])
    m = R.OracleRCNN(R.gs2d_cell() if nd == 2 else R.gs3d_cell(), step=steps, effective_step=list(range(steps)),
                     upscaler=R.OracleUpscaler(nd), init_state_low=torch.tensor(z["init_state_low"]))
    m.load_state_dict(sd)

    def ic(model):
        n = int(z["n"])
        tgt = torch.nn.functional.interpolate(model.init_state_low, (n,) * nd, mode="bicubic" if nd == 2 else "trilinear")
        return torch.nn.functional.mse_loss(model.UpconvBlock(model.init_state_low), tgt)

    loss, ld, lic = _iteration(m, z, "cat", ic)
    assert abs(loss.item() - float(z["loss"])) <= 1e-6 * abs(float(z["loss"]))
    assert abs(ld.item() - float(z["loss_data"])) <= 1e-6 * abs(float(z["loss_data"]))
    assert abs(lic.item() - float(z["loss_ic"])) <= 1e-6 * abs(float(z["loss_ic"]))
    assert rel_l2(m.init_state.detach().numpy(), z["init_state"]) < 1e-6
    got = {k: p.grad.numpy() for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(grads)
    for k in grads:
        assert rel_l2(got[k], grads[k]) < 2e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("reaction", ["poly", "factored"])
@pytest.mark.parametrize("route", ["cat", "observe", "loss_mse"])
@pytest.mark.parametrize("fam", ["gs2d", "gs3d"])
def test_training_iteration_vs_reference(fam, route, reaction):
    import percnn_amd as pa
    dev = torch.device("cuda:0")
    z, sd, grads = _load(fam)
    nd = 2 if fam == "gs2d" else 3
    steps = int(z["steps"])
    cell = pa.gs2d_cell(reaction=reaction) if nd == 2 else pa.gs3d_cell(reaction=reaction)
    m = pa.RCNN(cell, step=steps, effective_step=list(range(steps)), upscaler=pa.Upscaler(nd),
                init_state_low=torch.tensor(z["init_state_low"], device=dev))
    m.load_state_dict(sd)
    m.to(dev)
    loss, ld, lic = _iteration(m, z, route, lambda model: model.ic_loss())
    # IC generator forward (2D: stock MIOpen layers; 3D: the package's contraction kernels)
    assert rel_l2(m.init_state.detach().cpu().numpy(), z["init_state"]) < 2e-6
    assert abs(lic.item() - float(z["loss_ic"])) <= 5e-6 * abs(float(z["loss_ic"]))
    assert abs(ld.item() - float(z["loss_data"])) <= 1e-5 * abs(float(z["loss_data"]))
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    got = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(grads)
    worst = {}
    for k in grads:
        worst[k] = rel_l2(got[k], grads[k])
    # cell: the package's gradient bar (tests/util.py: 2e-5 in float32); upscaler tensors receive 0.25 / 5.0 x dL_ic/dW plus
    # dL_data/dh0 pulled back through the layers -- the same bar
    bad = {k: v for k, v in worst.items() if not v < TOL_GRAD[F32]}
    assert not bad, bad
