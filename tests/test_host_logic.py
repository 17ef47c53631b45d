"""CPU: host-side logic of the product package and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from util import GOLDEN, Golden, case_id, small_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if fn.endswith(".h"):
            src = open(os.path.join(inc, fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names += re.findall(r"\b(percnn_pi_\w+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import percnn_amd
    so = percnn_amd.build()
    L = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/*.h but not exported"
    from percnn_amd import _lib as _l
    assert percnn_amd.lib().percnn_pi_abi_version() == _l.ABI_VERSION == 3
    for hc in (2, 4, 8, 16):
        assert percnn_amd.lib().percnn_pi_param_count(hc) == 16 + 2 * (10 * hc + 1) == percnn_amd.param_count(hc)
    assert percnn_amd.lib().percnn_pi_param_count(0) == 36          # pre-contracted polynomial block
    shape = (ctypes.c_int64 * 2)(512, 512)
    assert percnn_amd.lib().percnn_pi_bwd_workspace_bytes(8, 2, shape, 4) > 2 * 2 * 512 * 512 * 4
    assert percnn_amd.lib().percnn_pi_bwd_workspace_bytes(8, 4, shape, 4) == 0       # bad ndim
    assert percnn_amd.lib().percnn_pi_set_option(b"nonsense", 1) == -1
    assert percnn_amd.lib().percnn_pi_set_option(b"bwd_cpl", 0) == -1
    assert percnn_amd.lib().percnn_pi_set_option(b"bwd_cpl", 2) == 0
    L = percnn_amd.lib()                                   # round 4: the resident forward's switches
    assert L.percnn_pi_set_option(b"fwd_persist", 0) == 0 and L.percnn_pi_set_option(b"fwd_persist", 1) == 0
    assert L.percnn_pi_set_option(b"fwd_persist_per_cu", 3) == -1 and L.percnn_pi_set_option(b"fwd_persist_per_cu", 0) == -1
    assert L.percnn_pi_set_option(b"fwd_persist_per_cu", 2) == 0 and L.percnn_pi_set_option(b"fwd_persist_per_cu", 1) == 0


def test_argument_errors_do_not_need_a_gpu():
    """Validation happens before any launch: bad arguments return PERCNN_PI_EINVAL."""
    import percnn_amd
    L = percnn_amd.lib()
    shape = (ctypes.c_int64 * 2)(8, 8)
    assert L.percnn_pi_step_fwd_f32(None, None, None, 8, 2, shape, None) == -1
    assert L.percnn_pi_step_fwd_f32(1, 2, 3, 8, 5, shape, None) == -1
    assert L.percnn_pi_step_fwd_f32(1, 2, 3, -7, 2, shape, None) == -1          # -1 = advective block, 0 = polynomial block
    bad = (ctypes.c_int64 * 2)(1, 8)
    assert L.percnn_pi_step_fwd_f64(1, 2, 3, 4, 2, bad, None) == -1
    assert L.percnn_pi_rollout_fwd_f32(1, 2, 8, 2, shape, -1, None) == -1
    assert L.percnn_pi_step_bwd_f32(1, 2, None, 3, 4, None, 0, 5, 8, 2, shape, None) == -2   # no workspace
    # 32-bit addressing limit of the direct kernels: a 2D field beyond 4 GiB per species is refused
    # before any launch with its own code (PERCNN_PI_ETOOLARGE = -3; was hipErrorInvalidValue), it is not silently mis-addressed
    huge = (ctypes.c_int64 * 2)(70000, 70000)
    assert L.percnn_pi_step_fwd_f32(16, 32, 48, 0, 2, huge, None) == -3
    assert L.percnn_pi_step_fwd_opt_f32(16, 32, 48, 0, 2, huge, b"tile=0", None) == -3
    # per-call option strings are validated before anything else
    assert L.percnn_pi_rollout_fwd_opt_f32(1, 2, 8, 2, shape, 0, b"tile_k=4,skip_wgrad=1", None) == 0     # T = 0
    for bad in (b"nonsense=1", b"tile_k=3", b"tile_k", b"=4", b"tile_k=4;tile=0", b"tile_k=x"):
        assert L.percnn_pi_rollout_fwd_opt_f32(1, 2, 8, 2, shape, 0, bad, None) == -1, bad
        assert L.percnn_pi_step_fwd_opt_f64(1, 2, 3, 4, 2, shape, bad, None) == -1, bad
    # ... and never leak into the process defaults: a later call without overrides still sees tile_k = 4
    assert L.percnn_pi_rollout_fwd_opt_f32(1, 2, 8, 2, shape, 0, b"tile_k=2", None) == 0
    assert L.percnn_pi_set_option(b"tile_k", 4) == 0
    # native slab rollouts: a slab thinner than the exchange width would forward halo planes as data
    thin = (ctypes.c_int64 * 2)(2, 8)
    assert L.percnn_pi_slab_rollout_fwd_f32(1, 2, 8, 2, thin, 4, 3, None, 0, None) == -1
    assert L.percnn_pi_slab_rollout_fwd_f32(1, 2, 8, 2, thin, 2, 0, None, 0, None) == 0       # T = 0: nothing to do
    # Stage-1 block (include/percnn_pi_stage1.h)
    assert L.percnn_pi_s1_param_count() == 5042 == percnn_amd.stage1.NP
    assert L.percnn_pi_s1_step_fwd_f32(None, None, None, shape, None) == -1
    assert L.percnn_pi_s1_step_fwd_f32(1, 1, 3, shape, None) == -1                 # aliasing
    small = (ctypes.c_int64 * 2)(4, 8)
    assert L.percnn_pi_s1_step_fwd_f32(1, 2, 3, small, None) == -1                 # H < 8
    assert L.percnn_pi_s1_rollout_bwd_workspace_bytes(small, 3) == 0
    assert L.percnn_pi_s1_rollout_bwd_workspace_bytes(shape, 3) > 4 * 2 * 64 * 4
    assert L.percnn_pi_s1_rollout_bwd_f32(1, 2, None, 3, 4, None, 0, 5, shape, 3, None) == -2
    assert L.percnn_pi_s1_set_option(b"nonsense", 1) == -1


@pytest.mark.parametrize("reaction", ["factored", "poly"])
@pytest.mark.parametrize("fn", small_cases(), ids=case_id)
def test_param_block_matches_oracle_layout(fn, reaction):
    """Factored block: bit-equal.  Polynomial block: the device-side einsum contraction against the
    oracle's plain triple-loop expansion (float64 summation order differs -> 1 ulp of the dtype)."""
    g = Golden(fn)
    cell = g.product_cell("cpu", reaction)
    P = cell.param_block().detach().numpy()
    Po = g.packed(reaction)
    assert len(P) == len(Po) == (36 if reaction == "poly" else 16 + 2 * (10 * g.hc + 1))
    used = np.ones(len(P), bool)
    if g.ndim == 2:
        used[12:16] = False
    assert P.dtype == Po.dtype == g.dtype
    assert np.array_equal(P[:12], Po[:12])
    if reaction == "factored":
        assert np.array_equal(P[used], Po[used])
    else:
        np.testing.assert_allclose(P[16:], Po[16:], rtol=4 * np.finfo(g.dtype).eps,
                                   atol=4 * np.finfo(g.dtype).eps * np.abs(Po[16:]).max())   # cancelled (true-zero) terms


@pytest.mark.parametrize("fn", small_cases()[::3], ids=case_id)
def test_contraction_chain_rule_matches_independent_expansion(fn):
    """autograd of contract_block == the oracle's hand-written multilinear chain rule dL/dc -> dL/dWh*."""
    from oracle import pi_oracle as O
    g = Golden(fn)
    cell = g.product_cell("cpu", "poly")
    Q = cell.param_block()
    qg = np.random.RandomState(0).randn(36)
    qg[0] = 0
    qg[3:16] = 0
    (Q.double() * torch.tensor(qg)).sum().backward()
    ref = O.poly_grads_to_params(qg, g.sd)
    for n, r in ref.items():
        if "." not in n:
            continue
        mod, attr = n.split(".")
        got = getattr(getattr(cell, mod), attr).grad.numpy().astype(np.float64)
        assert np.abs(got - r).max() <= 1e-6 * (np.abs(r).max() + 1e-30), n


def test_param_block_is_differentiable_to_named_parameters():
    import percnn_amd as pa
    cell = pa.gs2d_cell(4, reaction="factored")
    P = cell.param_block()
    w = torch.arange(P.numel(), dtype=P.dtype)
    (P * w).sum().backward()
    assert cell.CA.grad is not None and cell.CA.grad.abs() > 0
    assert cell.W_laplace.weight.grad is None
    base = 16
    assert torch.equal(cell.Wh1_u.weight.grad.reshape(4, 2), torch.stack(
        [torch.tensor([w[base + 10 * j], w[base + 10 * j + 1]]) for j in range(4)]))
    assert cell.Wh4_v.bias.grad.item() == w[16 + 41 + 40].item()


@pytest.mark.parametrize("fam", ["gs2d", "gs3d", "lo2d"])
def test_state_dict_schema_equals_reference(fam):
    """Keys, order, shapes and dtypes of the drop-in RCNN equal the reference module's
    (captured state_dict of the shipped checkpoint)."""
    import percnn_amd as pa
    z = np.load(os.path.join(GOLDEN, f"{fam}_rcnn_harness.npz"))
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("state/")}
    if fam == "lo2d":
        m = pa.RCNN(pa.lo2d_cell(), step=2, effective_step=[0, 1], init_state=torch.zeros(1, 2, 8, 8, dtype=torch.float64),
                    cell_name="rcnn_cell")
    else:
        nd = 2 if fam == "gs2d" else 3
        m = pa.RCNN(pa.gs2d_cell() if nd == 2 else pa.gs3d_cell(), step=2, effective_step=[0, 1],
                    upscaler=pa.Upscaler(nd), init_state_low=torch.zeros((1, 2) + (4,) * nd))
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())
    for k in sd:
        assert mine[k].shape == sd[k].shape and mine[k].dtype == sd[k].dtype, k
    m.load_state_dict(sd)     # "All keys matched"


def test_fresh_cell_deterministic_scalars():
    """CA/CB = (rand-0.5)*2 under np.random.seed(1234) (train_2drd.py:60-62)."""
    import percnn_amd as pa
    c = pa.gs2d_cell()
    assert abs(c.CA.item() - (-0.6169611)) < 1e-6 and abs(c.CB.item() - 0.2442175) < 1e-6
    w = c.W_laplace.weight
    assert w.requires_grad is False
    assert w[0, 0, 2, 2].item() == -50000.0 and w[0, 0, 2, 1].item() == np.float32(13333.333984375)
    z = np.load(os.path.join(GOLDEN, "gs2d_fresh_32x32.npz"))
    assert np.array_equal(w.numpy(), z["param/W_laplace.weight"])
    z3 = np.load(os.path.join(GOLDEN, "gs3d_fresh_16x16x16.npz"))
    assert np.array_equal(pa.gs3d_cell().W_laplace.weight.numpy(), z3["param/W_laplace.weight"])
    zl = np.load(os.path.join(GOLDEN, "lo2d_fresh_32x32.npz"))
    assert np.array_equal(pa.lo2d_cell().W_laplace.weight.numpy(), zl["param/W_laplace.weight"])


def test_no_cpu_fallback_and_bad_stencil_rejected():
    import percnn_amd as pa
    c = pa.gs2d_cell()
    with pytest.raises(RuntimeError, match="no CPU path"):
        c(torch.zeros(1, 2, 8, 8))
    with torch.no_grad():
        c.W_laplace.weight[0, 0, 0, 0] = 1.0           # off the star
    with pytest.raises(ValueError, match="star"):
        c.param_block()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "percnn_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_stage3_cell_schema_and_block():
    """Stage-3 lambda-omega cell -- parameter names/order of the reference, coefficient block."""
    import percnn_amd as pa
    z = np.load(os.path.join(GOLDEN, "lo3_stage3_32x32.npz"))
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    cell = pa.Stage3LambdaOmegaCell()
    assert list(cell.state_dict().keys()) == list(sd.keys())
    cell.load_state_dict(sd)
    Q = cell.param_block()
    assert Q.shape == (36,) and Q.dtype == torch.float64
    assert Q[0].item() == 0.0125 and Q[1].item() == sd["nu_u"].item() and Q[3].item() == -5.0 / 0.2 ** 2
    assert Q[16 + 1].item() == sd["C1_u"].item() and Q[26 + 1].item() == sd["C6_v"].item() and Q[26 + 2].item() == sd["C1_v"].item()
    Q.sum().backward()
    assert all(getattr(cell, n).grad is not None for n in pa.Stage3LambdaOmegaCell.INIT)


def test_stage3_burgers_cell_schema_and_block():
    """Stage-3 Burgers cell -- reference parameter names/order; advective block == oracle's."""
    import percnn_amd as pa
    from oracle import pi_oracle as O
    z = np.load(os.path.join(GOLDEN, "bur3_stage3_32x32.npz"))
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    cell = pa.Stage3BurgersCell()
    assert list(cell.state_dict().keys()) == list(sd.keys())
    cell.load_state_dict(sd)
    A = cell.param_block()
    assert A.shape == (60,) and A.dtype == torch.float64
    Ao = O.pack_burgers_stage3({k: v.numpy() for k, v in sd.items()}, float(z["dx"]), float(z["dt"]))
    used = np.ones(60, bool)
    used[12:16] = False                                    # unused third-axis Laplacian slots
    assert np.array_equal(A.detach().numpy()[used], Ao[used])
    A.sum().backward()
    assert all(getattr(cell, n).grad is not None for n in pa.Stage3BurgersCell.INIT)
    assert pa.lib().percnn_pi_param_count(-1) == 60


@pytest.mark.parametrize("name", ["lo3_stage3_32x32.npz", "lo3_stage3_24x40.npz", "bur3_stage3_32x32.npz", "bur3_stage3_24x40.npz"])
def test_stage3_forward_rk4_vs_reference(name):
    """`forward_rk4` of the Stage-3 cells (lo3:154-201, bur3:159-206: defined in the reference, never called by its scripts):
    five RK4 steps and the gradients of mean(h_5^2) against vectors produced by the imported reference's own method
    (tools/make_golden.py: stage3_rk4_vectors); the oracle restatement and the drop-in cell, both on stock tensor operations."""
    import percnn_amd as pa
    from oracle import restatement as R
    z = np.load(os.path.join(GOLDEN, name))
    sd = {k[6:]: torch.tensor(z[k]) for k in z.files if k.startswith("param/")}
    lo = name.startswith("lo3")
    for cell in ((R.OracleStage3LOCell if lo else R.OracleStage3BurgersCell)(), (pa.Stage3LambdaOmegaCell if lo else pa.Stage3BurgersCell)()):
        cell.load_state_dict(sd)
        h0 = torch.tensor(z["h0"], requires_grad=True)
        h, n = h0, int(z["rk4_steps"])
        for t in range(1, n + 1):
            h, _ = cell.forward_rk4(h)
            if f"rk4/{t}" in z.files:
                assert np.abs(h.detach().numpy() - z[f"rk4/{t}"]).max() <= 1e-14 * max(1.0, np.abs(z[f"rk4/{t}"]).max()), (type(cell).__name__, t)
        loss = (h ** 2).mean()
        assert abs(loss.item() - float(z["rk4_loss_meansq"])) <= 1e-14
        names = [k[len("rk4_grad_meansq/"):] for k in z.files if k.startswith("rk4_grad_meansq/")]
        g = torch.autograd.grad(loss, [getattr(cell, k) for k in names] + [h0])
        for k, gi in zip(names, g[:-1]):
            assert abs(gi.item() - float(z["rk4_grad_meansq/" + k])) <= 1e-12 * max(1.0, abs(float(z["rk4_grad_meansq/" + k]))), k
        assert np.abs(g[-1].numpy() - z["rk4_grad_meansq_h0"]).max() <= 1e-12 * max(1e-30, np.abs(z["rk4_grad_meansq_h0"]).max())


def test_frame_gradient_assembly_zero_copy_and_masked():
    """RCNN.forward hands back T+1 frames; their gradients come back as slices of one buffer when the caller cats
    them (zero-copy), or individually / partly missing (copied, rest masked)."""
    from percnn_amd import functional as Fp
    traj = torch.zeros(6, 2, 4, 4)
    big = torch.arange(2 * 6 * 2 * 4 * 4, dtype=torch.float32).reshape(12, 2, 4, 4)
    dense = big[3:9]                                        # a gradient buffer that does not start at its storage's origin
    grads = tuple(dense[k:k + 1] for k in range(6))         # what CatBackward produces
    g, mask = Fp._assemble_frame_grads(grads, tuple(range(6)), traj)
    assert mask is None and g.data_ptr() == dense.data_ptr() and torch.equal(g, dense)
    # the last frame would run past the end of the storage -> no aliasing, falls back to copies
    tail = tuple(big[7 + k:8 + k] for k in range(5)) + (torch.ones(1, 2, 4, 4),)
    g, mask = Fp._assemble_frame_grads(tail, tuple(range(6)), traj)
    assert mask == [True] * 6 and torch.equal(g[:5], big[7:12]) and torch.equal(g[5], torch.ones(2, 4, 4))
    # sparse frames, one missing gradient, one frame listed twice (second_last_state): summed
    frames = (0, 2, 5, 4, 4)
    gr = (torch.full((1, 2, 4, 4), 1.0), None, torch.full((1, 2, 4, 4), 3.0), torch.full((1, 2, 4, 4), 4.0),
          torch.full((1, 2, 4, 4), 0.5))
    g, mask = Fp._assemble_frame_grads(gr, frames, traj)
    assert mask == [True, False, False, False, True, True]
    assert float(g[0].mean()) == 1.0 and float(g[5].mean()) == 3.0 and float(g[4].mean()) == 4.5
    # what RCNN.forward really produces for step >= 2: the dense list FOLLOWED by second_last_state (frame T-1).
    # Extra frame unused (gradient None) -> still the zero-copy view; extra frame used -> one copy, summed, autograd's
    # buffer left untouched
    frames = tuple(range(6)) + (4,)
    g, mask = Fp._assemble_frame_grads(grads + (None,), frames, traj)
    assert mask is None and g.data_ptr() == dense.data_ptr()
    before = dense.clone()
    g, mask = Fp._assemble_frame_grads(grads + (torch.full((1, 2, 4, 4), 0.25),), frames, traj)
    assert mask is None and g.data_ptr() != dense.data_ptr() and torch.equal(dense, before)
    assert torch.equal(g[4], dense[4] + 0.25) and torch.equal(g[:4], dense[:4]) and torch.equal(g[5], dense[5])


def test_dt_change_after_first_use_is_honoured():
    """The reference reads self.dt at every step (train_2drd.py:117); the cached device scalar must follow it."""
    import percnn_amd as pa
    cell = pa.gs2d_cell(2, reaction="factored")
    assert float(cell.param_block()[0]) == 0.5
    cell.dt = 0.25
    assert float(cell.param_block()[0]) == 0.25
    cell.reaction = "poly"
    cell.dt = 0.125
    assert float(cell.param_block()[0]) == 0.125


def _cubic_well_block(a, k, dt, hc=8):
    """Stable, deliberately ill-conditioned block: r_s = -k (x_s - a)^3 - k (x_s - a)(x_o - a)^2 (gradient flow of a
    quartic well centred at a).  The factored form subtracts a BEFORE multiplying; the expanded cubic cancels
    monomials of size a^3 against each other."""
    P = np.zeros(16 + 2 * (10 * hc + 1))
    P[0], P[1], P[2], P[3] = dt, 0.02, 0.01, -5.0
    P[4:8] = P[8:12] = (-1 / 12, 4 / 3, 4 / 3, -1 / 12)
    for s in range(2):
        W = P[16 + s * (10 * hc + 1):16 + (s + 1) * (10 * hc + 1)]
        me, ot = s, 1 - s
        for kk in range(3):
            W[3 * kk + me], W[3 * kk + 2] = 1.0, -a
        W[9] = -k
        c1 = W[10:20]
        c1[0 + me], c1[2] = 1.0, -a
        c1[3 + ot], c1[5] = 1.0, -a
        c1[6 + ot], c1[8] = 1.0, -a
        c1[9] = -k
    return P


@pytest.mark.parametrize("a", [0.0, 2.0, 10.0, 50.0])
def test_poly_conditioning_rule(a):
    """Evidence for the rule in RCNNCell's docstring (when reaction='factored' is required): on a stable cubic whose
    monomials cancel, the pre-contracted evaluation loses 0.6 * eps * A of relative state accuracy (A =
    poly_amplification) while the factored evaluation stays at the float32 noise floor.  Runs on the plain-C oracle,
    whose per-point arithmetic the HIP kernels reproduce bit for bit (tests/test_hip_parity.py)."""
    from oracle import pi_oracle as O
    from percnn_amd import functional as Fp
    from util import rel_l2
    P32 = _cubic_well_block(a, 1.0, 0.1).astype(np.float32)
    P64 = P32.astype(np.float64)
    Q32 = Fp.contract_block(torch.tensor(P32)).numpy()
    Q64 = Fp.contract_block(torch.tensor(P64)).numpy()
    h0 = (a + np.random.RandomState(0).uniform(-1, 1, (2, 48, 48))).astype(np.float32)
    T = 100
    t64 = O.rollout_fwd(h0.astype(np.float64), P64, 8, T)
    tf = O.rollout_fwd(h0, P32, 8, T)
    tp = O.poly_rollout_fwd(h0, Q32, T)
    hm = float(np.abs(t64).max())
    phi = np.array([1, hm, hm, hm * hm, hm * hm, hm * hm, hm ** 3, hm ** 3, hm ** 3, hm ** 3])
    A = 0.1 * max(float((np.abs(Q64[16 + 10 * s:26 + 10 * s]) * phi).sum()) for s in range(2)) / hm
    eps = 2.0 ** -24
    e_fact, e_poly = rel_l2(tf[-1], t64[-1]), rel_l2(tp[-1], t64[-1])
    assert e_fact < 5e-7, (a, e_fact)                          # factored: noise floor whatever the conditioning
    assert e_poly < max(5e-7, 2.5 * eps * A), (a, A, e_poly)   # poly: predicted by the amplification
    if A >= 100:
        assert e_poly > 0.1 * eps * A and e_poly > 5 * e_fact  # ... and the prediction is tight: the rule is needed
    if A <= 10:
        assert e_poly < 1e-6


def test_shipped_checkpoints_are_inside_the_poly_rule():
    """The default reaction='poly' is justified for the weights the reference ships: A ~ 1 or below for states in [0, 1]."""
    import percnn_amd as pa
    for f, mk, bound in (("gs2d_ckpt_64x64.npz", pa.gs2d_cell, 2.0), ("gs3d_ckpt_16x16x16.npz", pa.gs3d_cell, 1.0),
                         ("lo2d_ckpt_64x64.npz", pa.lo2d_cell, 0.2)):
        g = Golden(os.path.join(GOLDEN, f))
        cell = mk()
        cell.load_state_dict({k: torch.tensor(v) for k, v in g.sd.items()})
        A = cell.poly_amplification(1.0, 1.0)
        assert 0 < A < bound, (f, A)
        keep = cell.reaction
        assert cell.reaction == keep == "poly"


def test_operators_are_registered_with_schemas_and_fake_impls():
    """torch.ops.percnn.*: schemas as documented, FakeTensor propagation without a device, loud failure on CPU tensors."""
    import percnn_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    ns = torch.ops.percnn
    assert str(ns.pi_step.default._schema) == 'percnn::pi_step(Tensor h, Tensor params, str options="") -> Tensor'
    assert "SymInt steps" in str(ns.pi_rollout.default._schema)
    for name in ("pi_step_backward", "pi_rollout_backward", "pi_rollout_observe", "pi_rollout_observe_backward"):
        assert hasattr(ns, name)
    with FakeTensorMode():
        h = torch.empty(1, 2, 16, 24, device="cuda")
        P = torch.empty(36, device="cuda")
        assert ns.pi_step(h, P).shape == h.shape
        assert ns.pi_rollout(h, P, 7).shape == (8, 2, 16, 24)
        pred, traj = ns.pi_rollout_observe(h, P, 20, [0, 5, 10, 15], [4, 4])
        assert pred.shape == (4, 2, 4, 6) and traj.shape == (21, 2, 16, 24)
        g0, gp = ns.pi_rollout_backward(traj, P, traj)
        assert g0.shape == (1, 2, 16, 24) and gp.shape == P.shape
    with pytest.raises(RuntimeError, match="no CPU path"):
        ns.pi_step(torch.zeros(1, 2, 8, 8), torch.zeros(36))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ns.pi_rollout(torch.zeros(1, 2, 8, 8), torch.zeros(36), 3)


def test_exchanger_is_cached_per_group_and_device():
    """slab_rollout(ex=None) must not create a communicator per call (one exchanger per process / group / transport)."""
    from percnn_amd import slab
    a, b = slab.make_exchanger(), slab.make_exchanger()
    assert a is b
    c = slab.make_exchanger(force_p2p=True)
    assert c is not a and c is slab.make_exchanger(force_p2p=True)
    slab.close_exchangers()
    assert slab.make_exchanger() is not a


def test_slab_transport_selection(monkeypatch):
    """transport names are validated; without a HIP device the peer-mailbox / RCCL transports are never attempted and the
    portable exchanger is what every name resolves to; the environment variable is the default of the argument."""
    import ctypes
    from percnn_amd import slab, _lib
    slab.close_exchangers()
    with pytest.raises(ValueError, match="unknown slab transport"):
        slab.make_exchanger(transport="carrier-pigeon")
    if not torch.cuda.is_available():
        for name in ("auto", "peer", "rccl", "dist"):
            assert type(slab.make_exchanger(transport=name)) is slab.HaloExchanger
        monkeypatch.setenv("PERCNN_SLAB_TRANSPORT", "smoke-signals")
        with pytest.raises(ValueError):
            slab.make_exchanger()
        monkeypatch.setenv("PERCNN_SLAB_TRANSPORT", "peer")
        assert type(slab.make_exchanger()) is slab.HaloExchanger
    slab.close_exchangers()
    # the ctypes mirrors of the two ring structs have the C layout (the library checks percnn_pi_halo_ring itself)
    assert ctypes.sizeof(_lib.PeerRing) == 3 * ctypes.sizeof(ctypes.c_void_p) + ctypes.sizeof(ctypes.c_size_t) + 16
    L = _lib.lib()
    assert L.percnn_pi_peer_box_bytes(1) == 4096 + 4 * 4096 and L.percnn_pi_peer_box_bytes(4097) == 4096 + 4 * 8192
    # argument errors surface as return codes before any HIP call
    assert L.percnn_pi_peer_box_alloc(None, 4096) == -1 and L.percnn_pi_peer_box_export(None, None) == -1
    assert L.percnn_pi_peer_box_open(None, None) == -1
    assert L.percnn_pi_peer_exchange_f32(None, 3, _lib.shape_arg((8, 8, 8)), 2, 2, None, None) == -1


def test_direct_kernel_block_decomposition_rules():
    """Host logic of the direct kernels' lane decomposition (set_blockmap) and of the adjoint's planes per pass (direct_rz),
    through the diagnostic entry point -- no device work: every chunk of every row is covered, the fitted rule never
    leaves more lanes idle than the earlier one (beyond its segment-length weights), power-of-two widths keep the
    decomposition the BASELINE sizes were measured with, the flat decomposition appears exactly where stated."""
    import ctypes
    from percnn_amd import _lib
    L = _lib.lib()

    def bm(shape, elem=4, options=None):
        out = (ctypes.c_int * 6)()
        rc = L.percnn_pi_debug_blockmap(len(shape), _lib.shape_arg(shape), elem, options.encode() if options else None, out)
        assert rc == 0, (shape, rc)
        return dict(lxs=out[0], nxb=out[1], nrg=out[2], nblk=out[3], rz=out[4], block=out[5])

    rs = np.random.RandomState(0)
    shapes = [(48,) * 3, (96,) * 3, (100,) * 3, (128,) * 3, (144,) * 3, (160,) * 3, (192,) * 3, (200,) * 3, (208,) * 3,
              (224,) * 3, (256,) * 3, (384,) * 3, (32, 256, 256), (1800, 1800), (2048, 2048), (100, 100), (7, 12, 40)]
    shapes += [tuple(int(v) for v in rs.randint(2, 200, 3)) for _ in range(40)] + [tuple(int(v) for v in rs.randint(2, 3000, 2)) for _ in range(20)]
    for shape in shapes:
        for elem in (4, 8):
            vecw = 16 // elem
            vec = vecw if shape[-1] % vecw == 0 else 1
            cpr = shape[-1] // vec
            rows = shape[1] if len(shape) == 3 else shape[0]
            planes = shape[0] if len(shape) == 3 else 1
            new, old = bm(shape, elem), bm(shape, elem, "lane_x=-1")
            for d in (new, old):
                groups = -(-planes // d["rz"]) if len(shape) == 3 else 1
                if d["lxs"] < 0:                                        # flat: consecutive chunks of the plane
                    assert d["nxb"] == 1 and d["nrg"] * d["block"] >= rows * cpr > (d["nrg"] - 1) * d["block"]
                else:
                    lx, rb = 1 << d["lxs"], d["block"] >> d["lxs"]
                    assert 2 <= d["lxs"] <= 6 and lx <= d["block"]
                    assert d["nxb"] * lx >= cpr > (d["nxb"] - 1) * lx and d["nrg"] * rb >= rows > (d["nrg"] - 1) * rb
                assert d["nblk"] == d["nxb"] * d["nrg"] * groups
            assert old["lxs"] >= 2                                       # the earlier rule never goes flat
            # useful lanes: the fitted rule trades at most its segment weights (>= 0.70 / 1.0) against the earlier one
            eff = lambda d: (planes * rows * cpr) / (d["nblk"] * d["block"] * d["rz"])
            assert eff(new) >= 0.69 * eff(old) - 1e-9, (shape, elem, new, old)
            npts = int(np.prod(shape))
            if new["lxs"] < 0:
                assert npts < (8 << 20) and new["rz"] <= 2 and eff(new) > eff(bm(shape, elem, "lane_x=6")) + 0.05
            assert new["rz"] in (1, 2) and (new["rz"] == 1 or npts >= (3 << 17))
    for shape in ((128,) * 3, (256,) * 3, (32, 256, 256), (2048, 2048), (512, 512)):       # the sizes the profiles were taken at
        assert bm(shape)["lxs"] == bm(shape, 4, "lane_x=-1")["lxs"]
    assert bm((128,) * 3)["rz"] == 1 and bm((32, 256, 256))["rz"] == 1                      # exactly one resident round
    assert bm((144,) * 3)["rz"] == 2 and bm((96,) * 3)["rz"] == 2 and bm((64,) * 3)["rz"] == 1 and bm((256,) * 3)["rz"] == 2
    assert bm((192,) * 3)["lxs"] == 4 and bm((160,) * 3)["lxs"] == -1 and bm((144,) * 3)["lxs"] == -1 and bm((48,) * 3)["lxs"] == -1
    assert bm((192,) * 3, 4, "lane_x=7")["lxs"] == -1 and bm((192,) * 3, 4, "lane_x=5")["lxs"] == 5


def test_param_block_cache_invalidation():
    """RCNNCell.param_block() is cached for callers that keep the reference's per-step loop: the same block
    object while nothing it depends on changes, a new one after a backward pass through it, optimizer.step(),
    load_state_dict(), `cell.dt = ...`, a switch of grad mode or of the reaction mode."""
    import percnn_amd as pa
    torch.manual_seed(0)
    for reaction in ("poly", "factored"):
        cell = pa.RCNNCell(2, 4, reaction=reaction)
        P1 = cell.param_block()
        assert cell.param_block() is P1 and P1.requires_grad
        P1.sum().backward()                                   # consumed: the graph behind P1 is gone
        P2 = cell.param_block()
        assert P2 is not P1 and torch.equal(P2, P1) and cell.param_block() is P2
        opt = torch.optim.SGD(cell.parameters(), lr=0.1)
        opt.step()                                            # parameters changed in place
        P3 = cell.param_block()
        assert P3 is not P2 and not torch.equal(P3, P2)
        sd = {k: v.clone() for k, v in cell.state_dict().items()}
        sd["CA"] = sd["CA"] + 1.0
        cell.load_state_dict(sd)
        P4 = cell.param_block()
        assert P4 is not P3 and float(P4[1]) != float(P3[1])
        cell.dt = 0.25                                        # the reference reads self.dt every step
        P5 = cell.param_block()
        assert P5 is not P4 and float(P5[0]) == 0.25
        with torch.no_grad():
            P6 = cell.param_block()
            assert P6 is not P5 and not P6.requires_grad and cell.param_block() is P6
        assert cell.param_block() is not P6 and cell.param_block().requires_grad
        cell.reaction = "factored" if reaction == "poly" else "poly"
        assert cell.param_block().numel() != P5.numel()
        # gradients of several uses of ONE cached block accumulate on it: two "steps" == twice the gradient of one
        cell.zero_grad()
        (cell.param_block().sum() + cell.param_block().sum()).backward()
        g2 = cell.Wh1_u.weight.grad.clone()
        cell.zero_grad()
        cell.param_block().sum().backward()
        assert torch.allclose(g2, 2 * cell.Wh1_u.weight.grad)


def test_rollout_plan_follows_the_dispatch_rules():
    """percnn_pi_debug_plan = the library's own answer to "which kernels would this rollout run on" (bench.py labels its
    roofline entries with it): the BASELINE configs and the switch points documented in DESIGN.md."""
    from percnn_amd import _lib
    plan = _lib.rollout_plan
    p = plan(0, (512, 512), 4)
    assert p["fwd"] == "tile2d" and p["bwd"] == "tile2d" and p["fused_gradients"] and p["fwd_steps_per_launch"] == 4
    assert p["tile"] == (32, 32, 512) and plan(0, (100, 100), 4)["tile"] == (32, 8, 256)
    # 16-row tiles up to 112 tiles of 32 x 32; whole-tile grids of 113 .. 128 take 32 x 32 (resident pyramid kernels, round 5)
    assert plan(0, (320, 320), 4)["tile"] == (32, 16, 320) and plan(0, (352, 352), 4)["tile"] == (32, 32, 512)
    assert plan(0, (256, 512), 4)["tile"] == (32, 32, 512) and plan(0, (340, 340), 4)["tile"] == (32, 16, 320)
    # past 512^2 the tiles grow while that keeps the grid in one round of 256 workgroups (float32 poly blocks only)
    assert plan(0, (544, 544), 4)["tile"] == (32, 40, 640) and plan(0, (544, 544), 4)["fused_gradients"]
    assert plan(0, (640, 640), 4)["tile"] == (40, 40, 768) and plan(0, (576, 576), 4)["tile"] == (40, 40, 768)
    # the forward stops at 32 x 40 (its second round of 32 x 32 tiles is co-resident; 40 x 40 measured slower)
    assert plan(0, (544, 544), 4)["tile_fwd"] == (32, 40, 640) and plan(0, (640, 640), 4)["tile_fwd"] == (32, 32, 512)
    assert plan(0, (1536, 1536), 4)["tile_fwd"] == (32, 32, 512) and plan(0, (1536, 1536), 4)["tile"] is None
    assert plan(0, (704, 704), 4)["tile"] == (32, 32, 512) and plan(0, (544, 544), 8)["tile"] == (32, 32, 512)
    assert plan(0, (544, 544), 4, "tile_wide=0")["tile"] == (32, 32, 512) and plan(8, (544, 544), 4)["tile"] == (32, 32, 512)
    p = plan(0, (128, 128, 128), 4)                       # 3D Gray-Scott 128^3: bricks, two planes forward, one adjoint
    assert p == {"fwd": "brick3d", "bwd": "brick3d", "fused_gradients": True, "fwd_steps_per_launch": 1,
                 "bwd_steps_per_launch": 1, "fwd_planes_per_pass": 2, "bwd_planes_per_pass": 1, "brick_lanes": 256, "tile": None, "tile_fwd": None, "bwd_persistent": False, "fwd_persistent": False}
    assert plan(0, (48, 48, 48), 4)["fwd_planes_per_pass"] == 1 and plan(0, (48, 48, 48), 4)["brick_lanes"] == 256
    assert plan(0, (32, 256, 256), 4, "brick_nt=512")["brick_lanes"] == 512 and plan(0, (144, 144, 144), 4)["brick_lanes"] == 256
    p = plan(0, (256, 256, 256), 4)                       # forward keeps the z-march from 8 M points on, the adjoint takes bricks
    assert p["fwd"] == "stream3d" and p["bwd"] == "brick3d" and p["bwd_planes_per_pass"] == 2
    assert plan(0, (64, 256, 256), 4)["fwd"] == "brick3d"
    p = plan(0, (384, 384, 384), 4)                       # rows of 96 chunks: 512-lane bricks since round 4 (were: direct kernels)
    assert p["fwd"] == "brick3d" and p["bwd"] == "direct" and p["brick_lanes"] == 512      # (adjoint: below 25 M points only)
    q = plan(0, (64, 384, 384), 4)
    assert q["fwd"] == "direct" and q["bwd"] == "brick3d" and q["brick_lanes"] == 512      # (forward: from 16 M points on)
    assert plan(0, (384, 384, 384), 4, "brick_wide=0")["fwd"] == "direct" and plan(0, (64, 64, 640), 4)["fwd"] == "direct"
    assert plan(2, (384, 384, 384), 4)["fwd"] == "direct"                 # (the wide flavours exist for pre-contracted blocks)
    assert plan(0, (128, 128, 128), 4, "brick3d=0")["fwd"] == "direct"
    assert plan(2, (48, 48, 48), 4)["bwd"] == "brick3d" and not plan(2, (48, 48, 48), 4)["fused_gradients"]
    assert plan(0, (30, 30, 30), 4)["fwd"] == "direct"                    # W % 4 != 0: 4-byte lanes
    p = plan(0, (512, 512), 8)
    assert p["fwd"] == "tile2d" and p["fused_gradients"]
    assert plan(0, (2048, 2048), 4)["bwd"] == "direct" and plan(-1, (100, 100), 8)["fwd"] == "advective"


def test_3d_upscaler_contraction_path_equals_stock_layers():
    """The 3D IC generator evaluates its transposed convolutions as matmuls (MIOpen's ConvTranspose3d is 15x slower on
    MI355X); values and all gradients must equal the stock torch.nn layers it holds (train_3drd.py:41-56)."""
    import percnn_amd as pa
    torch.manual_seed(1)
    up = pa.Upscaler(3).double()
    x = torch.rand(1, 2, 6, 5, 7, dtype=torch.float64, requires_grad=True)
    w = torch.randn(1, 2, 12, 10, 14, dtype=torch.float64)
    ref = up.convnet(x)
    gref = torch.autograd.grad((ref * w).sum(), [x] + list(up.parameters()))
    out = up(x)
    assert out.shape == ref.shape and torch.allclose(out, ref, rtol=1e-12, atol=1e-12)
    gout = torch.autograd.grad((out * w).sum(), [x] + list(up.parameters()))
    for a, b in zip(gout, gref):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)
    # float32, cubic, several z-slabs
    up32 = pa.Upscaler(3)
    x32 = torch.rand(1, 2, 8, 8, 8)
    assert torch.allclose(up32(x32), up32.convnet(x32), rtol=1e-4, atol=1e-5)



def test_bench_promotes_the_sharded_result_for_n_gt_1():
    """bench.py --gpus N (N > 1): the line's top level is configs[4] strong-scaled (value, scaling, transport, rank count, N = 1
    anchor, bit-identity flag); the 2D replicas become an extra; without a valid sharded measurement the replica line stays and
    says why (VERDICT r4 next #1d)."""
    import bench
    base = {"metric": "pi_block_rollout_fwd_bwd_steps_per_sec", "value": 2.0e6, "unit": "steps/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 4.0, "higher_is_better": True, "scaling": "weak", "dtype": "f32", "config": {"workload": "gs2d_512", "reaction": "poly"},
            "roofline": {"frac": 0.5}, "timed_region_s": 0.08, "fwd_us_per_time_step": 1.3, "bwd_us_per_time_step": 2.0,
            "fwd_only_steps_per_sec": 1.0, "also": {"gs3d_128": {"value": 1.0}}}
    head = {"steps_per_sec_fwd_bwd": 16000.0, "us_per_time_step_fwd_bwd": 62.5, "ms_per_step": 6.25, "steps": 20, "warmup": 5, "T": 100,
            "timed_region_s": 0.125, "transport": "RcclHaloExchanger", "exchange": "ncclSend / ncclRecv", "schedule": "plain",
            "ranks_seen_by_transport": 8, "forward_state_equals_single_domain_rollout": True, "frames_compared": 11,
            "global_points": 256 ** 3, "points_per_rank": 32 * 256 * 256, "grid": [256, 256, 256]}
    out = dict(base, slab_3d={"headline": head, "headline_n1_anchor": {"steps_per_sec_fwd_bwd": 4500.0, "us_per_time_step_fwd_bwd": 222.0, "T": 20}})
    bench.promote_sharded_headline(out, 8)
    assert out["value"] == 16000.0 and out["scaling"] == "strong" and out["ms_per_step"] == 6.25 and out["n_gpus"] == 8
    assert "configs[4]" in out["config"]["workload"] and out["config"]["grid"] == [256, 256, 256] and out["config"]["T"] == 100
    assert out["transport"] == "RcclHaloExchanger" and out["ranks_seen_by_transport"] == 8
    assert out["forward_state_equals_single_domain_rollout"] is True
    assert abs(out["speedup_vs_n1_anchor"] - 16000.0 / 4500.0) < 1e-12 and out["n1_anchor"]["steps_per_sec"] == 4500.0
    assert abs(out["roofline"]["achieved"] - 48.0 * 256 ** 3 * 16000.0 / 1e9) < 1e-6 and out["roofline"]["peak"] == 64000.0
    assert out["replicas_2d"]["value"] == 2.0e6 and out["replicas_2d"]["scaling"] == "weak" and "also" not in out
    for bad in ({"error": "boom"}, {"headline": {"error": "RuntimeError('x')"}}, {}):
        keep = dict(base, slab_3d=bad)
        bench.promote_sharded_headline(keep, 8)
        assert keep["value"] == 2.0e6 and keep["scaling"] == "weak" and keep["sharded_headline_missing"]


def test_frames_make_the_callers_cat_a_view():
    """functional.Frame (round 5): ``torch.cat(tuple(outputs), dim=0)`` -- the reference's own next line after ``model()``
    (train_2drd.py:394) -- returns the trajectory buffer for the frames of one rollout in order, a slice for a run of consecutive
    steps, and is the stock copying cat for everything else (other order, other dim, foreign tensors, two rollouts, ``out=``,
    every n-th step); operations on a frame return plain tensors."""
    from percnn_amd import functional as F_pi

    def frames_of(st):
        fr = []
        for k, v in enumerate(st.unsqueeze(1).unbind(0)):
            f = v.as_subclass(F_pi.Frame)
            f._pi_index = k
            fr.append(f)
        F_pi.link_frames(fr, st)
        return fr

    st = torch.arange(6 * 2 * 3 * 4, dtype=torch.float32).reshape(6, 2, 3, 4)
    fr = frames_of(st)
    assert all(isinstance(f, torch.Tensor) and f.shape == (1, 2, 3, 4) for f in fr)
    for seq in (tuple(fr), list(fr)):
        c = torch.cat(seq, dim=0)
        assert c is st and type(c) is torch.Tensor
    assert torch.cat(tuple(fr), 0) is st and torch.cat(tuple(fr), dim=-4) is st
    run = torch.cat(tuple(fr[2:5]), dim=0)
    assert run.data_ptr() == st[2].data_ptr() and run.shape == (3, 2, 3, 4) and torch.equal(run, st[2:5])
    # everything else: a copy with the stock semantics
    for seq, kw in (((fr[2], fr[0]), {"dim": 0}), (tuple(fr[::2]), {"dim": 0}), (tuple(fr), {"dim": 1}),
                    ((fr[0], st[1:2]), {"dim": 0}), ((st[0:1], fr[1]), {"dim": 0})):
        c = torch.cat(seq, **kw)
        ref = torch.cat(tuple(t.as_subclass(torch.Tensor) for t in seq), **kw)
        assert type(c) is torch.Tensor and torch.equal(c, ref) and c.data_ptr() != st.data_ptr()
    other = frames_of(st.clone())
    c = torch.cat((fr[0], other[1]), dim=0)
    assert c.data_ptr() not in (st.data_ptr(), other[0].data_ptr()) and torch.equal(c, st[0:2])
    buf = torch.empty(6, 2, 3, 4)
    assert torch.cat(tuple(fr), dim=0, out=buf) is buf and torch.equal(buf, st)
    # a frame behaves as a tensor and does not spread its type
    assert type(fr[1] * 2) is torch.Tensor and type(fr[1].clone()) is torch.Tensor and type(fr[1][0]) is torch.Tensor
    assert fr[3].detach().numpy().shape == (1, 2, 3, 4) and float(fr[1].sum()) == float(st[1].sum())
    assert type(torch.stack(tuple(fr))) is torch.Tensor and torch.stack(tuple(fr)).shape == (6, 1, 2, 3, 4)
    # copies and pickles are plain tensors (a copy is no view of the buffer; the link must not drag the trajectory along)
    import copy, io, pickle
    c = copy.deepcopy(fr[1])
    assert type(c) is torch.Tensor and c.data_ptr() != fr[1].data_ptr() and torch.equal(c, st[1:2])
    assert all(type(t) is torch.Tensor for t in copy.deepcopy(fr)) and type(copy.copy(fr[1])) is torch.Tensor
    buf_io = io.BytesIO()
    torch.save(fr[2], buf_io)
    buf_io.seek(0)
    back = torch.load(buf_io)
    assert type(back) is torch.Tensor and torch.equal(back, st[2:3]) and buf_io.getbuffer().nbytes < 4096
    assert type(pickle.loads(pickle.dumps(fr[3]))) is torch.Tensor
