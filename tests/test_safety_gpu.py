"""GPU tests of the defaults' safety nets:

* the persistent tile sweep checks residency instead of assuming it: with CUs held by another kernel it aborts without writing
  anything and the same call runs the launch-per-group sweep (handshake, default), or the next entry point reports it
  (``persist_handshake=0``) -- never NaNs, never a silent wrong gradient;
* reaction='poly' (the default, a rewrite of the reference's Wh4(Wh1*Wh2*Wh3), train_2drd.py:115-116) prices its own
  conditioning on the device and a cell whose weights leave the rule evaluates the factored form by itself;
* the parameter-block cache sees every kind of update except `.data` edits, for which invalidate_cache() exists and which
  RCNN-level rollouts never depend on.
"""
import warnings

import numpy as np
import pytest
import torch

from util import random_block, rel_l2

pytestmark = pytest.mark.gpu


def dev_t(a, device):
    return torch.tensor(np.ascontiguousarray(a), device=device)


def _hog(blocks, lds_bytes, ms, device):
    """Hold `blocks` CUs' LDS for `ms` milliseconds NEXT TO the current stream; returns the side stream the hog runs on.
    HIP multiplexes streams onto a few hardware queues: a freshly created stream may share its queue with the current stream, and
    then the hog and the kernel under test simply run one after the other (seen in the full suite, never in a short script: the
    queue a new stream gets depends on how many streams the process has created).  The entry point returns once every hog
    workgroup is resident; a probe on the current stream then tells whether that stream still makes progress."""
    import ctypes
    import time
    from percnn_amd import _lib
    for _ in range(8):
        side = torch.cuda.Stream(device=device)
        _lib.check(_lib.lib().percnn_pi_debug_hog(blocks, lds_bytes, ms, ctypes.c_void_p(side.cuda_stream)), "debug_hog")
        probe, ev = torch.zeros(1, device=device), torch.cuda.Event()
        probe.add_(1)
        ev.record()
        t0 = time.perf_counter()
        ev.synchronize()
        if time.perf_counter() - t0 < 0.25 * ms / 1000.0:
            return side
        side.synchronize()                                     # serialised behind the hog: it is over by now, try another stream
    pytest.skip("no side stream that runs concurrently with the current stream")


def _sweep_problem(hip_device, shape=(512, 512), T=41):
    import percnn_amd as pa
    rs = np.random.RandomState(4)
    P = dev_t(random_block(0, 2, np.float32, 21, scale=0.1), hip_device)
    traj = torch.empty((T + 1, 2) + shape, dtype=torch.float32, device=hip_device)
    traj[0] = dev_t(rs.uniform(0, 1, (2,) + shape).astype(np.float32), hip_device)
    pa.rollout_fwd_(traj, P)
    g = torch.randn(traj.shape, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(1)) / traj[0].numel()
    return traj, g, P


def test_persistent_sweep_aborts_cleanly_and_falls_back(hip_device):
    """CUs held by another kernel (a stand-in for a second process / a CU mask): the persistent launch cannot get all its
    workgroups resident, gives up within `persist_first_timeout_ms` WITHOUT writing outputs, and the same rollout_bwd call runs
    the launch-per-group sweep: results bit-identical to tile_persist=0, no NaNs, no exception; the device then stays on the
    launch-per-group path until persist_reset."""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    traj, g, P = _sweep_problem(hip_device)
    ref0, refg = pa.rollout_bwd(traj, g, P, options={"tile_persist": 0})
    s0 = _lib.persist_status()
    a0, ag = pa.rollout_bwd(traj, g, P)                          # healthy: the persistent launch runs
    s1 = _lib.persist_status()
    assert s1["launches"] == s0["launches"] + 1 and s1["aborts"] == s0["aborts"] and not s1["disabled_on_current_device"]
    assert torch.equal(a0, ref0)
    torch.cuda.synchronize()
    try:
        _hog(16, 150 * 1024, 1500, hip_device)                        # 16 CUs' LDS for 1.5 s: 16 of the 256 tiles cannot start
        b0, bg = pa.rollout_bwd(traj, g, P, options={"persist_first_timeout_ms": 20})
        s2 = _lib.persist_status()
        torch.cuda.synchronize()
        assert s2["launches"] == s1["launches"] + 1 and s2["aborts"] == s1["aborts"] + 1
        assert s2["disabled_on_current_device"] and s2["last_abort_group"] == 0
        assert torch.isfinite(b0).all() and torch.isfinite(bg).all()
        assert torch.equal(b0, ref0)
        assert rel_l2(bg.cpu().numpy(), refg.cpu().numpy()) < 2e-6
        # the device now keeps the launch-per-group path: no new persistent launch, same results
        c0, cg = pa.rollout_bwd(traj, g, P)
        assert _lib.persist_status()["launches"] == s2["launches"] and torch.equal(c0, ref0)
        assert not _lib.rollout_plan(0, (512, 512), 4)["bwd_persistent"]
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    assert _lib.rollout_plan(0, (512, 512), 4)["bwd_persistent"]
    d0, dg = pa.rollout_bwd(traj, g, P)
    assert _lib.persist_status()["launches"] == s2["launches"] + 1 and torch.equal(d0, ref0)


def test_resident_3d_sweep_aborts_cleanly_and_falls_back(hip_device):
    """Round 6: the resident 3D sweep (256 workgroups at 128^3, 159 KB of LDS each) under the same stress: with 16 CUs held it
    gives up at its first hand-over without writing dL/dh0 or a partial row, the same rollout_bwd call runs the brick sweep --
    bit-identical to res3d=0 -- and the device keeps the launch-per-step path until persist_reset."""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    shape, T = (128, 128, 128), 18
    rs = np.random.RandomState(6)
    P = dev_t(random_block(0, 3, np.float32, 21, scale=0.1), hip_device)
    traj = torch.empty((T + 1, 2) + shape, dtype=torch.float32, device=hip_device)
    traj[0] = dev_t(rs.uniform(0, 1, (2,) + shape).astype(np.float32), hip_device)
    pa.rollout_fwd_(traj, P)
    g = torch.randn(traj.shape, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(1)) / traj[0].numel()
    ref0, refg = pa.rollout_bwd(traj, g, P, options={"res3d": 0})
    s0 = _lib.persist_status()
    a0, ag = pa.rollout_bwd(traj, g, P)
    s1 = _lib.persist_status()
    assert s1["launches"] == s0["launches"] + 1 and s1["aborts"] == s0["aborts"] and torch.equal(a0, ref0)
    torch.cuda.synchronize()
    try:
        for attempt in range(3):
            side = _hog(16, 150 * 1024, 1500, hip_device)
            b0, bg = pa.rollout_bwd(traj, g, P, options={"persist_first_timeout_ms": 20})
            s2 = _lib.persist_status()
            torch.cuda.synchronize()
            assert s2["launches"] == s1["launches"] + 1
            if s2["aborts"] == s1["aborts"] + 1:
                break
            # (seen once in the full suite, never alone: this launch found its 256 CUs although the probe saw the hog running; the
            # precondition of the test was not met -- the result must be right all the same -- try again)
            assert s2["aborts"] == s1["aborts"] and torch.equal(b0, ref0)
            side.synchronize()
            s1 = s2
        else:
            pytest.skip("the hog never kept the resident 3D sweep from its CUs")
        assert s2["disabled_on_current_device"]
        assert torch.isfinite(b0).all() and torch.isfinite(bg).all() and torch.equal(b0, ref0)
        assert rel_l2(bg.cpu().numpy(), refg.cpu().numpy()) < 2e-6
        c0, cg = pa.rollout_bwd(traj, g, P)
        assert _lib.persist_status()["launches"] == s2["launches"] and torch.equal(c0, ref0)
        assert not _lib.rollout_plan(0, shape, 4)["bwd_persistent"]
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    assert _lib.rollout_plan(0, shape, 4)["bwd_persistent"]
    d0, dg = pa.rollout_bwd(traj, g, P)
    assert _lib.persist_status()["launches"] == s2["launches"] + 1 and torch.equal(d0, ref0)


def test_persistent_forward_aborts_cleanly_and_falls_back(hip_device):
    """The resident forward under the same stress (CUs held by another kernel): the launch gives up at its first hand-over, the
    same rollout_fwd_ call recomputes the trajectory launch by launch -- bit-identical to fwd_persist=0, no NaNs -- and the
    device stays on the launch-per-group path until persist_reset."""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    traj, _, P = _sweep_problem(hip_device)
    ref = torch.empty_like(traj)
    ref[0] = traj[0]
    pa.rollout_fwd_(ref, P, options={"fwd_persist": 0})
    s0 = _lib.persist_status()
    a = torch.full_like(traj, float("nan"))
    a[0] = traj[0]
    pa.rollout_fwd_(a, P)                                         # healthy: one resident launch
    s1 = _lib.persist_status()
    assert s1["launches"] == s0["launches"] + 1 and s1["aborts"] == s0["aborts"] and torch.equal(a, ref)
    torch.cuda.synchronize()
    try:
        _hog(144, 150 * 1024, 1500, hip_device)                  # (two of its 77 KB workgroups fit a CU: 112 free CUs hold 224 < 256)
        b = torch.full_like(traj, float("nan"))
        b[0] = traj[0]
        pa.rollout_fwd_(b, P, options={"persist_first_timeout_ms": 20})
        s2 = _lib.persist_status()
        torch.cuda.synchronize()
        assert s2["launches"] == s1["launches"] + 1 and s2["aborts"] == s1["aborts"] + 1 and s2["disabled_on_current_device"]
        assert torch.isfinite(b).all() and torch.equal(b, ref)
        assert not _lib.rollout_plan(0, (512, 512), 4)["fwd_persistent"]
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    assert _lib.rollout_plan(0, (512, 512), 4)["fwd_persistent"]


def test_persistent_forward_float64_aborts_cleanly_and_falls_back(hip_device):
    """the float64 resident forward (round 5) with CUs held by another kernel: gives up at its first hand-over, the same call
    recomputes the trajectory launch by launch -- bit-identical, no NaNs"""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    shape, T = (512, 512), 36
    P = torch.tensor(random_block(0, 2, np.float64, 37, scale=0.1), device=hip_device)
    h0 = torch.rand((2,) + shape, dtype=torch.float64, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(5))
    ref = torch.empty((T + 1, 2) + shape, dtype=torch.float64, device=hip_device)
    ref[0] = h0
    pa.rollout_fwd_(ref, P, options={"fwd_persist": 0})
    s0 = _lib.persist_status()
    a = torch.full_like(ref, float("nan"))
    a[0] = h0
    pa.rollout_fwd_(a, P)
    s1 = _lib.persist_status()
    assert s1["launches"] == s0["launches"] + 1 and s1["aborts"] == s0["aborts"] and torch.equal(a, ref)
    torch.cuda.synchronize()
    try:
        _hog(144, 150 * 1024, 1500, hip_device)                  # one 113 KB workgroup per CU: 112 free CUs cannot hold 256
        b = torch.full_like(ref, float("nan"))
        b[0] = h0
        pa.rollout_fwd_(b, P, options={"persist_first_timeout_ms": 20})
        s2 = _lib.persist_status()
        torch.cuda.synchronize()
        assert s2["launches"] == s1["launches"] + 1 and s2["aborts"] == s1["aborts"] + 1 and s2["disabled_on_current_device"]
        assert torch.isfinite(b).all() and torch.equal(b, ref)
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    assert _lib.rollout_plan(0, shape, 8)["fwd_persistent"]


def test_persistent_sweep_float64_aborts_cleanly_and_falls_back(hip_device):
    """the float64 resident sweep (round 5) with CUs held by another kernel: gives up at its first hand-over WITHOUT writing
    outputs, the same rollout_bwd call runs the launch-per-group sweep -- bit-identical dL/dh0, no NaNs"""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    shape, T = (512, 512), 41
    P = torch.tensor(random_block(0, 2, np.float64, 21, scale=0.1), device=hip_device)
    traj = torch.empty((T + 1, 2) + shape, dtype=torch.float64, device=hip_device)
    traj[0] = torch.rand((2,) + shape, dtype=torch.float64, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(6))
    pa.rollout_fwd_(traj, P)
    g = torch.randn(traj.shape, dtype=torch.float64, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(1)) / traj[0].numel()
    ref0, refg = pa.rollout_bwd(traj, g, P, options={"tile_persist": 0})
    s0 = _lib.persist_status()
    a0, ag = pa.rollout_bwd(traj, g, P)
    s1 = _lib.persist_status()
    assert s1["launches"] == s0["launches"] + 1 and s1["aborts"] == s0["aborts"] and torch.equal(a0, ref0)
    torch.cuda.synchronize()
    try:
        _hog(144, 150 * 1024, 1500, hip_device)                  # one 154 KB workgroup per CU: 112 free CUs cannot hold 256
        b0, bg = pa.rollout_bwd(traj, g, P, options={"persist_first_timeout_ms": 20})
        s2 = _lib.persist_status()
        torch.cuda.synchronize()
        assert s2["launches"] == s1["launches"] + 1 and s2["aborts"] == s1["aborts"] + 1 and s2["disabled_on_current_device"]
        assert torch.isfinite(b0).all() and torch.equal(b0, ref0) and torch.isfinite(bg).all()
        assert float((bg - refg).norm() / refg.norm()) < 1e-12
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    assert _lib.rollout_plan(0, shape, 8)["bwd_persistent"]


def test_small_tile_persistent_forward_aborts_cleanly_and_falls_back(hip_device):
    """The small-tile resident forward (round 5) under the same stress: CUs held by another kernel -> the launch gives up at its
    first hand-over, the same call recomputes the trajectory launch by launch (bit-identical, no NaNs), the device stays on the
    launch-per-group path until persist_reset."""
    import percnn_amd as pa
    from percnn_amd import _lib
    from util import random_block
    pa.set_option("persist_reset", 1)
    shape, T = (256, 256), 40                                      # 32 x 8 tiles: 256 workgroups, one per CU
    P = torch.tensor(random_block(0, 2, np.float32, 29, scale=0.1), device=hip_device)
    h0 = torch.rand((2,) + shape, device=hip_device, generator=torch.Generator(device=hip_device).manual_seed(4))
    ref = torch.empty((T + 1, 2) + shape, device=hip_device)
    ref[0] = h0
    pa.rollout_fwd_(ref, P, options={"fwd_persist": 0})
    s0 = _lib.persist_status()
    a = torch.full_like(ref, float("nan"))
    a[0] = h0
    pa.rollout_fwd_(a, P)
    s1 = _lib.persist_status()
    assert s1["launches"] == s0["launches"] + 1 and s1["aborts"] == s0["aborts"] and torch.equal(a, ref)
    torch.cuda.synchronize()
    try:
        _hog(200, 150 * 1024, 1500, hip_device)                  # 56 free CUs cannot hold 256 workgroups of 33 KB at 4 per CU ... 5 do
        b = torch.full_like(ref, float("nan"))
        b[0] = h0
        pa.rollout_fwd_(b, P, options={"persist_first_timeout_ms": 20})
        s2 = _lib.persist_status()
        torch.cuda.synchronize()
        assert torch.isfinite(b).all() and torch.equal(b, ref)
        # (whether the launch aborted depends on how many of its small workgroups the free CUs hold; either way: right answer)
        if s2["aborts"] == s1["aborts"] + 1:
            assert s2["disabled_on_current_device"] and not _lib.rollout_plan(0, shape, 4)["fwd_persistent"]
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    assert _lib.rollout_plan(0, shape, 4)["fwd_persistent"]


def test_persistent_sweep_abort_without_handshake_is_reported(hip_device):
    """persist_handshake=0 (fire and forget): an aborted launch leaves its outputs unwritten and the NEXT entry point raises
    (PERCNN_PI_EASYNC), once; after persist_reset everything is back."""
    import percnn_amd as pa
    from percnn_amd import _lib
    pa.set_option("persist_reset", 1)
    # (T = 41: ten groups.  With only a few groups the workgroups far from the missing ones run to the end, free their CUs, the
    # late ones start there and the launch completes by itself -- slower, but correct; it takes more groups than the torus of
    # tiles is wide for a missing workgroup to stall everybody)
    traj, g, P = _sweep_problem(hip_device, T=41)
    ref0, _ = pa.rollout_bwd(traj, g, P, options={"tile_persist": 0})
    torch.cuda.synchronize()
    try:
        _hog(16, 150 * 1024, 1000, hip_device)
        _lib.persist_fence()                                     # nothing pending: a plain stream wait
        pa.rollout_bwd(traj, g, P, options={"persist_first_timeout_ms": 20, "persist_handshake": 0})
        with pytest.raises(RuntimeError, match="EARLIER call's persistent launch"):
            _lib.persist_fence()                                 # (round 5) waits for the stream, reports the abort -- once
        _lib.persist_fence()
        e0, _ = pa.rollout_bwd(traj, g, P)                       # reported once; the device is on the launch-per-group path
        assert torch.equal(e0, ref0) and _lib.persist_status()["disabled_on_current_device"]
    finally:
        torch.cuda.synchronize()
        pa.set_option("persist_reset", 1)
    f0, _ = pa.rollout_bwd(traj, g, P)
    assert torch.equal(f0, ref0)


def _cell_from_block(P, device, reaction="poly"):
    """RCNNCell (2D, raw diffusion coefficients) whose parameters are the entries of the factored block P"""
    import percnn_amd as pa
    hc = ((len(P) - 16) // 2 - 1) // 10
    cell = pa.RCNNCell(2, hc, dx=1.0, dt=float(P[0]), mu_up=None, diffusion="raw", dtype=torch.float32, reaction=reaction)
    with torch.no_grad():
        cell.DA.fill_(float(P[1])); cell.DB.fill_(float(P[2]))
        w = torch.zeros(1, 1, 5, 5)
        w[0, 0, 2, 2] = float(P[3])
        for i, off in enumerate((-2, -1, 1, 2)):
            w[0, 0, 2 + off, 2] = float(P[4 + i])
            w[0, 0, 2, 2 + off] = float(P[8 + i])
        cell.W_laplace.weight.copy_(w)
        for s, name in enumerate(("u", "v")):
            B = P[16 + s * (10 * hc + 1):16 + (s + 1) * (10 * hc + 1)]
            for k in range(3):
                m = getattr(cell, f"Wh{k + 1}_{name}")
                for j in range(hc):
                    m.weight[j, 0, 0, 0], m.weight[j, 1, 0, 0], m.bias[j] = (float(B[10 * j + 3 * k + i]) for i in range(3))
            m4 = getattr(cell, f"Wh4_{name}")
            for j in range(hc):
                m4.weight[0, j, 0, 0] = float(B[10 * j + 9])
            m4.bias[0] = float(B[10 * hc])
    return cell.to(device)


def test_poly_guard_switches_an_ill_conditioned_cell_to_factored(hip_device):
    """VERDICT r3 #2a: the default reaction='poly' is guarded.  A cell whose expanded cubic is ill-conditioned (the cubic well
    of test_poly_conditioning_rule, a = 50: A ~ 4000 at |u|, |v| <= 51) packs the FACTORED block from its very first call,
    warns once, and its rollout is the reaction='factored' cell's bit for bit; a well-conditioned cell (a = 0) stays 'poly'."""
    import percnn_amd as pa
    from test_host_logic import _cubic_well_block
    T, a = 60, 50.0
    P32 = _cubic_well_block(a, 1.0, 0.1).astype(np.float32)
    h0 = dev_t((a + np.random.RandomState(0).uniform(-1, 1, (1, 2, 48, 48))).astype(np.float32), hip_device)
    ref = _cell_from_block(P32, hip_device, reaction="factored")
    cell = _cell_from_block(P32, hip_device)                     # reaction='poly', guard on (defaults)
    cell.state_bound = (a + 1.0, a + 1.0)
    assert cell.reaction == "poly" and cell.poly_guard
    with torch.no_grad():
        want = pa.RCNN(ref, step=T, effective_step=list(range(T)), init_state=h0).trajectory()
        with pytest.warns(RuntimeWarning, match="ill-conditioned"):
            got = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h0).trajectory()
        assert cell.effective_reaction == "factored" and cell._guard.A > 100
        assert cell.param_block().numel() == 16 + 2 * 81
        assert torch.equal(got, want)
        with warnings.catch_warnings():                          # ... once
            warnings.simplefilter("error")
            cell.invalidate_cache()
            cell.param_block()
        # unguarded, the same cell runs the expanded cubic and leaves the float32 noise floor
        cell.poly_guard = False
        raw = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h0).trajectory()
        assert cell.param_block().numel() == 36 and not torch.equal(raw, want)
        ok = _cell_from_block(_cubic_well_block(0.0, 1.0, 0.1).astype(np.float32), hip_device)
        ok.state_bound = (1.0, 1.0)
        ok.param_block()
        assert ok.effective_reaction == "poly" and ok.param_block().numel() == 36 and 0 < ok._guard.A < 10


def test_poly_guard_follows_the_weights_with_one_update_of_lag(hip_device):
    """After the first call the guard never synchronises: it reads the amplification the PREVIOUS pack launch left in its
    host-mapped slot.  Weights that grow past the rule are therefore caught one update later, and a cell that comes back
    under half the bound returns to 'poly' (hysteresis)."""
    from test_host_logic import _cubic_well_block
    cell = _cell_from_block(_cubic_well_block(0.0, 1.0, 0.1).astype(np.float32), hip_device)      # A = 0.2 at |u|, |v| <= 1
    big = _cell_from_block(_cubic_well_block(50.0, 1.0, 0.1).astype(np.float32), hip_device)      # A ~ 1e4 even there
    small_sd = {k: v.clone() for k, v in cell.state_dict().items()}
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert cell.param_block().numel() == 36
        cell.load_state_dict(big.state_dict())                  # a drastic update: version counters move
        cell.param_block()                                       # this pack still ran as 'poly' and priced the new weights
        torch.cuda.synchronize()
        assert cell._guard.read()[0] > 100
        cell.invalidate_cache()
        assert cell.param_block().numel() == 16 + 2 * 81 and cell.effective_reaction == "factored"
        cell.load_state_dict(small_sd)
        cell.param_block()
        torch.cuda.synchronize()
        cell.invalidate_cache()
        assert cell.param_block().numel() == 36 and cell.effective_reaction == "poly"


def test_param_block_cache_updates_and_invalidation(hip_device):
    """the cache key covers every parameter's storage and version; `.data` edits need invalidate_cache()
    in per-step loops (documented) but can never reach an RCNN rollout, which packs afresh."""
    import percnn_amd as pa
    cell = pa.gs2d_cell().to(hip_device)
    h = torch.rand(1, 2, 64, 64, device=hip_device)
    with torch.no_grad():
        P0 = cell.param_block()
        assert cell.param_block() is P0
        cell.Wh2_v.bias.data = cell.Wh2_v.bias.data + 0.25       # `p.data = new` on a tensor the old key did not watch
        P1 = cell.param_block()
        assert P1 is not P0 and not torch.equal(P1, P0)
        cell.Wh3_u.weight.mul_(1.5)                              # in-place under no_grad: version counter
        P2 = cell.param_block()
        assert P2 is not P1 and not torch.equal(P2, P1)
        cell.Wh1_u.weight.data.mul_(2.0)                         # `.data` in-place edit: invisible to the key ...
        assert cell.param_block() is P2
        model = pa.RCNN(cell, step=4, effective_step=[0, 1, 2, 3], init_state=h)
        t_fresh = model.trajectory()                             # ... but a rollout packs afresh
        cell.invalidate_cache()
        P3 = cell.param_block()
        assert not torch.equal(P3, P2)
        assert torch.equal(t_fresh, pa.pi_rollout(h, P3, 4))
        cell.init_filter(cell.filter_list, 0.02)                 # the module's own `.data` edits invalidate by themselves
        assert cell.param_block() is not P3


def test_two_backward_passes_through_one_cached_block(hip_device):
    """two forwards that share the cached block, then two separate backward() calls."""
    import percnn_amd as pa
    cell = pa.gs2d_cell().to(hip_device)
    h1 = torch.rand(1, 2, 64, 64, device=hip_device)
    h2 = torch.rand(1, 2, 64, 64, device=hip_device)
    out1, _ = cell(h1)
    out2, _ = cell(h2)
    out1.sum().backward()
    g1 = cell.Wh4_u.weight.grad.clone()
    out2.sum().backward()
    g12 = cell.Wh4_u.weight.grad.clone()
    cell.zero_grad()
    o, _ = cell(h2)
    o.sum().backward()
    assert torch.allclose(g12 - g1, cell.Wh4_u.weight.grad, rtol=1e-5, atol=1e-7)


def test_strided_time_slices_with_many_runs(hip_device):
    """loss_mse / traj_sqerr with a frame set of more than 64 runs (every 3rd of 400 frames)."""
    import percnn_amd as pa
    cell = pa.gs2d_cell().to(hip_device)
    T = 399
    h = torch.rand(1, 2, 32, 32, device=hip_device)
    model = pa.RCNN(cell, step=T, effective_step=list(range(T)), init_state=h)
    loss = model.loss_mse(t_slice=slice(0, None, 3))
    traj = model.last_trajectory
    want = (traj[0::3].double() ** 2).mean()
    assert abs(float(loss.detach()) - float(want)) < 1e-5 * float(want)
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in cell.parameters() if p.grad is not None)


def test_physics_loss_falls_back_when_the_fused_pass_declines(hip_device):
    """a grid inside physics_loss's size guard that the fused pass turns down (PERCNN_PI_ETOOLARGE: an
    unaligned float32 view above 2^22 points loses the 16-byte lanes) takes the residual-tensor expression instead of raising."""
    import percnn_amd as pa
    from percnn_amd import physics
    n = 2049
    Q = dev_t(random_block(0, 2, np.float32, 3, scale=0.05), hip_device)
    out = torch.rand(3, 2, n, n, device=hip_device) * 0.1
    a = physics.physics_loss(out, Q)
    b = physics.physics_loss(out, Q, fused=False)
    assert torch.isfinite(a) and abs(float(a) - float(b)) <= 1e-5 * abs(float(b))
