"""CPU, multi-process (gloo): the N>1 path -- slab decomposition, ring halo exchange (wide halos in
the forward, 2 planes per step in the adjoint sweep), gradient all-reduce -- against the
single-domain plain-C oracle.  The compute kernel is replaced by the oracle's range-restricted step
(injected explicitly; the product package itself has no CPU path), so what is tested here is
exactly the orchestration that runs around the HIP slab kernels on a multi-GPU node."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rendezvous():
    """A fresh path for torch.distributed's file store: no TCP port to lose to another process between choosing and binding it."""
    import tempfile
    return os.path.join(tempfile.mkdtemp(prefix="percnn_rdzv_"), "store")


def _random_block(hc, ndim, dtype, seed):
    rs = np.random.RandomState(seed)
    P = np.zeros(16 + 2 * (10 * hc + 1), dtype=dtype)
    P[0] = 0.1
    P[1:3] = rs.uniform(0.01, 0.05, 2)
    P[3] = -2.0 * ndim * 1.25
    for a in range(ndim):
        P[4 + 4 * a:8 + 4 * a] = (-1 / 12, 4 / 3, 4 / 3, -1 / 12) + rs.uniform(-0.01, 0.01, 4)
    P[16:] = rs.uniform(-0.5, 0.5, len(P) - 16)
    return P


def _oracle_slab_steps(hc):
    """Test-only stand-ins for percnn_amd.functional.step_fwd / step_bwd (slab=True) on CPU tensors."""
    from oracle import pi_oracle as O

    def step_fwd(h, P, out=None, slab=True, halo=2, skip=0, planes=None):
        assert slab
        N = h.shape[1]
        lo, hi = planes if planes is not None else (skip + 2, N - skip - 2)
        hn, on = h.numpy(), out.numpy()
        O.step_fwd_range(hn, on, P.numpy(), hc, lo, hi)
        return out

    def step_bwd(h, g_out, P, g_inject=None, g_in=None, param_grad=None, slab=True, halo=2, ws=None, planes=None):
        assert slab
        n = h.shape[1] - 2 * halo
        lo, hi = planes if planes is not None else (halo, halo + n)
        inj = g_inject.contiguous().numpy() if g_inject is not None else None
        O.step_bwd_range(h.numpy(), g_out.numpy(), inj, g_in.numpy(), param_grad.numpy(), P.numpy(), hc, lo, hi)
        return g_in, param_grad

    return step_fwd, step_bwd


def _worker(rank, world, port, shape, halo, T, hc, dtype_name, q):
    sys.path.insert(0, ROOT)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from percnn_amd import slab
        from oracle import pi_oracle as O
        dtype = np.dtype(dtype_name)
        ndim = len(shape)
        rs = np.random.RandomState(3)
        P = _random_block(hc, ndim, dtype, 5)
        h0 = rs.uniform(0, 1, (2,) + shape).astype(dtype)
        # single-domain reference (every rank recomputes it; tiny)
        traj_ref = O.rollout_fwd(h0, P, hc, T)
        g_ref = rs.uniform(-1, 1, traj_ref.shape).astype(dtype)
        g0_ref, pg_ref = O.rollout_bwd(traj_ref, g_ref, P, hc)

        ex = slab.HaloExchanger()
        assert (ex.rank, ex.world) == (rank, world)
        lo, hi = slab.split_extent(shape[0], world)[rank]
        n = hi - lo
        fwd, bwd = _oracle_slab_steps(hc)
        Pt = torch.tensor(P)
        local0 = slab.scatter_slab(torch.tensor(h0), rank, world, halo)
        ok_fwd, ok_g0, err_pg = True, True, 0.0
        # both schedules: faces first + asynchronous exchange + planes in between (what runs on a multi-GPU node),
        # and the un-split one
        for overlap in (True, False):
            traj = torch.zeros((T + 1,) + tuple(local0.shape), dtype=local0.dtype)
            traj[0] = local0
            slab.slab_rollout_fwd_(traj, Pt, ex, halo, step_fwd=fwd, overlap=overlap)
            got = traj[:, :, halo:halo + n].numpy()
            ok_fwd &= np.array_equal(got, traj_ref[:, :, lo:hi])       # bit-identical to the single domain

            g_local = torch.zeros_like(traj)
            g_local[:, :, halo:halo + n] = torch.tensor(g_ref[:, :, lo:hi])
            g0, pg = slab.slab_rollout_bwd(traj, g_local, Pt, ex, halo, step_bwd=bwd, wgrad=None, overlap=overlap)
            ok_g0 &= np.array_equal(g0[:, halo:halo + n].numpy(), g0_ref[:, lo:hi])
            err_pg = max(err_pg, float(np.linalg.norm(pg.numpy() - pg_ref) / np.linalg.norm(pg_ref)))
        q.put((rank, ok_fwd, ok_g0, err_pg))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,halo,T", [
    (2, (16, 12), 2, 5),            # exchange every step
    (2, (16, 12), 4, 7),            # wide halo: 2 steps per exchange, T not a multiple
    (3, (20, 8), 6, 8),             # uneven split (7,7,6), 3 steps per exchange
    (2, (12, 6, 8), 4, 4),          # 3D slabs
    (1, (8, 10), 4, 5),             # single rank: local periodic wrap
])
def test_slab_rollout_matches_single_domain(world, shape, halo, T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rendezvous()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, halo, T, 3, "float64", q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_fwd, ok_g0, err_pg in sorted(res):
        assert ok_fwd, f"rank {rank}: forward interior differs from the single-domain rollout"
        assert ok_g0, f"rank {rank}: dL/dh0 interior differs"
        assert err_pg < 1e-12, f"rank {rank}: all-reduced parameter gradient rel err {err_pg}"


def test_split_extent_and_scatter():
    sys.path.insert(0, ROOT)
    from percnn_amd import slab
    assert slab.split_extent(256, 8) == [(32 * r, 32 * r + 32) for r in range(8)]
    assert slab.split_extent(20, 3) == [(0, 7), (7, 14), (14, 20)]
    full = torch.arange(2 * 10 * 3, dtype=torch.float32).reshape(2, 10, 3)
    loc = slab.scatter_slab(full, 1, 2, 2)
    assert loc.shape == (2, 9, 3) and torch.equal(loc[:, 2:7], full[:, 5:10]) and loc[:, :2].abs().sum() == 0
    with pytest.raises(ValueError):
        slab.scatter_slab(full, 0, 5, 4)


def _probe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    try:
        from percnn_amd import slab
        out = []
        for round_ in range(2):                             # two process groups in a row: the exchanger cache must follow
            dist.init_process_group("gloo", init_method=f"file://{port}.{round_}", rank=rank, world_size=world)
            sample = torch.rand(2, 6 + 2 * 2, 4, 8, generator=torch.Generator().manual_seed(rank))
            keep = sample.clone()
            name, report = slab.probe_transport(sample, 2, candidates=("dist",))
            ex = slab.make_exchanger(prefer_rccl=False, transport="dist")
            again = slab.make_exchanger(prefer_rccl=False, transport="dist")
            out.append((name, report["picked"], report["dist"]["usable_on_every_rank"],
                        report["dist"]["halos_equal_portable_exchange"], bool(torch.equal(sample, keep)), ex is again, id(ex)))
            work = sample.clone()
            ex.exchange(work, 2, 2)                          # the cached exchanger works on THIS group
            slab.close_exchangers()
            dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:
        q.put((rank, "error", repr(e)[:500]))
        raise


def test_transport_probe_and_exchanger_cache_across_process_groups():
    """slab.probe_transport (start-up probe instead of a default, VERDICT r2 #2d) on a gloo ring of two: the portable
    transport is usable on every rank, its halos equal the reference exchange, the sample is left untouched; and
    make_exchanger's cache hands out ONE exchanger per live process group -- after destroy_process_group() + a new
    init_process_group() a fresh one (it used to be keyed on id(group) / None for the life of the process)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    port = _rendezvous()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not (len(r) == 3 and r[1] == "error"), r
        rank, rounds = r
        for name, picked, usable, equal, untouched, cached, _ in rounds:
            assert name == picked == "dist" and usable and equal and untouched and cached
