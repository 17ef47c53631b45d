"""GPU, multi-process: the N > 1 slab path with the REAL HIP slab kernels -- world_size 2 and 3 on ONE MI355X.

RCCL refuses several ranks on one device, and the build box has one GPU per call, so the ranks share cuda:0 and the ring
exchange runs through the portable ``HaloExchanger`` (gloo; faces staged through pinned host buffers) or through the
PEER-MAILBOX transport (``PeerHaloExchanger``: hipIpc-mapped mailboxes of the other processes, put / take kernels, epoch
flags -- the very code path of a multi-GPU node, here with every "peer" on the same device).  What is exercised
on hardware here, for the first time with more than one process: slab scatter, the skip-schedule forward with wide halos,
the adjoint sweep with 2-plane exchanges, fused moments in the slab sweep, the gradient all-reduce -- against the
single-domain rollout of the same kernels (state and dL/dh0 bit-identical, parameter gradients to reduction round-off)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, halo, T, hc, dtype_name, overlap, transport, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # file rendezvous (`port` is a fresh path): no TCP port to lose to another process between choosing and binding it
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        import percnn_amd as pa
        from percnn_amd import slab
        from util import random_block
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dtype = np.dtype(dtype_name)
        ndim = len(shape)
        rs = np.random.RandomState(3)
        P = torch.tensor(random_block(hc, ndim, dtype, 5, scale=0.3), device=dev)
        h0 = torch.tensor(rs.uniform(0, 1, (2,) + shape).astype(dtype), device=dev)
        # single-domain reference with the same kernels (every rank recomputes it)
        traj_ref = torch.empty((T + 1, 2) + shape, dtype=h0.dtype, device=dev)
        traj_ref[0] = h0
        pa.rollout_fwd_(traj_ref, P)
        g_ref = torch.tensor(rs.uniform(-1, 1, tuple(traj_ref.shape)).astype(dtype), device=dev)
        g0_ref, pg_ref = pa.rollout_bwd(traj_ref, g_ref, P)
        # float64 yardstick for the float32 parameter gradients (two reduction orders of heavily cancelling sums): the same
        # rollout in float64 is "exact" at this scale; the slab path may be as far from it as the plain path is
        pg_exact = None
        if dtype == np.float32:
            P64, t64 = P.double(), torch.empty(traj_ref.shape, dtype=torch.float64, device=dev)
            t64[0] = h0.double()
            pa.rollout_fwd_(t64, P64)
            _, pg_exact = pa.rollout_bwd(t64, g_ref.double(), P64)

        ex = slab.make_exchanger(prefer_rccl=False, transport=transport)
        want = slab.PeerHaloExchanger if transport == "peer" else slab.HaloExchanger
        assert type(ex) is want and (ex.rank, ex.world) == (rank, world)
        lo, hi = slab.split_extent(shape[0], world)[rank]
        n = hi - lo
        local0 = slab.scatter_slab(h0, rank, world, halo)
        traj = torch.zeros((T + 1,) + tuple(local0.shape), dtype=local0.dtype, device=dev)
        traj[0] = local0
        slab.slab_rollout_fwd_(traj, P, ex, halo, overlap=overlap)
        ok_fwd = bool(torch.equal(traj[:, :, halo:halo + n], traj_ref[:, :, lo:hi]))
        g_local = torch.zeros_like(traj)
        g_local[:, :, halo:halo + n] = g_ref[:, :, lo:hi]
        g0, pg = slab.slab_rollout_bwd(traj, g_local, P, ex, halo, overlap=overlap)
        ok_g0 = bool(torch.equal(g0[:, halo:halo + n], g0_ref[:, lo:hi]))
        if pg_exact is None:
            err_pg, err_plain = float((pg - pg_ref).norm() / pg_ref.norm()), 0.0
        else:
            err_pg = float((pg - pg_exact).norm() / pg_exact.norm())
            err_plain = float((pg_ref - pg_exact).norm() / pg_exact.norm())
        # the autograd wrapper (what a training script calls) on the same split
        loc = local0.clone().requires_grad_(True)
        out = slab.slab_rollout(loc, P, T, halo=halo, ex=ex)
        (out[:, :, halo:halo + n] * g_ref[:, :, lo:hi]).sum().backward()
        ok_auto = bool(torch.equal(loc.grad[:, halo:halo + n], g0_ref[:, lo:hi]))
        # no exchanger named: the transport is picked by the start-up probe (RCCL cannot come up with two ranks on one device:
        # rejected on every rank; the mailboxes are accepted only if they reproduce the portable exchange bit for bit)
        with torch.no_grad():
            out_p = slab.slab_rollout(local0.clone(), P, T, halo=halo)
        ok_auto = ok_auto and bool(torch.equal(out_p[:, :, halo:halo + n], traj_ref[:, :, lo:hi]))
        if transport == "peer":
            ok_auto = ok_auto and ex.status() == 0            # no take ever timed out
            # the Python orchestration (one exchange call per step) over the same mailboxes
            class PyLoop(type(ex)):
                def native_ring(self):
                    return False, None
            ex.__class__ = PyLoop
            traj2 = torch.zeros_like(traj)
            traj2[0] = local0
            slab.slab_rollout_fwd_(traj2, P, ex, halo, overlap=overlap)
            g0b, pgb = slab.slab_rollout_bwd(traj2, g_local, P, ex, halo, overlap=overlap)
            ok_fwd = ok_fwd and bool(torch.equal(traj2[:, :, halo:halo + n], traj_ref[:, :, lo:hi]))
            ok_g0 = ok_g0 and bool(torch.equal(g0b[:, halo:halo + n], g0_ref[:, lo:hi])) and ex.status() == 0
        q.put((rank, ok_fwd, ok_g0, err_pg, ok_auto, err_plain))
    except Exception as e:                             # report instead of leaving the parent waiting for its queue time-out
        q.put((rank, "error", repr(e)[:500]))
        raise
    finally:
        try:
            slab.close_exchangers()
        except Exception:
            pass
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["dist", "peer"])
@pytest.mark.parametrize("world,shape,halo,T,hc,dtype,overlap", [
    (2, (16, 12, 64), 4, 5, 0, "float32", False),       # 3D, wide halo (2 steps per exchange), fused moments in the sweep
    (2, (16, 12, 64), 4, 5, 0, "float32", True),        # faces first + asynchronous exchange + planes in between
    (3, (20, 8, 16), 2, 4, 2, "float32", False),        # uneven split (7,7,6), factored block: sweep + slab_wgrad
    (2, (24, 40), 4, 6, 0, "float64", False),           # 2D slabs, float64
    (8, (64, 24, 64), 4, 4, 0, "float32", False),       # BASELINE configs[4]'s decomposition in small: eight ranks, prev != next
                                                        # on every rank, 8 planes each, wide halo
])
def test_multi_process_slab_rollout_on_one_gpu(world, shape, halo, T, hc, dtype, overlap, transport, hip_device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import tempfile
    port = os.path.join(tempfile.mkdtemp(prefix="percnn_rdzv_"), "store")
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, halo, T, hc, dtype, overlap, transport, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    errors = [r for r in res if len(r) == 3 and r[1] == "error"]
    for p in procs:
        p.join(timeout=120)
    assert not errors, errors
    for p in procs:
        assert p.exitcode == 0
    for rank, ok_fwd, ok_g0, err_pg, ok_auto, err_plain in sorted(res):
        # float32: measured against a float64 rollout of the same problem -- "as close to it as the single-domain path, within
        # a factor of three" (was a flat 5e-4 against the plain path); float64: against the plain path
        tol = 3.0 * err_plain + 2e-6 if dtype == "float32" else 1e-11
        assert ok_fwd, f"rank {rank}: forward interior differs from the single-domain rollout"
        assert ok_g0, f"rank {rank}: dL/dh0 interior differs"
        assert ok_auto, f"rank {rank}: autograd wrapper dL/dh0 differs"
        assert err_pg < tol, f"rank {rank}: all-reduced parameter gradient rel err {err_pg}"


def test_bench_two_ranks_on_one_gpu(hip_device):
    """bench.py --gpus 2, launched exactly as the driver launches it (python -m torch.distributed.run ...), with both ranks on
    cuda:0 (PERCNN_BENCH_ONE_GPU: gloo as the control plane) and the sharded series on small grids (PERCNN_BENCH_SMALL): the
    N > 1 control flow of the harness -- replicas of the headline, the isolated slab child per rank, the transport probe, the
    weak- and strong-scaling series with their bit-identity checks, the watchdog, ONE JSON line from rank 0 -- runs end to
    end before the driver's 8-GPU run does (VERDICT r2 #2b)."""
    import json
    import subprocess
    env = dict(os.environ, PERCNN_BENCH_ONE_GPU="1", PERCNN_BENCH_SMALL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    # N > 1: the top level is the SHARDED path (the fixed grid cut into N slabs), the 2D replicas are an extra (VERDICT r4 #1d)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "strong", out
    assert "sharded_headline_missing" not in out, out["sharded_headline_missing"]
    assert out["config"]["workload"].startswith("gs3d_32 strong scaling") and out["config"]["grid"] == [32, 32, 32]
    assert out["config"]["parallelism"] == "spatial slabs x2" and out["config"]["points_per_rank"] == 16 * 32 * 32
    assert out["transport"] and out["ranks_seen_by_transport"] == 2
    assert out["forward_state_equals_single_domain_rollout"] is True
    assert out["n1_anchor"]["steps_per_sec"] > 0 and out["speedup_vs_n1_anchor"] > 0
    assert abs(out["value"] - out["config"]["T"] / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert out["roofline"]["peak"] == 16000.0 and 0 < out["roofline"]["frac"] < 1
    rep = out["replicas_2d"]
    assert rep["scaling"] == "weak" and rep["value"] > 0 and rep["also"]["gs3d_128"]["value"] > 0
    sl = out["slab_3d"]
    assert "error" not in sl, sl
    assert "incomplete" not in sl, sl
    assert sl["transport_probe"]["picked"] in ("dist", "peer") and sl["transport_probe"]["peer"]["halos_equal_portable_exchange"]
    weak = sl["weak_scaling"]["by_transport"]
    assert set(weak) == {"dist", "peer", "dist_faces_first_overlap", "peer_faces_first_overlap", "peer_fused_adjoint_put"} and all(w.get("forward_state_equals_single_domain_rollout") for w in weak.values()), weak
    for key in ("dist_faces_first_overlap", "peer_faces_first_overlap"):
        assert "overlap=1" in weak[key]["workload"] and weak[key]["us_per_time_step_fwd_bwd"] > 0
    assert weak["peer_fused_adjoint_put"]["us_per_time_step_fwd_bwd"] > 0
    assert sl["strong_scaling"]["schedule"].startswith(("faces first", "exchange between two steps"))
    for key, w in weak.items():
        if key.endswith("_overlap") or key == "peer_fused_adjoint_put":
            continue                              # (timed without the compute / exchange split)
        b = w["per_time_step_us"]
        assert b["total"] > 0 and b["compute_alone"] > 0 and b["exchanges_alone"] > 0
    strong = sl["strong_scaling"]["by_grid"]
    assert set(strong) == {"32^3", "16^3"}
    for g in strong.values():
        assert g.get("forward_state_equals_single_domain_rollout") is True and g["steps_per_sec_fwd_bwd"] > 0, g


def test_bench_eight_ranks_on_one_gpu_at_256cubed(hip_device):
    """BASELINE configs[4] as the driver will launch it on an 8-GPU node -- ``python -m torch.distributed.run --nproc-per-node 8
    bench.py --gpus 8`` -- with all eight ranks on cuda:0 (no 8-GPU node exists for the builder): 256^3 cut into
    8 slabs of 32 planes, previous != next neighbour on every rank, the transport probe, both transports, the strong-scaling
    series at the REAL grid sizes with its bit-identity check against the single-domain rollout.  What it cannot show is the
    wire: eight processes share one device's memory here."""
    import json
    import subprocess
    env = dict(os.environ, PERCNN_BENCH_ONE_GPU="1", PERCNN_BENCH_HEADLINE_T="12")     # (T = 100 on the driver's node)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PERCNN_BENCH_SMALL"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-also", "--slab-timeout", "900"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["value"] > 0
    # the top level IS configs[4]: the 256^3 grid cut into 8 slabs, strong scaling, transport and rank count in the line
    assert "sharded_headline_missing" not in out, out["sharded_headline_missing"]
    assert out["scaling"] == "strong" and out["config"]["grid"] == [256, 256, 256] and "configs[4]" in out["config"]["workload"]
    assert out["config"]["parallelism"] == "spatial slabs x8" and out["config"]["points_per_rank"] == 32 * 256 * 256
    assert out["transport"] and out["ranks_seen_by_transport"] == 8 and out["schedule"]
    assert out["forward_state_equals_single_domain_rollout"] is True and out["frames_compared"] == 11
    assert out["n1_anchor"]["steps_per_sec"] > 0 and out["speedup_vs_n1_anchor"] > 0
    assert out["replicas_2d"]["value"] > 0 and out["replicas_2d"]["scaling"] == "weak"
    sl = out["slab_3d"]
    assert "error" not in sl and "incomplete" not in sl, sl
    probe = sl["transport_probe"]
    # Eight processes that spin-wait on each other's kernels cannot count on running concurrently on ONE device (seen: some takes
    # wait for a put the queue scheduler has not started, and time out) -- the mailboxes are either bit-identical to the portable
    # exchange or REJECTED by the probe within its bound, never silently wrong and never a 300 s stall per exchange
    assert probe["picked"] in ("dist", "peer"), probe
    if probe["picked"] == "peer":
        assert probe["peer"]["usable_on_every_rank"] and probe["peer"]["halos_equal_portable_exchange"], probe
    strong = sl["strong_scaling"]["by_grid"]
    assert set(strong) == {"256^3", "128^3"}, strong
    for key, g in strong.items():
        assert g.get("forward_state_equals_single_domain_rollout") is True and g["steps_per_sec_fwd_bwd"] > 0, (key, g)
    assert sl["weak_scaling"]["by_transport"]["dist"].get("forward_state_equals_single_domain_rollout") is True


# ---------------------------------------------------------------------------------------------------------------------------
# the native loops' ring callbacks with world > 1: an in-process fake of RCCL's group semantics (tests/fake_ring.py)
# ---------------------------------------------------------------------------------------------------------------------------
def _fake_ring_run(world, shape, halo, T, overlap, packed, fail_at=None, dtype=torch.float32):
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import percnn_amd as pa
    from percnn_amd import slab
    from fake_ring import FakeFabric, FakeRingExchanger
    from util import random_block
    dev = torch.device("cuda:0")
    ndim = len(shape)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    rs = np.random.RandomState(11)
    P = torch.tensor(random_block(0, ndim, np.dtype(npdt), 5, scale=0.3), device=dev)
    h0 = torch.tensor(rs.uniform(0, 1, (2,) + shape).astype(npdt), device=dev)
    ref = torch.empty((T + 1, 2) + shape, dtype=dtype, device=dev)
    ref[0] = h0
    pa.rollout_fwd_(ref, P)
    g_ref = torch.tensor(rs.uniform(-1, 1, tuple(ref.shape)).astype(npdt), device=dev)
    g0_ref, pg_ref = pa.rollout_bwd(ref, g_ref, P)
    torch.cuda.synchronize()
    fabric = FakeFabric(world, take_timeout_s=10.0)
    exs = [FakeRingExchanger(fabric, r, fail_at=fail_at if r == 0 else None, packed=packed) for r in range(world)]
    cuts = slab.split_extent(shape[0], world)
    out, errs = [None] * world, [None] * world

    def rank_main(r):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            lo, hi = cuts[r]
            n = hi - lo
            with torch.cuda.stream(st):
                traj = torch.zeros((T + 1, 2, n + 2 * halo) + shape[1:], dtype=dtype, device=dev)
                traj[0, :, halo:halo + n] = h0[:, lo:lo + n]
                gt = torch.zeros_like(traj)
                gt[:, :, halo:halo + n] = g_ref[:, :, lo:lo + n]
                slab.slab_rollout_fwd_(traj, P, exs[r], halo, overlap=overlap)
                g0, pg = slab.slab_rollout_bwd(traj, gt, P, exs[r], halo, overlap=overlap)
                st.synchronize()
                out[r] = (traj[:, :, halo:halo + n].clone(), g0[:, halo:halo + n].clone(), pg.clone(), lo, n)
        except Exception as e:
            errs[r] = e
            try:
                fabric.barrier.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in ths), "a rank hangs"
    return out, errs, exs, fabric, (ref, g0_ref, pg_ref)


@pytest.mark.parametrize("world,shape,halo,T,overlap,packed", [
    (2, (16, 24, 32), 4, 7, False, True),            # prev == next: per peer, sends and receives pair up in issue order
    (3, (24, 16, 64), 4, 6, False, True),            # prev != next
    (3, (24, 16, 64), 2, 5, False, False),           # per-species messages straight from the slab (4 + 4 per exchange)
    (2, (16, 24, 32), 4, 6, True, True),             # faces first, exchange on the side stream
    (4, (32, 256, 256), 4, 4, False, True),          # the per-rank shape class of configs[4] (64-chunk rows), 2 MiB faces
])
def test_native_slab_loops_through_a_fake_rccl_ring(world, shape, halo, T, overlap, packed, hip_device):
    """the native loops' ncclGroupStart / Send / Recv / End call pattern with MORE THAN ONE rank -- group
    discipline, neighbour addressing (prev != next), pairing order for prev == next, packed and per-species messages, the
    overlap schedule's side stream -- before real hardware sees it.  Interior states and dL/dh0 bit-identical to the
    single-domain rollout, gradient block to reduction round-off."""
    out, errs, exs, fabric, (ref, g0_ref, pg_ref) = _fake_ring_run(world, shape, halo, T, overlap, packed)
    assert all(e is None for e in errs), errs
    assert not fabric.errors, fabric.errors
    k = halo // 2
    n_exch = (T + k - 1) // k + T                    # forward: one per k steps; adjoint: one per step
    for r, ex in enumerate(exs):
        assert not ex.in_group and ex.groups == n_exch, (r, ex.groups, n_exch)
        assert ex.ops == n_exch * (4 if packed else 8), (r, ex.ops)
        # every operation of a rank names one of its two ring neighbours, and both of them when they differ
        peers = {p for (rk, _, p, _) in fabric.log if rk == r}
        assert peers == {ex.prev, ex.next}, (r, peers)
    for q in fabric.fifo.values():
        assert q.empty()                             # every send met its receive
    for r in range(world):
        traj, g0, pg, lo, n = out[r]
        assert torch.equal(traj, ref[:, :, lo:lo + n]), r
        assert torch.equal(g0, g0_ref[:, lo:lo + n]), r
        err = float((pg - pg_ref).norm() / pg_ref.norm())
        assert err < 2e-5, (r, err)
        assert torch.equal(pg, out[0][2])            # the all-reduced block is the same on every rank


@pytest.mark.parametrize("fail_at", [("send", 3), ("recv", 2), ("group_end", 2), ("group_start", 1)])
def test_native_slab_loops_hand_back_a_failing_ring_call(fail_at, hip_device):
    """an RCCL call that returns an error inside the native loop must surface as that loop's return code (-> RuntimeError on
    the rank it happened on), not be swallowed; the other rank ends with a bounded 'no matching send', not a hang"""
    out, errs, exs, fabric, _ = _fake_ring_run(2, (16, 16, 32), 4, 5, False, True, fail_at=fail_at)
    assert isinstance(errs[0], RuntimeError) and "slab_rollout" in str(errs[0]), errs
    assert errs[1] is None or isinstance(errs[1], (RuntimeError, Exception))


def test_slab_schedules_against_an_injected_wire(hip_device, tmp_path):
    """to self every message arrives as fast as a device copy, so the overlap schedules of the native slab loops had
    only ever shown their own cost.  examples/slab_delay_ring.cpp drives percnn_pi_slab_rollout_fwd / _bwd on the per-rank shape of
    configs[4] (32 x 256^2, halo 4) with an INJECTED wire of DESIGN.md 6's xGMI budget -- 30 us per forward exchange (2 MiB faces
    every two steps), 16 us per adjoint exchange (1 MiB every step) -- through (a) a percnn_pi_halo_ring of C++ callbacks with RCCL's
    group semantics and (b) the peer mailboxes (option peer_wire_us: the put holds its arrival flag back).
    What the numbers show (profiles/r06_slab_injected_wire.txt) and what is asserted here:
      (i)   the plain schedule pays the wire in full (its time grows by > 80 % of the wire) -- the harness measures what it claims;
      (ii)  the put fused into the step / sweep launches never loses to the plain schedule once there is a wire, and hides part of
            the adjoint's (an in-launch put can only hide what is left of ITS launch after the faces are stored);
      (iii) faces-first + side stream (three launches and two cross-stream hops per step) does NOT pay at this size even with a
            wire: asserted as measured, so a change that makes it pay shows up;
      (iv)  all schedules and wires give the same bits."""
    import re
    import subprocess
    import percnn_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "slab_delay_ring")
    csrc = os.path.dirname(percnn_amd.LIB_PATH)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "slab_delay_ring.cpp"), "-L" + csrc, "-lpercnn_pi", "-Wl,-rpath," + csrc, "-o", exe])
    wire = 30.0 / 2 + 16.0                                      # per fwd+bwd time step
    for mode in ("ring", "peer"):
        out = subprocess.run([exe, "40", "5", "30", "16", "4", mode], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "bitwise: identical" in out.stdout and "slab_delay_ring ok" in out.stdout      # (iv)
        rows = {}
        for m in re.finditer(r"RESULT (.+?)\s+fwd_wire_us\s+([\d.]+) bwd_wire_us\s+([\d.]+) \| fwd\s+([\d.]+) bwd\s+([\d.]+) total\s+([\d.]+)", out.stdout):
            rows[m.group(1).strip()] = tuple(float(m.group(i)) for i in range(2, 7))
        print(out.stdout)
        p0, o0, p1, o1 = (rows[k][4] for k in ("plain, no wire", "overlap, no wire", "plain, wire", "overlap, wire"))
        assert p1 - p0 > 0.8 * wire, (mode, p0, p1, wire)       # (i)
        assert o1 > p1 and o0 > p0, (mode, o0, p0, o1, p1)      # (iii)
        if mode == "peer":
            f0, f1 = rows["fused put, no wire"][4], rows["fused put, wire"][4]
            assert f1 < p1 + 0.1 * wire, (f1, p1)               # (ii)
            assert rows["fused put, wire"][3] - rows["fused put, no wire"][3] < 0.9 * 16.0, rows      # part of the adjoint's wire hidden
