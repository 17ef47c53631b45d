"""Seeded random sweep over shapes / block kinds / rollout lengths / frame masks: every kernel family and every
dispatch boundary (tile eligibility, ragged tiles, 32x16 vs 32x32 tiles, plane streaming, fused vs separate gradient
reduction, vector width fallbacks) against the plain-C oracle -- state and adjoint fields bit-identical, parameter
gradients to reduction round-off."""
import numpy as np
import pytest
import torch

from util import o_rollout_bwd, o_rollout_fwd, random_block, rel_l2

pytestmark = pytest.mark.gpu


def _cases():
    rs = np.random.RandomState(20240928)
    out = []
    w2 = [4, 6, 8, 12, 20, 24, 28, 32, 36, 44, 48, 52, 64, 96, 100, 132, 256]
    h2 = [2, 3, 5, 8, 17, 23, 24, 25, 31, 32, 33, 40, 47, 48, 49, 64, 70, 100, 129]
    for i in range(120):
        ndim = 2 if i % 3 else 3
        if ndim == 2:
            shape = (int(rs.choice(h2)), int(rs.choice(w2)))
        else:
            shape = (int(rs.choice([2, 3, 5, 8, 12])), int(rs.choice([2, 4, 5, 8, 16])), int(rs.choice([4, 6, 8, 20, 64, 128, 256])))
        hc = int(rs.choice([0, 0, 2, 3, 4, 8]))
        dtype = np.float32 if rs.rand() < 0.7 else np.float64
        T = int(rs.choice([1, 2, 3, 4, 5, 7, 9, 13]))
        masked = bool(rs.rand() < 0.4)
        out.append((i, ndim, shape, hc, dtype, T, masked))
    return out


def _large2d_cases():
    """2D grids with more than 128 tiles of 32x32: the regime of the fused tile sweep (float32 poly) next to the split
    schedule (other block kinds / float64) -- ragged edges, rollout lengths around multiples of K = 4, sparse masks."""
    rs = np.random.RandomState(777)
    shapes = [(384, 384), (416, 352), (500, 396), (640, 260), (356, 676), (512, 512)]
    out = []
    for i in range(14):
        shape = shapes[i % len(shapes)]
        hc = [0, 0, 0, 8, 0, 2, 0][i % 7]
        dtype = np.float64 if i % 5 == 4 else np.float32
        T = int(rs.choice([3, 4, 5, 8, 9, 11]))
        out.append((200 + i, 2, shape, hc, dtype, T, bool(i % 2)))
    return out


@pytest.mark.parametrize("case", _cases() + _large2d_cases(), ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[2]))}-hc{c[3]}-{np.dtype(c[4]).name}-T{c[5]}{'-mask' if c[6] else ''}")
def test_random_rollout_vs_oracle(case, hip_device):
    import percnn_amd as pa
    i, ndim, shape, hc, dtype, T, masked = case
    rs = np.random.RandomState(1000 + i)
    P = random_block(hc, ndim, dtype, 50 + i, scale=0.3)
    h0 = rs.uniform(0.1, 0.9, (2,) + shape).astype(dtype)
    traj_o = o_rollout_fwd(h0, P, T)
    g = rs.standard_normal(traj_o.shape).astype(dtype)
    mask = None
    if masked:
        mask = [bool(b) for b in rs.rand(T + 1) < 0.5]
        g[[not m for m in mask]] = 0
    g0_o, pg_o = o_rollout_bwd(traj_o, g, P)
    for stream in ((1, 2) if ndim == 3 else (1,)):
        pa.set_option("stream3d", stream)
        try:
            traj = torch.empty((T + 1, 2) + shape, dtype=torch.from_numpy(h0).dtype, device=hip_device)
            traj[0] = torch.from_numpy(h0).to(hip_device)
            Pd = torch.from_numpy(P).to(hip_device)
            pa.rollout_fwd_(traj, Pd)
            assert np.array_equal(traj.cpu().numpy(), traj_o)
            gd = torch.from_numpy(g).to(hip_device)
            if masked:                      # masked-out frames must not be read: poison them
                gd[[not m for m in mask]] = float("nan")
            g0, pg = pa.rollout_bwd(traj, gd, Pd, frame_mask=mask)
        finally:
            pa.set_option("stream3d", 1)
        assert np.array_equal(g0.cpu().numpy(), g0_o)
        denom = max(np.linalg.norm(pg_o), 1e-30)
        assert np.linalg.norm(pg.cpu().numpy() - pg_o) / denom < (1e-4 if dtype == np.float32 else 1e-10)
