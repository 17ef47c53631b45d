"""Physics-residual loss of a rollout -- the loss consumer right after the hot path.

Reference: ``loss_generator.get_phy_Loss`` + ``loss_gen`` (DataDrivenModeling/2d_gs_rd/train_2drd.py:
270-353, 3d_gs_rd/train_3drd.py:287-346, ForwardSimulationOfPDEs/2d_lambda_omega/percnn_LO_eqn.py:283-357).
There the trajectory is periodically padded, convolved frame by frame with the dense 5x5(x5) Laplacian,
permuted/reshaped into a ``Conv1d`` for the time difference, and the analytic PDE right-hand side is
assembled from ~15 full-trajectory temporaries.  Here the residual of the TRUE equation

    R_s(f, x) = D_s * Lap(h_f)_s + r_s(h_f) - (h_{f+1,s} - h_{f,s}) / dt

is one frame-parallel kernel launch (and one for its adjoint).  The true equation is handed over as a
pre-contracted coefficient block (36 entries), the same format the ``poly`` reaction mode uses.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch

from . import _lib
from . import functional as F_pi

# monomial order of the coefficient block: 1, u, v, u^2, uv, v^2, u^3, u^2 v, u v^2, v^3
_IDX = {"1": 0, "u": 1, "v": 2, "uu": 3, "uv": 4, "vv": 5, "uuu": 6, "uuv": 7, "uvv": 8, "vvv": 9}


def pde_block(w_laplace: torch.Tensor, dt: float, Du: float, Dv: float, ru: dict, rv: dict) -> torch.Tensor:
    """Coefficient block of  u_t = Du Lap u + ru(u,v),  v_t = Dv Lap v + rv(u,v)  on the grid of a cell:
    stencil taps taken from the cell's (pre-scaled) ``W_laplace.weight``; ru / rv map monomial names to
    coefficients, e.g. Gray-Scott ru = {"1": f, "u": -f, "uvv": -1}."""
    F_pi.check_star_stencil(w_laplace)
    dev, dtype = w_laplace.device, w_laplace.dtype
    ndim = w_laplace.dim() - 2
    flat = torch.cat([torch.tensor([dt, Du, Dv], dtype=dtype, device=dev), w_laplace.detach().reshape(-1)])
    head = flat.index_select(0, F_pi._gather_index(1, ndim, dev)[:16])      # header slots only reference flat[0:3+5^ndim]
    c = torch.zeros(20, dtype=torch.float64)
    for s, r in enumerate((ru, rv)):
        for k, val in r.items():
            c[10 * s + _IDX["".join(sorted(k))]] = val
    return torch.cat([head.detach(), c.to(dtype).to(dev)])


def gray_scott_block(cell, Du: float, Dv: float, f: float, k: float) -> torch.Tensor:
    """u_t = Du Lap u - u v^2 + f (1 - u);  v_t = Dv Lap v + u v^2 - (f + k) v
    (2D: train_2drd.py:321-327 Du=2e-5, Dv=Du/4, f=1/25, k=3/50; 3D: train_3drd.py:316-322 Du=.2, Dv=.1, f=.025, k=.055)."""
    return pde_block(cell.W_laplace.weight, cell.dt, Du, Dv, {"1": f, "u": -f, "uvv": -1.0}, {"uvv": 1.0, "v": -(f + k)})


def lambda_omega_block(cell, D: float = 0.1) -> torch.Tensor:
    """u_t = D Lap u + (1-u^2-v^2) u + (u^2+v^2) v;  v_t = D Lap v - (u^2+v^2) u + (1-u^2-v^2) v   (percnn_LO_eqn.py:339-340)."""
    return pde_block(cell.W_laplace.weight, cell.dt, D, D,
                     {"u": 1.0, "uuu": -1.0, "uvv": -1.0, "uuv": 1.0, "vvv": 1.0},
                     {"uuu": -1.0, "uvv": -1.0, "v": 1.0, "uuv": -1.0, "vvv": -1.0})


def _call(name, traj, *ptrs, Q, nframes):
    shape = traj.shape[2:]
    f = getattr(_lib.lib(), name + F_pi._SUF[traj.dtype])
    with torch.cuda.device(traj.device):
        _lib.check(f(traj.data_ptr(), *[p.data_ptr() for p in ptrs], Q.data_ptr(), len(shape), _lib.shape_arg(shape),
                     nframes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), name)


class PhysicsResidualFunction(torch.autograd.Function):
    """traj [F+1, 2, *S] -> residual [F, 2, *S]; the equation block is a constant."""

    @staticmethod
    def forward(ctx, traj, Q):
        F_pi._require(traj, "traj"); F_pi._require(Q, "pde block", traj.dtype)
        if Q.numel() != F_pi.NPOLY:
            raise RuntimeError("percnn_amd: the physics residual takes a 36-entry pre-contracted equation block")
        nframes = traj.shape[0] - 1
        R = torch.empty((nframes,) + tuple(traj.shape[1:]), dtype=traj.dtype, device=traj.device)
        _call("percnn_pi_residual_fwd_", traj, R, Q=Q, nframes=nframes)
        ctx.save_for_backward(traj, Q)
        return R

    @staticmethod
    def backward(ctx, gR):
        traj, Q = ctx.saved_tensors
        gR = gR.contiguous()
        nframes = traj.shape[0] - 1
        g = torch.zeros_like(traj)
        _call("percnn_pi_residual_bwd_", traj, gR, g[:nframes], Q=Q, nframes=nframes)   # d/d traj[f]
        g[1:] -= gR / Q[0]                                                             # d/d traj[f+1] = -G/dt
        return g, None


class PhysicsLossFunction(torch.autograd.Function):
    """output [F+2, 2, *S] (a rollout incl. its last frame, which ``loss_gen`` drops) -> the scalar residual loss over frames
    0 .. F-1, as ONE node: the forward is one reducing pass over the trajectory (no residual tensor), the backward two launches
    that write dL/d output completely (``percnn_pi_residual_sqloss_*``).  The expression it replaces -- residual, square, two
    weight multiplies, two sums, and autograd's mirror image of them, then zero-fill / adjoint / divide / subtract -- moved
    ~20x the trajectory's bytes: lambda-omega 512^2 x 400 spent 13 of its 16 ms per training iteration there."""

    @staticmethod
    def forward(ctx, output, Q, nres, weighted):
        F_pi._require(output, "output"); F_pi._require(Q, "pde block", output.dtype)
        if Q.numel() != F_pi.NPOLY:
            raise RuntimeError("percnn_amd: the physics residual takes a 36-entry pre-contracted equation block")
        if nres < 1 or output.shape[0] < nres + 1:
            raise RuntimeError("percnn_amd: the residual of frame f needs frame f + 1")
        L = _lib.lib()
        ws = torch.empty(L.percnn_pi_residual_sqloss_workspace_bytes() // 8, dtype=torch.float64, device=output.device)
        loss = torch.empty((), dtype=output.dtype, device=output.device)
        shape = output.shape[2:]
        f = getattr(L, "percnn_pi_residual_sqloss_" + F_pi._SUF[output.dtype])
        with torch.cuda.device(output.device):
            _lib.check(f(output.data_ptr(), Q.data_ptr(), len(shape), _lib.shape_arg(shape), int(nres), int(bool(weighted)),
                         loss.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "residual_sqloss")
        ctx.save_for_backward(output, Q)
        ctx.nres, ctx.weighted = int(nres), bool(weighted)
        return loss

    @staticmethod
    def backward(ctx, gl):
        output, Q = ctx.saved_tensors
        gl = gl.to(output.dtype).contiguous()
        shape = output.shape[2:]
        scratch = torch.empty((ctx.nres,) + tuple(output.shape[1:]), dtype=output.dtype, device=output.device)
        g = torch.empty_like(output)
        f = getattr(_lib.lib(), "percnn_pi_residual_sqloss_bwd_" + F_pi._SUF[output.dtype])
        with torch.cuda.device(output.device):
            _lib.check(f(output.data_ptr(), gl.data_ptr(), Q.data_ptr(), len(shape), _lib.shape_arg(shape), ctx.nres,
                         output.shape[0], int(ctx.weighted), scratch.data_ptr(), g.data_ptr(),
                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "residual_sqloss_bwd")
        return g, None, None, None


def physics_residual(traj: torch.Tensor, Q: torch.Tensor) -> torch.Tensor:
    return PhysicsResidualFunction.apply(traj.contiguous(), Q)


def physics_loss(output: torch.Tensor, Q: torch.Tensor, reference_weighting: bool = True, fused: bool = True) -> torch.Tensor:
    """Drop-in for ``loss_gen(output, loss_func)`` (train_2drd.py:340-353): MSE of f_u plus MSE of f_v over
    frames ``output[0:-2]``.  The reference pads 2 cells on the low and 3 on the high side, so it evaluates
    the residual on an (N+1)^d grid in which the first row / column / plane appears twice;
    ``reference_weighting`` reproduces that weighting exactly (False: plain mean over the periodic grid).
    One autograd node (``PhysicsLossFunction``); ``fused=False`` keeps the residual-tensor expression (tests compare the two)."""
    # (the one-node route keeps one partial sum per workgroup in a fixed 16384-slot workspace: grids beyond 2^24 points per
    # species take the residual-tensor expression)
    # and so does any grid the fused pass turns down (PERCNN_PI_ETOOLARGE: e.g. float64 256^3, or a float32 grid above 2^22
    # points that loses its 16-byte lanes -- more workgroups than partial slots; nothing has been launched then)
    if fused and output.is_cuda and 3 <= output.shape[0] <= 65535 and output[0, 0].numel() <= (1 << 24):
        try:
            return PhysicsLossFunction.apply(output.contiguous(), Q, output.shape[0] - 2, reference_weighting)
        except _lib.GridTooLargeError:
            pass
    R = physics_residual(output[:-1], Q)              # frames 0 .. len-3
    sq = R * R
    if reference_weighting:
        ndim = R.dim() - 2
        for ax in range(ndim):
            n = R.shape[2 + ax]
            w = torch.ones(n, dtype=R.dtype, device=R.device)
            w[0] = 2.0
            sq = sq * w.reshape((1, 1) + (1,) * ax + (n,) + (1,) * (ndim - ax - 1))
        denom = R.shape[0] * float(torch.tensor([n + 1 for n in R.shape[2:]]).prod())
        return sq[:, 0].sum() / denom + sq[:, 1].sum() / denom
    return sq[:, 0].mean() + sq[:, 1].mean()
