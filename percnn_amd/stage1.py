"""Stage-1 Pi-block of the equation-discovery pipeline on the MI355X matrix cores.

Host-side mirror of the reference's Stage-1 ``RCNNCell`` -- three parallel 5x5 periodic convolutions 2 -> 16
per species, Hadamard product, 1x1 contraction, FD Laplacian with a sigmoid-bounded diffusion coefficient,
explicit Euler; float32 --
    DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/rcnn_Burgers_[...].py:54-187      (dx=1/100, dt=2.5e-4, nu_up=0.01)
    DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-1/rcnn_LO_[...].py:53-180      (dx=0.2, dt=0.0125, nu_up=0.2)
with the same constructor meaning, attribute names and ``state_dict`` schema, so the reference's Stage-1
checkpoints load unchanged.  The math runs in ``libpercnn_pi.so`` (``include/percnn_pi_stage1.h``): the branch
evaluation, its input gradient and its weight gradient are ``v_mfma_f32_16x16x4_f32`` contractions.
No CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .functional import _require, _stream, check_star_stencil, _assemble_frame_grads, _observe, _scatter_observed, Frame

NP = 16 + 6 * 16 * 52 + 32 + 2          # PERCNN_PI_S1_PARAMS
_OFF_W, _OFF_W4, _OFF_B4 = 16, 16 + 4992, 16 + 4992 + 32
_idx_cache: dict = {}


def _gather_index(device) -> torch.Tensor:
    """Index into cat[dt, coef_u, coef_v, W_laplace(25), 0, {Wh1.w(800) Wh1.b(16) Wh2.w Wh2.b Wh3.w Wh3.b Wh4.w(16)
    Wh4.b(1)} for u then v] that yields the block of include/percnn_pi_stage1.h."""
    key = str(device)
    if key in _idx_cache:
        return _idx_cache[key]
    idx = np.zeros(NP, dtype=np.int64)
    idx[0:3] = (0, 1, 2)
    lap = 3
    idx[3] = lap + 12
    for i, off in enumerate((-2, -1, 1, 2)):
        idx[4 + i] = lap + (2 + off) * 5 + 2
        idx[8 + i] = lap + 2 * 5 + (2 + off)
    zero = lap + 25
    idx[12:16] = zero
    per_species = 3 * (800 + 16) + 16 + 1
    for s in range(2):
        src = zero + 1 + s * per_species
        for k in range(3):
            w0 = src + k * 816
            for j in range(16):
                base = _OFF_W + ((s * 3 + k) * 16 + j) * 52
                idx[base:base + 50] = w0 + j * 50 + np.arange(50)
                idx[base + 50] = w0 + 800 + j
                idx[base + 51] = zero
        w4 = src + 3 * 816
        idx[_OFF_W4 + s * 16:_OFF_W4 + (s + 1) * 16] = w4 + np.arange(16)
        idx[_OFF_B4 + s] = w4 + 16
    t = torch.from_numpy(idx).to(device)
    _idx_cache[key] = t
    return t


def pack_params(dt, coef_u, coef_v, w_laplace, branch: Sequence[torch.Tensor]) -> torch.Tensor:
    """Differentiable assembly of the parameter block; ``branch`` = [Wh1_u.w, Wh1_u.b, ..., Wh4_u.w, Wh4_u.b,
    (same for v)].  Gradients reach CA / CB through ``coef = nu_up * sigmoid(C)`` by stock autograd."""
    dev = w_laplace.device
    f32 = lambda t: torch.as_tensor(t, dtype=torch.float32, device=dev).reshape(-1)
    flat = torch.cat([f32(dt), f32(coef_u), f32(coef_v), w_laplace.reshape(-1), torch.zeros(1, dtype=torch.float32, device=dev)]
                     + [b.reshape(-1) for b in branch])
    return flat.index_select(0, _gather_index(dev))


def set_option(key: str, value: int) -> None:
    _lib.check(_lib.lib().percnn_pi_s1_set_option(key.encode(), int(value)), f"s1 set_option({key}={value})")


def _check(P: torch.Tensor, *states: torch.Tensor) -> None:
    _require(P, "params", torch.float32)
    if P.numel() != NP:
        raise RuntimeError(f"percnn_amd.stage1: parameter block has {P.numel()} entries, expected {NP}")
    for t in states:
        _require(t, "state", torch.float32)


def step_fwd(h: torch.Tensor, P: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h: [2,H,W] -> next state."""
    _check(P, h)
    out = torch.empty_like(h) if out is None else out
    with torch.cuda.device(h.device):
        _lib.check(_lib.lib().percnn_pi_s1_step_fwd_f32(h.data_ptr(), out.data_ptr(), P.data_ptr(),
                                                       _lib.shape_arg(h.shape[1:]), _stream()), "s1_step_fwd")
    return out


def rollout_fwd_(traj: torch.Tensor, P: torch.Tensor) -> torch.Tensor:
    """In place: traj[0] holds the initial state, frames 1..T are written.  traj: [T+1,2,H,W]."""
    _check(P, traj)
    with torch.cuda.device(traj.device):
        _lib.check(_lib.lib().percnn_pi_s1_rollout_fwd_f32(traj.data_ptr(), P.data_ptr(), _lib.shape_arg(traj.shape[2:]),
                                                          traj.shape[0] - 1, _stream()), "s1_rollout_fwd")
    return traj


def rollout_bwd(traj: torch.Tensor, g_traj: torch.Tensor, P: torch.Tensor, frame_mask=None, ws=None):
    """-> (dL/dh0 [2,H,W], dL/dparams double[NP])"""
    _check(P, traj, g_traj)
    T, shape = traj.shape[0] - 1, traj.shape[2:]
    L = _lib.lib()
    if ws is None:
        nbytes = L.percnn_pi_s1_rollout_bwd_workspace_bytes(_lib.shape_arg(shape), T)
        if nbytes == 0:
            raise RuntimeError("percnn_amd.stage1: invalid problem shape")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=traj.device)
    mask = None
    if frame_mask is not None:
        assert len(frame_mask) == T + 1
        mask = bytes(bytearray(1 if m else 0 for m in frame_mask))
    g_h0 = torch.empty_like(traj[0])
    pg = torch.empty(NP, dtype=torch.float64, device=traj.device)
    with torch.cuda.device(traj.device):
        _lib.check(L.percnn_pi_s1_rollout_bwd_f32(traj.data_ptr(), g_traj.data_ptr(), mask, g_h0.data_ptr(), pg.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), P.data_ptr(), _lib.shape_arg(shape), T,
                                                  _stream()), "s1_rollout_bwd")
    return g_h0, pg


class Stage1RolloutFunction(torch.autograd.Function):
    """T fused steps -> trajectory [T+1,2,H,W] (frame 0 = h0)."""

    @staticmethod
    def forward(ctx, h0, P, steps):
        if h0.dim() != 4 or h0.shape[0] != 1 or h0.shape[1] != 2:
            raise RuntimeError("percnn_amd.stage1: state must be [1,2,H,W] (batch 1, as everywhere in the reference)")
        P = P.contiguous()
        traj = torch.empty((steps + 1,) + tuple(h0.shape[1:]), dtype=torch.float32, device=h0.device)
        traj[0].copy_(h0[0])
        rollout_fwd_(traj, P)
        ctx.save_for_backward(traj, P)
        return traj

    @staticmethod
    def backward(ctx, g_traj):
        traj, P = ctx.saved_tensors
        g_h0, pg = rollout_bwd(traj, g_traj.contiguous(), P)
        return g_h0[None], pg.to(torch.float32), None


def stage1_rollout(h0: torch.Tensor, P: torch.Tensor, steps: int) -> torch.Tensor:
    return Stage1RolloutFunction.apply(h0, P, int(steps))


class Stage1RolloutFramesFunction(torch.autograd.Function):
    """T fused steps returned as the reference returns them -- a tuple of [1,2,H,W] frames (bur1:277-303 appends to a
    Python list) -- as the outputs of ONE autograd node (see functional.PiRolloutFramesFunction for why)."""

    @staticmethod
    def forward(ctx, h0, P, steps, frames):
        if h0.dim() != 4 or h0.shape[0] != 1 or h0.shape[1] != 2:
            raise RuntimeError("percnn_amd.stage1: state must be [1,2,H,W] (batch 1, as everywhere in the reference)")
        P = P.contiguous()
        traj = torch.empty((steps + 1,) + tuple(h0.shape[1:]), dtype=torch.float32, device=h0.device)
        traj[0].copy_(h0[0])
        rollout_fwd_(traj, P)
        ctx.save_for_backward(traj, P)
        ctx.frames = tuple(int(k) for k in frames)
        ctx.set_materialize_grads(False)
        views = traj.unsqueeze(1).unbind(0)                 # all [1,2,*S] frame views in one call
        outs = []
        for k in ctx.frames:                                # functional.Frame: the caller's torch.cat(tuple(output)) is a view
            f = views[k].as_subclass(Frame)
            f._pi_index = k
            outs.append(f)
        return tuple(outs) + (traj.view(traj.shape),)       # last output: the trajectory itself (bur1:607's torch.cat, no copy)

    @staticmethod
    def backward(ctx, *grads):
        traj, P = ctx.saved_tensors
        g_stacked, grads = grads[-1], grads[:-1]
        if g_stacked is None and all(g is None for g in grads):
            return None, None, None, None
        if g_stacked is not None and all(g is None for g in grads):
            g_traj, mask = g_stacked.contiguous(), None
        elif g_stacked is None:
            g_traj, mask = _assemble_frame_grads(grads, ctx.frames, traj)
        else:
            g_traj, mask = g_stacked.clone(), None
            for k, g in zip(ctx.frames, grads):
                if g is not None:
                    g_traj[k].add_(g[0])
        g_h0, pg = rollout_bwd(traj, g_traj, P, frame_mask=mask)
        return g_h0[None], pg.to(torch.float32), None, None


class Stage1RolloutObserveFunction(torch.autograd.Function):
    """Rollout + observation operator ``traj[t_idx][:, :, ::s, ::s]`` in one autograd node (the Stage-1 data loss looks
    at every 5th snapshot and every second grid point, bur1:610); see functional.PiRolloutObserveFunction."""

    @staticmethod
    def forward(ctx, h0, P, steps, t_idx, strides):
        if h0.dim() != 4 or h0.shape[0] != 1 or h0.shape[1] != 2:
            raise RuntimeError("percnn_amd.stage1: state must be [1,2,H,W] (batch 1, as everywhere in the reference)")
        P = P.contiguous()
        traj = torch.empty((steps + 1,) + tuple(h0.shape[1:]), dtype=torch.float32, device=h0.device)
        traj[0].copy_(h0[0])
        rollout_fwd_(traj, P)
        ctx.save_for_backward(traj, P)
        ctx.t_idx = tuple(int(t) % (steps + 1) for t in t_idx)
        ctx.sub = (slice(None),) + tuple(slice(None, None, int(s)) for s in strides)
        pred = _observe(traj, ctx.t_idx, ctx.sub)
        ctx.mark_non_differentiable(traj)
        return pred, traj

    @staticmethod
    def backward(ctx, g_pred, _unused):
        traj, P = ctx.saved_tensors
        g_traj = torch.empty_like(traj)                    # unobserved frames are masked out, never initialised
        mask = _scatter_observed(g_traj, ctx.t_idx, ctx.sub, g_pred)
        g_h0, pg = rollout_bwd(traj, g_traj, P, frame_mask=mask)
        return g_h0[None], pg.to(torch.float32), None, None, None


class Upscaler(nn.Module):
    """IC generator of the Stage-1 scripts (bur1:38-52, lo1:38-51): ConvTranspose2d(2 -> 16, 5, stride 2) - tanh -
    Conv2d(16 -> 2, 1); stock torch.nn, off the hot path.  Registers ``up0`` / ``out`` AND ``convnet`` like the
    reference, so the checkpoint's duplicated keys (UpconvBlock.up0.* and UpconvBlock.convnet.0.*) load."""

    def __init__(self):
        super().__init__()
        self.up0 = nn.ConvTranspose2d(2, 16, kernel_size=5, padding=2, stride=2, output_padding=1, bias=True)
        self.tanh = nn.Tanh()
        self.out = nn.Conv2d(16, 2, 1, 1, padding=0, bias=True)
        self.convnet = nn.Sequential(self.up0, self.tanh, self.out)

    def forward(self, h):
        return self.convnet(h)


class Stage1Cell(nn.Module):
    """Drop-in for the Stage-1 ``RCNNCell`` (bur1:54-187, lo1:53-180).  ``cell(h) -> (ch, ch)``;
    ``cell.rollout(h0, T)`` -> [T+1,2,H,W] runs the reference's T-step loop (bur1:283-303) fused."""
    CONFIG = {"burgers": dict(dx=1 / 100, dt=0.00025, nu_up=0.01), "lo": dict(dx=0.2, dt=0.0125, nu_up=0.2)}

    def __init__(self, family: str = "burgers", input_channels: int = 2, hidden_channels: int = 16,
                 output_channels: int = 2, input_kernel_size: int = 5, input_stride: int = 1, input_padding: int = 2,
                 dx: Optional[float] = None, dt: Optional[float] = None, nu_up: Optional[float] = None):
        super().__init__()
        if (input_channels, hidden_channels, output_channels, input_kernel_size, input_stride) != (2, 16, 2, 5, 1):
            raise ValueError("Stage1Cell: the MFMA kernels are built for 2 -> 16 (5x5) -> 2, stride 1 (the reference's shape)")
        cfg = dict(self.CONFIG[family])
        self.input_channels, self.hidden_channels, self.output_channels = input_channels, hidden_channels, output_channels
        self.input_kernel_size, self.input_stride, self.input_padding = 5, input_stride, input_padding
        self.dx = cfg["dx"] if dx is None else dx
        self.dt = cfg["dt"] if dt is None else dt
        self.nu_up = cfg["nu_up"] if nu_up is None else nu_up
        rs = np.random.RandomState(1234)                                       # bur1:97-99
        self.CA = nn.Parameter(torch.tensor(rs.rand(), dtype=torch.float32))
        self.CB = nn.Parameter(torch.tensor(rs.rand(), dtype=torch.float32))
        lap = torch.zeros(5, 5, dtype=torch.float64)
        lap[2, :] = lap[:, 2] = torch.tensor([-1 / 12, 4 / 3, 0.0, 4 / 3, -1 / 12], dtype=torch.float64)
        lap[2, 2] = -5.0
        self.W_laplace = nn.Conv2d(1, 1, 5, 1, padding=0, bias=False)
        self.W_laplace.weight.data = (lap / self.dx ** 2).to(torch.float32).reshape(1, 1, 5, 5)
        self.W_laplace.weight.requires_grad = False
        for s in "uv":
            for k in (1, 2, 3):
                setattr(self, f"Wh{k}_{s}", nn.Conv2d(2, 16, 5, 1, padding=0, bias=True))
            setattr(self, f"Wh4_{s}", nn.Conv2d(16, 1, 1, 1, padding=0, bias=True))
        for s in "uv":                                                         # init_filter(c=0.5), bur1:130-141
            for k in (1, 2, 3, 4):
                f = getattr(self, f"Wh{k}_{s}")
                bound = 0.5 * float(np.sqrt(1 / np.prod(f.weight.shape[:-1])))
                f.weight.data.uniform_(-bound, bound)
                f.bias.data.fill_(0.0)

    def param_block(self) -> torch.Tensor:
        check_star_stencil(self.W_laplace.weight)
        branch = []
        for s in "uv":
            for k in (1, 2, 3, 4):
                f = getattr(self, f"Wh{k}_{s}")
                branch += [f.weight, f.bias]
        return pack_params(self.dt, self.nu_up * torch.sigmoid(self.CA), self.nu_up * torch.sigmoid(self.CB),
                           self.W_laplace.weight, branch)

    def rollout(self, h0: torch.Tensor, steps: int) -> torch.Tensor:
        return stage1_rollout(h0, self.param_block(), steps)

    def rollout_observe(self, h0: torch.Tensor, steps: int, t_idx: Sequence[int], strides: Sequence[int]):
        """hook used by ``percnn_amd.RCNN.observe``: -> (observed sub-tensor, detached full trajectory)"""
        return Stage1RolloutObserveFunction.apply(h0, self.param_block(), int(steps), tuple(t_idx), tuple(strides))

    def rollout_frames(self, h0: torch.Tensor, steps: int, frames: Sequence[int], with_stacked: bool = False):
        """hook used by ``percnn_amd.RCNN``: the requested frames as outputs of one autograd node (with_stacked: plus, last, the
        [steps+1,2,H,W] trajectory they are views of)"""
        out = Stage1RolloutFramesFunction.apply(h0, self.param_block(), int(steps), tuple(frames))
        return out if with_stacked else out[:-1]

    def forward(self, h: torch.Tensor):
        ch = self.rollout(h, 1)[1:2]
        return ch, ch

    def init_hidden_tensor(self, prev_state):                                  # bur1:181-187
        return prev_state.to(self.CA.device)
