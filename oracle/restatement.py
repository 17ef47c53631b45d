"""CPU restatement of PeRCNN's recurrent Pi-block hot path (TEST INFRASTRUCTURE ONLY).

This file is the *oracle*: a from-scratch, pure-PyTorch (CPU) restatement of the
reference algorithm, with the SAME ATen op sequence as the reference so that its
rounding behaviour is the reference's own.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- never by
the product package ``percnn_amd`` (which fails loudly without its HIP library).

Parity status: PINNED.  ``tools/make_golden.py`` imports the real reference scripts
from ``/root/reference`` in the build container, asserts that this restatement is
bit-identical to them on the same inputs/weights (forward trajectory, loss, all
parameter gradients, dL/dh0) and writes the golden vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the restatement against those vectors.

Reference citations (paths relative to /root/reference):
  2dgs = DataDrivenModeling/2d_gs_rd/train_2drd.py
  3dgs = DataDrivenModeling/3d_gs_rd/train_3drd.py
  lo   = ForwardSimulationOfPDEs/2d_lambda_omega/percnn_LO_eqn.py
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a1: stencil constants  (2dgs:20-24, lo:18-22, 3dgs:22-39)
# --------------------------------------------------------------------------------------
def laplace_stencil(ndim: int) -> np.ndarray:
    """4th-order star Laplacian as a dense [1,1,5,5(,5)] float64 array.

    Per axis taps (-1/12, 4/3, ., 4/3, -1/12); centre -5 in 2D (2dgs:20-24),
    -15/2 in 3D (3dgs:23).
    """
    shape = (1, 1) + (5,) * ndim
    w = np.zeros(shape, dtype=np.float64)
    centre = (0, 0) + (2,) * ndim
    w[centre] = -5.0 if ndim == 2 else -15.0 / 2.0
    for ax in range(ndim):
        for off, val in ((-2, -1 / 12), (-1, 4 / 3), (1, 4 / 3), (2, -1 / 12)):
            idx = list(centre)
            idx[2 + ax] += off
            w[tuple(idx)] = val
    return w


def _conv(ndim):
    return nn.Conv2d if ndim == 2 else nn.Conv3d


def periodic_pad(h: torch.Tensor, ndim: int) -> torch.Tensor:
    """a3: wrap-around halo of width 2 on every spatial axis, last axis first
    (2dgs:108-109, 3dgs:125-127, lo:100-101)."""
    for ax in range(ndim - 1, -1, -1):
        d = 2 + ax
        n = h.shape[d]
        h = torch.cat((h.narrow(d, n - 2, 2), h, h.narrow(d, 0, 2)), dim=d)
    return h


class OracleCell(nn.Module):
    """Restatement of ``RCNNCell`` (2dgs:43-125, 3dgs:58-148, lo:24-121).

    diffusion='sigmoid': coefficient = mu_up*sigmoid(CA|CB)   (2dgs:115-116, 3dgs:133-134)
    diffusion='raw'    : coefficient = DA|DB                   (lo:107-108)
    ``state_dict`` keys equal the reference's.
    """

    def __init__(self, ndim=2, hidden_channels=8, dx=0.01, dt=0.5, mu_up=3.99e-5,
                 diffusion="sigmoid", dtype=torch.float32, scale_form="premul",
                 init="xavier", init_c=0.02):
        super().__init__()
        self.ndim, self.hidden_channels = ndim, hidden_channels
        self.dx, self.dt, self.mu_up, self.diffusion = dx, dt, mu_up, diffusion
        Conv = _conv(ndim)
        if diffusion == "sigmoid":
            # 2dgs:60-62 -- np.random.seed(1234); (rand-0.5)*2 twice
            rs = np.random.RandomState(1234)
            self.CA = nn.Parameter(torch.tensor((rs.rand() - 0.5) * 2, dtype=dtype))
            self.CB = nn.Parameter(torch.tensor((rs.rand() - 0.5) * 2, dtype=dtype))
        else:
            # lo:42-43
            self.DA = nn.Parameter(torch.tensor(0.2, dtype=dtype))
            self.DB = nn.Parameter(torch.tensor(0.2, dtype=dtype))
        self.W_laplace = Conv(1, 1, 5, 1, padding=0, bias=False, dtype=dtype)
        st = torch.tensor(laplace_stencil(ndim), dtype=dtype)
        if scale_form == "premul":      # 2dgs:66, 3dgs:81: 1/dx**2 * tensor
            self.W_laplace.weight.data = 1 / dx ** 2 * st
        else:                           # lo:49: tensor / dx**2
            self.W_laplace.weight.data = st / dx ** 2
        self.W_laplace.weight.requires_grad = False
        for s in ("u", "v"):
            for k in (1, 2, 3):
                setattr(self, f"Wh{k}_{s}", Conv(2, hidden_channels, 1, 1, padding=0, bias=True, dtype=dtype))
            setattr(self, f"Wh4_{s}", Conv(hidden_channels, 1, 1, 1, padding=0, bias=True, dtype=dtype))
        self.filter_list = [getattr(self, f"Wh{k}_{s}") for s in ("u", "v") for k in (1, 2, 3, 4)]
        for f in self.filter_list:
            if init == "xavier":        # 2dgs:92-103 (c=0.02), 3dgs:109-120 (c=0.01)
                nn.init.xavier_uniform_(f.weight)
                f.weight.data = init_c * f.weight.data
            else:                       # lo:86-95 (c=0.5)
                b = init_c * np.sqrt(1 / np.prod(f.weight.shape[:-1]))
                f.weight.data.uniform_(-b, b)
            f.bias.data.fill_(0.0)

    def coefficients(self):
        if self.diffusion == "sigmoid":
            return self.mu_up * torch.sigmoid(self.CA), self.mu_up * torch.sigmoid(self.CB)
        return self.DA, self.DB

    def forward(self, h):
        # 2dgs:105-121 / 3dgs:123-139 / lo:98-112 -- identical op order.
        h_pad = periodic_pad(h, self.ndim)
        u_pad, v_pad = h_pad[:, 0:1, ...], h_pad[:, 1:2, ...]
        u_prev, v_prev = h[:, 0:1, ...], h[:, 1:2, ...]
        cu, cv = self.coefficients()
        u_res = cu * self.W_laplace(u_pad) + self.Wh4_u(self.Wh1_u(h) * self.Wh2_u(h) * self.Wh3_u(h))
        v_res = cv * self.W_laplace(v_pad) + self.Wh4_v(self.Wh1_v(h) * self.Wh2_v(h) * self.Wh3_v(h))
        u_next = u_prev + u_res * self.dt
        v_next = v_prev + v_res * self.dt
        ch = torch.cat((u_next, v_next), dim=1)
        return ch, ch


def gs2d_cell(hidden_channels=8):
    """2D Gray-Scott cell constants (2dgs:56-58)."""
    return OracleCell(2, hidden_channels, dx=0.01, dt=0.5, mu_up=3.99e-5, diffusion="sigmoid",
                      dtype=torch.float32, scale_form="premul", init="xavier", init_c=0.02)


def gs3d_cell(hidden_channels=2):
    """3D Gray-Scott cell constants (3dgs:71-73)."""
    return OracleCell(3, hidden_channels, dx=100 / 48, dt=0.5, mu_up=0.274, diffusion="sigmoid",
                      dtype=torch.float32, scale_form="premul", init="xavier", init_c=0.01)


def lo2d_cell(hidden_channels=4):
    """2D lambda-omega cell constants, float64 (lo:12, lo:38-43)."""
    return OracleCell(2, hidden_channels, dx=0.2, dt=0.0125, mu_up=None, diffusion="raw",
                      dtype=torch.float64, scale_form="div", init="uniform", init_c=0.5)


class OracleUpscaler(nn.Module):
    """IC generator (2dgs:26-41, 3dgs:41-56). Stock torch.nn; off the hot path."""

    def __init__(self, ndim=2):
        super().__init__()
        if ndim == 2:
            layers = [nn.ConvTranspose2d(2, 8, 5, padding=2, stride=2, output_padding=1, bias=True),
                      nn.Sigmoid(),
                      nn.ConvTranspose2d(8, 8, 5, padding=2, stride=2, output_padding=1, bias=True),
                      nn.Conv2d(8, 2, 1, 1, padding=0, bias=True)]
        else:
            layers = [nn.ConvTranspose3d(2, 8, 5, padding=2, stride=2, output_padding=1, bias=True),
                      nn.Sigmoid(),
                      nn.ConvTranspose3d(8, 8, 5, padding=2, stride=1, output_padding=0, bias=True),
                      nn.Conv3d(8, 2, 1, 1, padding=0, bias=True)]
        self.convnet = nn.Sequential(*layers)

    def forward(self, h):
        return self.convnet(h)


class OracleRCNN(nn.Module):
    """Restatement of ``RCNN`` (2dgs:128-190, 3dgs:151-214, lo:124-218).

    Either ``init_state`` (lo:158: a fixed tensor) or ``upscaler`` + ``init_state_low``
    (2dgs:150,164).  ``cell_name`` is 'crnn_cell' (2dgs:152) or 'rcnn_cell' (lo:160).
    """

    def __init__(self, cell, step=1, effective_step=(1,), init_state=None, upscaler=None,
                 init_state_low=None, cell_name="crnn_cell"):
        super().__init__()
        self.step, self.effective_step = step, list(effective_step)
        self.cell_name = cell_name
        self.init_state = init_state
        self.init_state_low = init_state_low
        if upscaler is not None:                # registered before the cell (2dgs:150-158)
            self.UpconvBlock = upscaler
        setattr(self, cell_name, cell)

    def forward(self):
        if hasattr(self, "UpconvBlock"):
            self.init_state = self.UpconvBlock(self.init_state_low)      # 2dgs:164
        cell = getattr(self, self.cell_name)
        outputs = [self.init_state]
        second_last_state = []
        h = self.init_state
        for step in range(self.step):                                   # 2dgs:169
            h, o = cell(h)
            if step == (self.step - 2):                                 # 2dgs:182-184
                second_last_state = h.clone()
            if step in self.effective_step:                             # 2dgs:187
                outputs.append(o)
        return outputs, second_last_state


# --------------------------------------------------------------------------------------
# Synthetic inputs of SURVEY 8(d) (no datasets ship with the reference)
# --------------------------------------------------------------------------------------
def gs_initial_state(shape, seed=0, dtype=torch.float32):
    """u=1, v=0; centred square/cube (half-width N/10 in 2D, N/8 in 3D) set to u=.5, v=.25;
    plus 0.01*randn from manual_seed(seed)."""
    ndim = len(shape)
    h = torch.zeros((1, 2) + tuple(shape), dtype=dtype)
    h[:, 0] = 1.0
    sl = []
    for n in shape:
        hw = max(1, n // 10 if ndim == 2 else n // 8)
        sl.append(slice(n // 2 - hw, n // 2 + hw))
    h[(slice(None), 0) + tuple(sl)] = 0.5
    h[(slice(None), 1) + tuple(sl)] = 0.25
    g = torch.Generator().manual_seed(seed)
    h = h + 0.01 * torch.randn(h.shape, generator=g, dtype=dtype)
    return h


def lo_initial_state(n, dtype=torch.float64):
    """lambda-omega spiral: x=(i-N/2)*0.2; u=tanh(R)cos(theta-R), v=tanh(R)sin(theta-R)."""
    x = (torch.arange(n, dtype=torch.float64) - n / 2) * 0.2
    yy, xx = torch.meshgrid(x, x, indexing="ij")
    r = torch.sqrt(xx ** 2 + yy ** 2)
    th = torch.atan2(yy, xx)
    u = torch.tanh(r) * torch.cos(th - r)
    v = torch.tanh(r) * torch.sin(th - r)
    return torch.stack((u, v))[None].to(dtype)


def rollout(cell, h0, steps):
    """Dense trajectory [steps+1, 2, *S] exactly as callers build it (2dgs:394)."""
    model = OracleRCNN(cell, step=steps, effective_step=list(range(steps)), init_state=h0)
    outs, _ = model()
    return torch.cat(tuple(outs), dim=0)


# --------------------------------------------------------------------------------------
# physics-residual loss -- restatement of loss_generator.get_phy_Loss + loss_gen
# (2dgs:270-353, 3dgs:287-346, lo:283-357): same padding (2 low / 3 high), same dense Laplacian conv
# divided by dx^2, same staggered time difference, same analytic right-hand sides.
# --------------------------------------------------------------------------------------
def physics_loss_reference(output: torch.Tensor, family: str, dx: float, dt: float) -> torch.Tensor:
    ndim = output.dim() - 2
    for ax in range(ndim - 1, -1, -1):                       # 2dgs:345-346 / 3dgs:337-339
        d = 2 + ax
        n = output.shape[d]
        output = torch.cat((output.narrow(d, n - 2, 2), output, output.narrow(d, 0, 3)), dim=d)
    W = torch.tensor(laplace_stencil(ndim), dtype=output.dtype)
    conv = F.conv2d if ndim == 2 else F.conv3d
    inner = (slice(2, -2),) * ndim
    lap_u = conv(output[0:-2, 0:1], W) / (dx ** 2)           # Conv2dDerivative: filter(input) / deno (2dgs:214-216)
    lap_v = conv(output[0:-2, 1:2], W) / (dx ** 2)
    uu = output[(slice(None), slice(0, 1)) + inner]
    vv = output[(slice(None), slice(1, 2)) + inner]
    u_t = (uu[1:-1] - uu[:-2]) / dt                           # Conv1d with [-1, 1, 0] then / deno (2dgs:265-268)
    v_t = (vv[1:-1] - vv[:-2]) / dt
    u, v = uu[0:-2], vv[0:-2]
    if family == "gs2d":                                      # 2dgs:321-327
        Du = 2e-5; Dv = Du / 4; f = 1 / 25; k = 3 / 50
        f_u = (Du * lap_u - u * (v ** 2) + f * (1 - u) - u_t) / 1
        f_v = (Dv * lap_v + u * (v ** 2) - (f + k) * v - v_t) / 1
    elif family == "gs3d":                                    # 3dgs:316-322
        Du = 0.2; Dv = 0.1; f = 0.025; k = 0.055
        f_u = (Du * lap_u - u * v ** 2 + f * (1 - u) - u_t)
        f_v = (Dv * lap_v + u * v ** 2 - (f + k) * v - v_t)
    else:                                                     # lo:339-340
        f_u = 0.1 * lap_u + (1 - u ** 2 - v ** 2) * u + (u ** 2 + v ** 2) * v - u_t
        f_v = 0.1 * lap_v - (u ** 2 + v ** 2) * u + (1 - u ** 2 - v ** 2) * v - v_t
    mse = torch.nn.MSELoss()
    return mse(f_u, torch.zeros_like(f_u)) + mse(f_v, torch.zeros_like(f_v))


# --------------------------------------------------------------------------------------
# Stage-3 physics-based cell, lambda-omega.  Restates
# DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-3/fine_tuning_LO_[10%noise,41x51x51].py  (tag lo3):
#   Conv2dDerivative (circular 5x5 conv, result / resol)  lo3:56-82
#   RCNNCell.__init__ (13 scalar coefficients, dx=dy=0.2, dt=0.0125)  lo3:83-147
#   f_rhs  lo3:149-152,  Euler forward  lo3:203-216
# --------------------------------------------------------------------------------------
class OracleStage3LOCell(nn.Module):
    INIT = dict(nu_u=0.09465, nu_v=0.09455, C1_u=1.0081, C2_u=-1.0167, C3_u=0.9973, C4_u=-1.0176, C5_u=0.9981,
                C1_v=0.9873, C2_v=-0.9987, C3_v=-0.9945, C4_v=-0.9985, C5_v=-0.9928, C6_v=0.0065)

    def __init__(self):
        super().__init__()
        for k, v in self.INIT.items():
            setattr(self, k, nn.Parameter(torch.tensor(v, dtype=torch.float64)))
        self.dx, self.dy, self.dt = 0.2, 0.2, 0.0125

        class _Lap(nn.Module):
            def __init__(self, resol):
                super().__init__()
                self.resol = resol
                self.filter = nn.Conv2d(1, 1, 5, 1, padding=2, padding_mode="circular", bias=False)
                self.filter.weight.data = torch.tensor(laplace_stencil(2), dtype=torch.float64)
                self.filter.weight.requires_grad = False

            def forward(self, x):
                return self.filter(x) / self.resol

        self.laplace_op = _Lap(self.dx ** 2)

    def f_rhs(self, u, v):
        f_u = self.nu_u*self.laplace_op(u) + self.C1_u*u + self.C2_u*u**3 + self.C3_u*u**2*v + self.C4_u*u*v**2 + self.C5_u*v**3
        f_v = self.nu_v*self.laplace_op(v) + self.C1_v*v + self.C2_v*u**3 + self.C3_v*u**2*v + self.C4_v*u*v**2 + self.C5_v*v**3 + self.C6_v*u
        return f_u, f_v

    def forward(self, h):
        u0, v0 = h[:, 0:1, ...], h[:, 1:2, ...]
        f_u, f_v = self.f_rhs(u0, v0)
        u_next = u0 + self.dt * f_u
        v_next = v0 + self.dt * f_v
        ch = torch.cat((u_next, v_next), dim=1)
        return ch, ch

    def forward_rk4(self, h):
        """lo3:154-201: classical RK4 with the same f_rhs (defined in the reference, never called there)."""
        u0, v0 = h[:, 0:1, ...], h[:, 1:2, ...]
        k1_u, k1_v = self.f_rhs(u0, v0)
        u1, v1 = u0 + k1_u * self.dt / 2.0, v0 + k1_v * self.dt / 2.0
        k2_u, k2_v = self.f_rhs(u1, v1)
        u2, v2 = u0 + k2_u * self.dt / 2.0, v0 + k2_v * self.dt / 2.0
        k3_u, k3_v = self.f_rhs(u2, v2)
        u3, v3 = u0 + k3_u * self.dt, v0 + k3_v * self.dt
        k4_u, k4_v = self.f_rhs(u3, v3)
        u_next = u0 + self.dt * (k1_u + 2 * k2_u + 2 * k3_u + k4_u) / 6.0
        v_next = v0 + self.dt * (k1_v + 2 * k2_v + 2 * k3_v + k4_v) / 6.0
        ch = torch.cat((u_next, v_next), dim=1)
        return ch, ch


# --------------------------------------------------------------------------------------
# Stage-3 physics-based cell, 2D Burgers.  Restates
# DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-3/fine_tuning_[5%noise,41x51x51].py (tag bur3):
#   dx_2d_op / dy_2d_op / lap_2d_op  bur3:20-36;  Conv2dDerivative (circular conv, / resol)  bur3:56-82
#   RCNNCell.__init__  bur3:84-152 (6 scalars, dx=dy=1/100, dt=0.00025);  f_rhs  bur3:154-157;  forward  bur3:209-221
# --------------------------------------------------------------------------------------
def first_derivative_stencils():
    dx_op = np.zeros((1, 1, 5, 5)); dy_op = np.zeros((1, 1, 5, 5))
    for i, val in zip((0, 1, 3, 4), (1 / 12, -8 / 12, 8 / 12, -1 / 12)):
        dx_op[0, 0, i, 2] = val          # differentiates along tensor dim 2 (rows)   bur3:20-24
        dy_op[0, 0, 2, i] = val          # along dim 3 (columns)                      bur3:26-30
    return dx_op, dy_op


class OracleStage3BurgersCell(nn.Module):
    INIT = dict(nu_u=0.0050078, nu_v=0.0050228, C1_u=-0.982252, C2_u=-0.992132, C1_v=-0.983758, C2_v=-0.971269)

    class _Der(nn.Module):
        def __init__(self, stencil, resol):
            super().__init__()
            self.resol = resol
            self.filter = nn.Conv2d(1, 1, 5, 1, padding=2, padding_mode="circular", bias=False)
            self.filter.weight.data = torch.tensor(stencil, dtype=torch.float64)
            self.filter.weight.requires_grad = False

        def forward(self, x):
            return self.filter(x) / self.resol

    def __init__(self):
        super().__init__()
        for k, v in self.INIT.items():
            setattr(self, k, nn.Parameter(torch.tensor(v, dtype=torch.float64)))
        self.dx, self.dy, self.dt = 1 / 100, 1 / 100, 0.00025
        dxo, dyo = first_derivative_stencils()
        self.laplace_op = self._Der(laplace_stencil(2), self.dx ** 2)
        self.dx_op = self._Der(dxo, self.dx)
        self.dy_op = self._Der(dyo, self.dy)

    def f_rhs(self, u, v):
        f_u = self.nu_u*self.laplace_op(u) + self.C1_u*u*self.dx_op(u) + self.C2_u*v*self.dy_op(u)
        f_v = self.nu_v*self.laplace_op(v) + self.C1_v*u*self.dx_op(v) + self.C2_v*v*self.dy_op(v)
        return f_u, f_v

    def forward(self, h):
        u0, v0 = h[:, 0:1, ...], h[:, 1:2, ...]
        f_u, f_v = self.f_rhs(u0, v0)
        ch = torch.cat((u0 + self.dt * f_u, v0 + self.dt * f_v), dim=1)
        return ch, ch

    def forward_rk4(self, h):
        """bur3:159-206: classical RK4 with the same f_rhs (defined in the reference, never called there)."""
        u0, v0 = h[:, 0:1, ...], h[:, 1:2, ...]
        k1_u, k1_v = self.f_rhs(u0, v0)
        u1, v1 = u0 + k1_u * self.dt / 2.0, v0 + k1_v * self.dt / 2.0
        k2_u, k2_v = self.f_rhs(u1, v1)
        u2, v2 = u0 + k2_u * self.dt / 2.0, v0 + k2_v * self.dt / 2.0
        k3_u, k3_v = self.f_rhs(u2, v2)
        u3, v3 = u0 + k3_u * self.dt, v0 + k3_v * self.dt
        k4_u, k4_v = self.f_rhs(u3, v3)
        u_next = u0 + self.dt * (k1_u + 2 * k2_u + 2 * k3_u + k4_u) / 6.0
        v_next = v0 + self.dt * (k1_v + 2 * k2_v + 2 * k3_v + k4_v) / 6.0
        ch = torch.cat((u_next, v_next), dim=1)
        return ch, ch


# ---------------------------------------------------------------------------------------------
# Stage-1 Pi-block of the equation-discovery pipeline: the three parallel branches
# are 5x5 convolutions 2 -> 16 (periodic), their Hadamard product is contracted by a 1x1 conv 16 -> 1.
#   Burgers   DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/rcnn_Burgers_[...].py:54-178
#             (explicit cat padding :160-163, convs with padding=0 :103-124, update :172-176)
#   lambda-omega  .../2D_Lambda_Omega_eqn/stage-1/rcnn_LO_[...].py:53-172
#             (padding_mode='circular' :102-124, update :165-168)
# Both are float32; the state_dict schema (CA, CB, W_laplace.weight, Wh{1..4}_{u,v}.{weight,bias}) is shared.
# ---------------------------------------------------------------------------------------------
class OracleStage1Cell(nn.Module):
    CONFIG = {"burgers": dict(dx=1 / 100, dt=0.00025, nu_up=0.01),
              "lo": dict(dx=0.2, dt=0.0125, nu_up=0.2)}

    def __init__(self, family="burgers", hidden_channels=16, dtype=torch.float32):
        super().__init__()
        cfg = self.CONFIG[family]
        self.family, self.hidden_channels = family, hidden_channels
        self.dx, self.dt, self.nu_up = cfg["dx"], cfg["dt"], cfg["nu_up"]
        rs = np.random.RandomState(1234)                                   # bur1:97-99
        self.CA = nn.Parameter(torch.tensor(rs.rand(), dtype=dtype))
        self.CB = nn.Parameter(torch.tensor(rs.rand(), dtype=dtype))
        # the two reference scripts pad differently (same values, different autograd graph => the backward
        # accumulates in a different order); mirror each so gradients are bit-identical too
        pad = dict(padding=0) if family == "burgers" else dict(padding=2, padding_mode="circular")
        self.W_laplace = nn.Conv2d(1, 1, 5, 1, bias=False, **pad)
        self.W_laplace.weight.data = (1 / self.dx ** 2 * torch.tensor(laplace_stencil(2))).to(dtype).reshape(1, 1, 5, 5)
        self.W_laplace.weight.requires_grad = False
        for s in "uv":
            for k in (1, 2, 3):
                setattr(self, f"Wh{k}_{s}", nn.Conv2d(2, hidden_channels, 5, 1, bias=True, dtype=dtype, **pad))
            setattr(self, f"Wh4_{s}", nn.Conv2d(hidden_channels, 1, 1, 1, padding=0, bias=True, dtype=dtype))
        for s in "uv":                                                     # bur1:130-141 (c = 0.5)
            for k in (1, 2, 3, 4):
                f = getattr(self, f"Wh{k}_{s}")
                bound = 0.5 * np.sqrt(1 / np.prod(f.weight.shape[:-1]))
                f.weight.data.uniform_(-bound, bound)
                f.bias.data.fill_(0.0)

    def forward(self, h):
        if self.family == "burgers":                                       # bur1:160-166
            hp = torch.cat((h[:, :, :, -2:], h, h[:, :, :, 0:2]), dim=3)
            hp = torch.cat((hp[:, :, -2:, :], hp, hp[:, :, 0:2, :]), dim=2)
            lap_in = (hp[:, 0:1, ...], hp[:, 1:2, ...])
            u_prev, v_prev = h[:, 0:1, ...], h[:, 1:2, ...]
        else:                                                              # lo1:161-166 (one slice feeds both uses)
            hp = h
            u_prev, v_prev = h[:, 0:1, ...], h[:, 1:2, ...]
            lap_in = (u_prev, v_prev)
        u_res = self.nu_up * torch.sigmoid(self.CA) * self.W_laplace(lap_in[0]) + self.Wh4_u(self.Wh1_u(hp) * self.Wh2_u(hp) * self.Wh3_u(hp))
        v_res = self.nu_up * torch.sigmoid(self.CB) * self.W_laplace(lap_in[1]) + self.Wh4_v(self.Wh1_v(hp) * self.Wh2_v(hp) * self.Wh3_v(hp))
        u_next = u_prev + u_res * self.dt
        v_next = v_prev + v_res * self.dt
        ch = torch.cat((u_next, v_next), dim=1)
        return ch, ch
