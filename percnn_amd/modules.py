"""Drop-in ``RCNNCell`` / ``RCNN`` modules backed by the HIP Pi-block kernels.

Same parameter names, shapes and dtypes as the reference modules, so the shipped checkpoints
load with ``load_state_dict`` (keys ``CA, CB | DA, DB, W_laplace.weight, Wh{1..4}_{u,v}.{weight,
bias}`` under prefix ``crnn_cell.`` -- DataDrivenModeling/2d_gs_rd/train_2drd.py:46-90,152-158;
``rcnn_cell.`` in ForwardSimulationOfPDEs/2d_lambda_omega/percnn_LO_eqn.py:160).  Unlike the
reference, the per-PDE constants are real constructor arguments instead of source literals.
"""
from __future__ import annotations

from typing import Optional, Sequence

import os
import numpy as np
import torch
import torch.nn as nn

from . import functional as F_pi


def laplace_stencil(ndim: int) -> np.ndarray:
    """4th-order star Laplacian, dense [1,1,5,5(,5)] (train_2drd.py:20-24; train_3drd.py:22-39)."""
    w = np.zeros((1, 1) + (5,) * ndim)
    c = (0, 0) + (2,) * ndim
    w[c] = -5.0 if ndim == 2 else -7.5
    for a in range(ndim):
        for off, val in ((-2, -1 / 12), (-1, 4 / 3), (1, 4 / 3), (2, -1 / 12)):
            i = list(c)
            i[2 + a] += off
            w[tuple(i)] = val
    return w


_ext_probe = [False, None]


def _native_ext(required: bool = False):
    """The operator library's module (csrc/torch_ext.cpp), or None while the package has not been built (CPU-side tools)."""
    if _ext_probe[1] is not None:
        return _ext_probe[1]
    if required or not _ext_probe[0]:
        import os
        from . import _lib
        _ext_probe[0] = True
        if required or (os.path.exists(_lib.TORCH_EXT_PATH) and os.path.exists(_lib.LIB_PATH)):
            from . import ops
            ops.load_native()
            _ext_probe[1] = _lib.torch_ext()
    return _ext_probe[1]


class RCNNCell(nn.Module):
    """Pi-block cell: ``forward(h[1,2,*S]) -> (h_next, h_next)`` (train_2drd.py:105-121).

    diffusion='sigmoid' -> coefficient mu_up*sigmoid(CA|CB) (train_2drd.py:115-116);
    diffusion='raw'     -> coefficient DA|DB                 (percnn_LO_eqn.py:107-108).

    reaction='factored' evaluates the six 1x1 branches, their Hadamard product and the 1x1 aggregation per point in
    the reference's operation order (train_2drd.py:115-116).  reaction='poly' (default) evaluates the SAME cubic from
    its 2 x 10 monomial coefficients c_m (contracted in float64 on the device, Horner form) -- ~4x fewer operations at
    Hc = 8, but a different rounding: monomials that cancel each other (e.g. (u - a)^3 with |a| >> |u - a|) cancel
    exactly in the factored form and only to rounding in the expanded one.

    WHEN 'factored' IS REQUIRED.  The per-step rounding noise of the expanded form, relative to the state, is
    eps * A with the amplification

        A = dt * max_s sum_m |c_m^s| * phi_m(|u|max, |v|max) / max(|u|max, |v|max)        (``poly_amplification``)

    Measured (tests/test_host_logic.py::test_poly_conditioning_rule, stable cubic with roots at a = 0 ... 50, 100
    steps): state rel-L2 of 'poly' vs a float64 run = 0.6 * eps * A (A = 170 -> 6e-6, A = 4000 -> 1.3e-4) while
    'factored' stays at the 1.6e-7 float32 noise floor for every A.  Rule: keep 'poly' while A <= 10 in float32
    (<= 6e-7 extra, inside the reference's own fp32-vs-fp64 spread of 4e-7 over 1000 steps); use
    reaction='factored' beyond -- always for A >= 100.  float64: A <= 1e4.  The shipped checkpoints sit at
    A = 1.04 (2D GS), 0.39 (3D GS), 0.06 (lambda-omega) for states in [0, 1].
    """

    def __init__(self, ndim: int = 2, hidden_channels: int = 8, dx: float = 0.01, dt: float = 0.5,
                 mu_up: Optional[float] = 3.99e-5, diffusion: str = "sigmoid", dtype=torch.float32,
                 stencil_scale: str = "premul", init: str = "xavier", init_c: float = 0.02,
                 reaction: str = "poly"):
        super().__init__()
        if reaction not in ("poly", "factored"):
            raise ValueError("reaction must be 'poly' or 'factored'")
        # 'factored': the kernels evaluate the six 1x1 branches, their Hadamard product and the 1x1
        #             aggregation per point, in the reference's operation order.
        # 'poly'    : the same cubic, pre-contracted on the device to 2 x 10 monomial coefficients
        #             (functional.contract_block) -- ~4x fewer VALU operations per point at Hc = 8,
        #             cost independent of Hc; same function, different rounding (parity-tested).
        self.reaction = reaction
        if ndim not in (2, 3):
            raise ValueError("ndim must be 2 or 3")
        if diffusion not in ("sigmoid", "raw"):
            raise ValueError("diffusion must be 'sigmoid' or 'raw'")
        self.ndim, self.hidden_channels = ndim, hidden_channels
        self.input_channels = 2
        self.dx, self.dt, self.mu_up, self.diffusion = dx, dt, mu_up, diffusion
        Conv = nn.Conv2d if ndim == 2 else nn.Conv3d
        if diffusion == "sigmoid":
            rs = np.random.RandomState(1234)                      # train_2drd.py:60-62
            self.CA = nn.Parameter(torch.tensor((rs.rand() - 0.5) * 2, dtype=dtype))
            self.CB = nn.Parameter(torch.tensor((rs.rand() - 0.5) * 2, dtype=dtype))
        else:
            self.DA = nn.Parameter(torch.tensor(0.2, dtype=dtype))  # percnn_LO_eqn.py:42-43
            self.DB = nn.Parameter(torch.tensor(0.2, dtype=dtype))
        # frozen but serialised (train_2drd.py:65-67); the kernels read THESE values tap by tap
        self.W_laplace = Conv(1, 1, 5, 1, padding=0, bias=False, dtype=dtype)
        st = torch.tensor(laplace_stencil(ndim), dtype=dtype)
        self.W_laplace.weight.data = (1 / dx ** 2 * st) if stencil_scale == "premul" else (st / dx ** 2)
        self.W_laplace.weight.requires_grad = False
        for s in ("u", "v"):
            for k in (1, 2, 3):
                setattr(self, f"Wh{k}_{s}", Conv(2, hidden_channels, 1, 1, padding=0, bias=True, dtype=dtype))
            setattr(self, f"Wh4_{s}", Conv(hidden_channels, 1, 1, 1, padding=0, bias=True, dtype=dtype))
        self.filter_list = [getattr(self, f"Wh{k}_{s}") for s in ("u", "v") for k in (1, 2, 3, 4)]
        self.init_filter(self.filter_list, init_c, init)
        self._stencil_checked_version = None
        self._dt_cache = None
        self._block_cache = None            # (key, packed block) of the last param_block() call, see _block_key
        # conditioning guard of reaction='poly' (class docstring): on a HIP device the pack launch prices the expanded cubic
        # (amplification A for states inside `state_bound`) and the cell packs the FACTORED block -- the reference's own
        # operation order -- while A is above `poly_guard_max` (None: 10 in float32, 1e4 in float64)
        self.poly_guard = True
        self.state_bound = (1.0, 1.0)       # |u|, |v| the amplification is priced at (Gray-Scott / lambda-omega states)
        self.poly_guard_max = None
        self._guard = None
        self._guard_warned = False
        # A step loop that hands every output back as the next input (`for step in range(T): h, _ = cell(h)`, train_2drd.py:169-188)
        # gets its next 4 / 8 / 16 states from ONE fused launch (csrc/torch_ext.cpp, BlockState::step): bit-identical, and only
        # when the input IS the previous output, unmodified.  False: every call is a single-step launch.
        self.speculate = True

    # caches and device-side handles are not state: copy.deepcopy(cell) / pickling a whole model start without them
    _TRANSIENT = ("_block_cache", "_block_acc", "_dt_cache")

    def __getstate__(self):
        return {k: (None if k in self._TRANSIENT else v) for k, v in self.__dict__.items() if k != "_pack_src_list"}

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def init_filter(self, filter_list, c, mode="xavier"):
        for f in filter_list:
            if mode == "xavier":                                   # train_2drd.py:92-103
                nn.init.xavier_uniform_(f.weight)
                f.weight.data = c * f.weight.data
            else:                                                  # percnn_LO_eqn.py:86-95
                b = c * np.sqrt(1 / np.prod(f.weight.shape[:-1]))
                f.weight.data.uniform_(-b, b)
            if f.bias is not None:
                f.bias.data.fill_(0.0)
        self.invalidate_cache()            # `.data` edits do not move the version counters the cache is keyed on

    # -- parameter block ----------------------------------------------------------------------
    def coefficients(self):
        if self.diffusion == "sigmoid":
            return self.mu_up * torch.sigmoid(self.CA), self.mu_up * torch.sigmoid(self.CB)
        return self.DA, self.DB

    def _validate_stencil(self):
        w = self.W_laplace.weight
        key = (w._version, w.data_ptr())
        if self._stencil_checked_version != key:
            F_pi.check_star_stencil(w)
            self._stencil_checked_version = key

    def _pack_src(self):
        """Where the 19 tensors of the block live, in packing order: (dict, sub-module name or None, parameter name)."""
        src = self.__dict__.get("_pack_src_list")
        if src is None:
            names = ("CA", "CB") if self.diffusion == "sigmoid" else ("DA", "DB")
            src = [(self._parameters, None, names[0]), (self._parameters, None, names[1]), (self._modules, "W_laplace", "weight")]
            for s in ("u", "v"):
                for k in (1, 2, 3, 4):
                    src += [(self._modules, f"Wh{k}_{s}", "weight"), (self._modules, f"Wh{k}_{s}", "bias")]
            self.__dict__["_pack_src_list"] = src
        return src

    def _pack_tensors(self):
        """The 19 tensors of the block in packing order (read afresh from the module tree: only a cache MISS asks)."""
        return [(d if m is None else d[m]._parameters)[k] for d, m, k in self._pack_src()]

    def invalidate_cache(self) -> None:
        """Drop the cached parameter block.  REQUIRED after editing a parameter through ``.data`` (``p.data.mul_()``,
        ``.data.uniform_()``, ``.data.fill_()``, ...): such edits do not move the version counters the cache is keyed on
        (``init_filter`` calls this itself).  Ordinary updates -- ``optimizer.step()``, ``load_state_dict``, ``p.copy_()`` under
        ``no_grad``, ``.to()``, replacing a Parameter or its ``.data`` -- are seen without it."""
        self._drop_block()

    def _drop_block(self) -> None:
        """Forget the cached block AND empty its native state (block, speculated frames): the state is reachable from autograd
        nodes' frames, so whatever it still held would keep an iteration's graph alive."""
        acc = self.__dict__.get("_block_acc")
        self.__dict__["_block_cache"] = None
        self.__dict__["_block_acc"] = None
        if acc is not None:
            ext = _native_ext()
            if ext is not None:
                ext.release_block(acc)

    def _block_key(self):
        """What a packed block depends on: identity, version counter AND storage address of every parameter (optimizer.step(),
        load_state_dict, .to(), parameter surgery, replaced sub-modules, `p.data = new` on any of the 19 tensors -- ADVICE r3),
        dt (the reference reads self.dt every step, train_2drd.py:117), the reaction mode, the guard's settings and whether
        autograd records.  The tensor part is ONE call into the operator library (this runs once per time step of a
        reference-style loop; ~60 Python-level lookups otherwise)."""
        ext = _native_ext()
        if ext is not None:
            self._pack_src()
            k = ext.cell_key(self)                             # the same function fast_forward validates the cache with
            if k >= 0:
                return k
        ts = self._pack_tensors()
        return ((tuple(id(t) for t in ts), tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts)), float(self.dt),
                self.reaction, self.diffusion, torch.is_grad_enabled(), self.poly_guard, tuple(self.state_bound),
                self.poly_guard_max, self.speculate)

    def param_block(self, fresh: bool = False) -> torch.Tensor:
        """The packed parameter block the kernels read.  A caller that keeps the reference's own step loop
        (``for step in range(T): h, _ = cell(h)``, train_2drd.py:169-188) calls this once per time step; the block is
        therefore cached until something it depends on changes (_block_key) or until a backward pass has run through it
        (the gradient of that pass has been delivered: the next iteration gets a fresh block): one pack launch and ONE
        pack-backward per training iteration instead of T, the per-step gradients accumulate on the shared block.
        ``fresh=True`` (what ``RCNN``'s rollout entry points pass: one pack launch per ROLLOUT is free) never returns a cached
        block, so a rollout sees ``.data`` edits even without ``invalidate_cache()``."""
        if not torch.compiler.is_compiling():
            key = self._block_key()
            hit = self._block_cache
            if hit is not None and hit[0] == key and not fresh:
                return hit[1]
            tensors = self._pack_tensors()
            # the block's native state (csrc/torch_ext.cpp: BlockState): the rows a reference-style step loop's per-step nodes
            # leave their parameter-gradient sums in (delivered once per backward pass by the pack node) and the speculative
            # multi-step forward of such a loop
            acc = None
            w = tensors[2]
            if w.is_cuda:
                ext = _native_ext()
                if ext is not None:
                    acc = ext.new_block_state(w, max(F_pi.NPOLY, F_pi.param_count(self.hidden_channels)))
                    acc.speculate = bool(self.speculate)
            old = self.__dict__.get("_block_acc")
            if old is not None:
                self._drop_block()                           # the replaced block's state lets go of its frames / block
            self.__dict__["_acc_delivered"] = False
            P = self._param_block_uncached(acc)
            if acc is not None and P.requires_grad and not self.__dict__.get("_acc_delivered"):
                # the block came from the tensor-op assembly (a trainable stencil, a non-contiguous / mixed-dtype parameter):
                # no node would ever deliver the sums per-step nodes leave in the state -> those steps return the block's
                # gradient themselves (F_pi.pi_step), the state is not used (gradients were silently lost)
                acc = None
            self.__dict__["_block_acc"] = acc
            if acc is not None and isinstance(key, int):
                ext.bind_block(acc, P, key)                  # forward()'s one-call hit path (csrc/torch_ext.cpp: fast_forward)
            if P.requires_grad:
                import weakref
                me = weakref.ref(self)

                def consumed(_g, me=me, P_id=id(P)):          # a backward pass reached the block: next iteration, new block
                    cell = me()
                    if cell is not None and cell._block_cache is not None and id(cell._block_cache[1]) == P_id:
                        cell._drop_block()
                P.register_hook(consumed)
            self._block_cache = (key, P)
            return P
        return self._param_block_uncached()

    # -- conditioning guard of the pre-contracted form ------------------------------------------------
    def _guard_bound(self, dtype) -> float:
        if self.poly_guard_max is not None:
            return float(self.poly_guard_max)
        return 10.0 if dtype == torch.float32 else 1.0e4

    def _pack_guarded(self, tensors, meta_head, acc=None):
        """Pack with the guard slot attached.  The decision uses the amplification of the latest pack launch that has
        COMPLETED (no synchronisation: in a training loop that is the previous iteration's, and the weights move by one
        optimizer step in between; hysteresis a_max -> a_max / 2); only the very first pack of a cell waits for its own
        value (one event synchronisation), so a freshly loaded ill-conditioned checkpoint never runs a step in 'poly'."""
        w = tensors[2]
        if self._guard is None:
            self._guard = F_pi.PolyGuard(w.device)
        gd = self._guard
        a_max = self._guard_bound(w.dtype)
        first = gd.seen == 0
        was = gd.factored
        factored = gd.decide(a_max)
        ub, vb = (float(x) for x in self.state_bound)
        sink = None if acc is None else acc.sink          # (the node holds the SINK: holding the state would be a cycle)
        self.__dict__["_acc_delivered"] = sink is not None
        P = F_pi.PackBlockFunction.apply(meta_head + (not factored,), (gd.address, gd.next_seq(), ub, vb), sink, *tensors)
        if first and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
            if gd.decide(a_max) != factored:                  # the first value is in: ill-conditioned from the start
                factored = gd.factored
                P = F_pi.PackBlockFunction.apply(meta_head + (not factored,), (gd.address, gd.next_seq(), ub, vb), sink, *tensors)
        if factored != was:
            import warnings
            if factored and not self._guard_warned:
                self._guard_warned = True
                warnings.warn(
                    f"percnn_amd: reaction='poly' is ill-conditioned for these weights (amplification A = {gd.A:.3g} > "
                    f"{a_max:g} for states within {tuple(self.state_bound)}): this cell evaluates the FACTORED form -- the "
                    f"reference's own operation order, train_2drd.py:115-116 -- until A drops below {0.5 * a_max:g}.  "
                    f"The factored evaluation costs arithmetic the pre-contracted one removes (measured on MI355X, forward + "
                    f"backward rollout at 512^2, Hc = 8: 81 k instead of 295 k time steps/s, 3.6x).  "
                    f"Construct the cell with reaction='factored' to silence this, or set cell.state_bound / "
                    f"cell.poly_guard_max.", RuntimeWarning, stacklevel=4)
        return P

    @property
    def effective_reaction(self) -> str:
        """'factored' while the guard has switched a reaction='poly' cell to the literal evaluation, else ``reaction``."""
        gd = self._guard
        return "factored" if (self.reaction == "poly" and gd is not None and gd.factored) else self.reaction

    def _param_block_uncached(self, acc=None) -> torch.Tensor:
        w = self.W_laplace.weight
        if w.is_cuda and not w.requires_grad:
            # one launch each way (torch.ops.percnn.pack_block): the tensor-op assembly below costs 79 us per call on
            # MI355X without autograd and 540 us forward + backward with it -- as much as a 200-step rollout at 100^2
            if not torch.compiler.is_compiling():
                self._validate_stencil()
            tensors = self._pack_tensors()
            # (anything the kernel's pointer table cannot describe -- views, mixed dtypes / devices -- takes the tensor-op
            # assembly below instead of raising)
            if all(t.is_contiguous() and t.dtype == w.dtype and t.device == w.device for t in tensors):
                head = (self.hidden_channels, self.ndim, float(self.dt),
                        float(self.mu_up) if self.diffusion == "sigmoid" else 0.0, self.diffusion == "sigmoid")
                meta = head + (self.reaction == "poly",)
                if torch.compiler.is_compiling():            # the registered operator is what a graph can hold (no guard
                    return torch.ops.percnn.pack_block(tensors, *meta)    # inside a traced graph: see INTEGRATION.md)
                if self.reaction == "poly" and self.poly_guard:
                    return self._pack_guarded(tensors, head, acc)
                self.__dict__["_acc_delivered"] = acc is not None
                return F_pi.PackBlockFunction.apply(meta, None, None if acc is None else acc.sink, *tensors)
        if torch.compiler.is_compiling():
            # traced by torch.compile: no host-side checks / caches inside the graph (the stencil was validated by the
            # eager call that preceded compilation or is validated by the first eager use)
            dt_t = torch.tensor([self.dt], dtype=w.dtype, device=w.device)
        else:
            self._validate_stencil()
            # the reference reads self.dt every step (train_2drd.py:117): the cached device scalar is keyed on its value
            key = (float(self.dt), w.device, w.dtype)
            if self._dt_cache is None or self._dt_cache[0] != key:
                self._dt_cache = (key, torch.tensor([self.dt], dtype=w.dtype, device=w.device))
            dt_t = self._dt_cache[1]
        cu, cv = self.coefficients()
        branch = []
        for s in ("u", "v"):
            for k in (1, 2, 3, 4):
                m = getattr(self, f"Wh{k}_{s}")
                branch += [m.weight, m.bias]
        P = F_pi.pack_params(dt_t, cu, cv, w, branch)
        return F_pi.contract_block(P) if self.reaction == "poly" else P

    def poly_amplification(self, u_max: float = 1.0, v_max: float = 1.0) -> float:
        """A of the class docstring for states bounded by |u| <= u_max, |v| <= v_max: how much larger the rounding
        noise of reaction='poly' is than the rounding of the state update itself.  Host-side diagnostic (one
        device-to-host copy); see the rule in the class docstring."""
        with torch.no_grad():
            keep, self.reaction = self.reaction, "factored"
            try:
                P = self.param_block().detach().to("cpu", torch.float64)
            finally:
                self.reaction = keep
            Q = F_pi.contract_block(P)
        u, v = float(u_max), float(v_max)
        phi = torch.tensor([1, u, v, u * u, u * v, v * v, u ** 3, u * u * v, u * v * v, v ** 3], dtype=torch.float64)
        a = (Q[16:36].abs().reshape(2, 10) * phi).sum(1).max()
        return float(abs(self.dt) * a / max(u, v))

    # -- reference interface -------------------------------------------------------------------
    def forward(self, h):
        if torch.compiler.is_compiling():
            ch = F_pi.pi_step(h, self.param_block())       # the registered operator is what a graph holds
            return ch, ch
        ext = _native_ext(required=True)
        # hit path of a step loop (train_2drd.py:169-188 calls this T times per iteration): ONE call validates the cached block
        # against the module tree and steps -- a speculated frame, a single launch, or one C++ autograd node whose backward
        # leaves the step's parameter-gradient sums in the block's shared workspace rows
        ch = ext.fast_forward(self, h)
        if ch is None:
            P = self.param_block()                         # (re)pack, then the same entry points
            acc = self.__dict__.get("_block_acc")
            if acc is not None and P is not self._block_cache[1]:
                acc = None
            if not (torch.is_grad_enabled() and (h.requires_grad or P.requires_grad)):
                ch = ext.step_nograd(h, P, acc)
            elif acc is not None:
                ch = ext.cell_step(h, P, acc)
            else:
                ch = F_pi.pi_step(h, P)
        return ch, ch

    def init_hidden_tensor(self, prev_state):
        return prev_state.to(self.W_laplace.weight.device)


def gs2d_cell(hidden_channels: int = 8, reaction: str = "poly") -> RCNNCell:
    """2D Gray-Scott constants (train_2drd.py:56-58, init c=0.02 :90)."""
    return RCNNCell(2, hidden_channels, dx=0.01, dt=0.5, mu_up=3.99e-5, diffusion="sigmoid", dtype=torch.float32,
                    stencil_scale="premul", init="xavier", init_c=0.02, reaction=reaction)


def gs3d_cell(hidden_channels: int = 2, reaction: str = "poly") -> RCNNCell:
    """3D Gray-Scott constants (train_3drd.py:71-73, init c=0.01 :106)."""
    return RCNNCell(3, hidden_channels, dx=100 / 48, dt=0.5, mu_up=0.274, diffusion="sigmoid", dtype=torch.float32,
                    stencil_scale="premul", init="xavier", init_c=0.01, reaction=reaction)


def lo2d_cell(hidden_channels: int = 4, reaction: str = "poly") -> RCNNCell:
    """2D lambda-omega constants, float64 (percnn_LO_eqn.py:12,38-43, init c=0.5 :73)."""
    return RCNNCell(2, hidden_channels, dx=0.2, dt=0.0125, mu_up=None, diffusion="raw", dtype=torch.float64,
                    stencil_scale="div", init="uniform", init_c=0.5, reaction=reaction)


class Stage3LambdaOmegaCell(nn.Module):
    """Stage-3 physics-based lambda-omega cell: drop-in for ``RCNNCell`` of
    DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-3/fine_tuning_LO_[10%noise,41x51x51].py:83-216 --
    13 trainable scalars of the discovered PDE

        u_t = nu_u Lap u + C1_u u + C2_u u^3 + C3_u u^2 v + C4_u u v^2 + C5_u v^3
        v_t = nu_v Lap v + C1_v v + C2_v u^3 + C3_v u^2 v + C4_v u v^2 + C5_v v^3 + C6_v u     (f_rhs, :149-152)

    advanced by explicit Euler (forward, :203-216), float64, periodic.  Same parameter names as the reference
    (``nu_u, nu_v, C1_u..C5_u, C1_v..C6_v, laplace_op.filter.weight``).  It runs on the pre-contracted kernels:
    the scalars ARE the monomial coefficients of the 36-entry block."""

    INIT = dict(nu_u=0.09465, nu_v=0.09455, C1_u=1.0081, C2_u=-1.0167, C3_u=0.9973, C4_u=-1.0176, C5_u=0.9981,
                C1_v=0.9873, C2_v=-0.9987, C3_v=-0.9945, C4_v=-0.9985, C5_v=-0.9928, C6_v=0.0065)   # :126-140

    class _Derivative(nn.Module):
        def __init__(self, resol):
            super().__init__()
            self.resol = resol
            self.filter = nn.Conv2d(1, 1, 5, 1, padding=2, padding_mode="circular", bias=False, dtype=torch.float64)
            self.filter.weight.data = torch.tensor(laplace_stencil(2), dtype=torch.float64)
            self.filter.weight.requires_grad = False

        def forward(self, x):                                       # stock evaluation (f_rhs / forward_rk4 only)
            return self.filter(x) / self.resol

    def __init__(self, dx: float = 0.2, dt: float = 0.0125):
        super().__init__()
        for k, v in self.INIT.items():
            setattr(self, k, nn.Parameter(torch.tensor(v, dtype=torch.float64)))
        self.dx = self.dy = dx
        self.dt = dt
        self.ndim, self.reaction = 2, "poly"
        self.laplace_op = self._Derivative(dx ** 2)
        self._checked = None

    def param_block(self) -> torch.Tensor:
        w = self.laplace_op.filter.weight
        key = (w._version, w.data_ptr())
        if self._checked != key:
            F_pi.check_star_stencil(w)
            self._checked = key
        z = torch.zeros((), dtype=w.dtype, device=w.device)
        taps = (w / self.laplace_op.resol).reshape(-1)              # the reference divides the conv result by dx^2 (:78-80)
        flat = torch.cat([torch.tensor([self.dt], dtype=w.dtype, device=w.device), self.nu_u.reshape(1),
                          self.nu_v.reshape(1), taps])
        head = flat.index_select(0, F_pi._gather_index(1, 2, w.device)[:16])
        # monomial order: 1, u, v, u^2, uv, v^2, u^3, u^2 v, u v^2, v^3
        cu = torch.stack([z, self.C1_u, z, z, z, z, self.C2_u, self.C3_u, self.C4_u, self.C5_u])
        cv = torch.stack([z, self.C6_v, self.C1_v, z, z, z, self.C2_v, self.C3_v, self.C4_v, self.C5_v])
        return torch.cat([head, cu, cv])

    def forward(self, h):
        ch = F_pi.pi_step(h, self.param_block())
        return ch, ch

    def f_rhs(self, u, v):
        """The discovered right-hand side on stock tensor operations (reference f_rhs, lo3:149-152)."""
        u2, v2 = u ** 2, v ** 2
        f_u = self.nu_u * self.laplace_op(u) + self.C1_u * u + self.C2_u * u ** 3 + self.C3_u * u2 * v + self.C4_u * u * v2 \
            + self.C5_u * v ** 3
        f_v = self.nu_v * self.laplace_op(v) + self.C1_v * v + self.C2_v * u ** 3 + self.C3_v * u2 * v + self.C4_v * u * v2 \
            + self.C5_v * v ** 3 + self.C6_v * u
        return f_u, f_v

    def forward_rk4(self, h):
        """One classical Runge-Kutta step with ``f_rhs`` (reference: ``forward_rk4``, lo3:154-201 -- defined there, never called).
        Offered for completeness on stock tensor operations (differentiable through stock autograd, any device); the fused
        kernels implement the explicit Euler ``forward`` the reference trains with."""
        dt = self.dt
        u0, v0 = h[:, 0:1], h[:, 1:2]
        ku, kv = self.f_rhs(u0, v0)
        su, sv = ku, kv                                             # k1 + 2 k2 + 2 k3 + k4, summed in the reference's order
        for weight, frac in ((2, 2.0), (2, 2.0), (1, 1.0)):          # stage states u0 + k dt / 2, u0 + k dt / 2, u0 + k dt
            if frac == 1.0:
                ku, kv = self.f_rhs(u0 + ku * dt, v0 + kv * dt)
            else:
                ku, kv = self.f_rhs(u0 + ku * dt / frac, v0 + kv * dt / frac)
            su = su + (weight * ku if weight != 1 else ku)
            sv = sv + (weight * kv if weight != 1 else kv)
        ch = torch.cat((u0 + dt * su / 6.0, v0 + dt * sv / 6.0), dim=1)
        return ch, ch

    def init_hidden_tensor(self, prev_state):
        return prev_state.to(self.nu_u.device)


class Stage3BurgersCell(nn.Module):
    """Stage-3 physics-based 2D Burgers cell: drop-in for ``RCNNCell`` of
    DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-3/fine_tuning_[5%noise,41x51x51].py:84-221 --

        u_t = nu_u Lap u + C1_u u u_x + C2_u v u_y,      v_t = nu_v Lap v + C1_v u v_x + C2_v v v_y    (f_rhs, :154-157)

    with 4th-order central first derivatives (``dx_2d_op`` differentiates along tensor dim 2, ``dy_2d_op`` along
    dim 3, :20-30), explicit Euler (:209-221), float64, periodic.  Parameter names as in the reference
    (``nu_u, nu_v, C1_u, C2_u, C1_v, C2_v, laplace_op/dx_op/dy_op.filter.weight``).  One fused launch per step
    and per adjoint step (``csrc/pi_adv.h``) instead of 6 convolutions + ~20 elementwise launches."""

    INIT = dict(nu_u=0.0050078, nu_v=0.0050228, C1_u=-0.982252, C2_u=-0.992132, C1_v=-0.983758, C2_v=-0.971269)  # :123-130

    class _Derivative(nn.Module):
        def __init__(self, stencil, resol):
            super().__init__()
            self.resol = resol
            self.filter = nn.Conv2d(1, 1, 5, 1, padding=2, padding_mode="circular", bias=False, dtype=torch.float64)
            self.filter.weight.data = torch.tensor(stencil, dtype=torch.float64)
            self.filter.weight.requires_grad = False

        def forward(self, x):                                       # stock evaluation (f_rhs / forward_rk4 only)
            return self.filter(x) / self.resol

    def __init__(self, dx: float = 1 / 100, dt: float = 0.00025):
        super().__init__()
        for k, v in self.INIT.items():
            setattr(self, k, nn.Parameter(torch.tensor(v, dtype=torch.float64)))
        self.dx = self.dy = dx
        self.dt = dt
        self.ndim, self.reaction = 2, "adv"
        d0, d1 = np.zeros((1, 1, 5, 5)), np.zeros((1, 1, 5, 5))
        for i, val in zip((0, 1, 3, 4), (1 / 12, -8 / 12, 8 / 12, -1 / 12)):
            d0[0, 0, i, 2] = val
            d1[0, 0, 2, i] = val
        self.laplace_op = self._Derivative(laplace_stencil(2), dx ** 2)
        self.dx_op = self._Derivative(d0, dx)
        self.dy_op = self._Derivative(d1, dx)
        self._checked = None

    def param_block(self) -> torch.Tensor:
        w = self.laplace_op.filter.weight
        key = (w._version, w.data_ptr(), self.dx_op.filter.weight._version, self.dy_op.filter.weight._version)
        if self._checked != key:
            F_pi.check_star_stencil(w)
            for op, ax in ((self.dx_op, 0), (self.dy_op, 1)):          # derivative taps must sit on their own axis
                m = torch.ones(5, 5, dtype=torch.bool)
                if ax == 0:
                    m[:, 2] = False
                else:
                    m[2, :] = False
                if bool((op.filter.weight.detach().reshape(5, 5).cpu()[m] != 0).any()):
                    raise ValueError("first-derivative stencil has entries off its axis; unsupported")
            self._checked = key
        dev, dt_ = w.device, w.dtype
        z = torch.zeros((), dtype=dt_, device=dev)
        flat = torch.cat([torch.tensor([self.dt], dtype=dt_, device=dev), self.nu_u.reshape(1), self.nu_v.reshape(1),
                          (w / self.laplace_op.resol).reshape(-1)])
        head = flat.index_select(0, F_pi._gather_index(1, 2, dev)[:16])
        d0 = (self.dx_op.filter.weight / self.dx_op.resol).reshape(5, 5)
        d1 = (self.dy_op.filter.weight / self.dy_op.resol).reshape(5, 5)
        taps = torch.stack([d0[0, 2], d0[1, 2], d0[3, 2], d0[4, 2], d1[2, 0], d1[2, 1], d1[2, 3], d1[2, 4], z, z, z, z])
        # species u: (C1_u u) D0(u) + (C2_u v) D1(u);  species v: (C1_v u) D0(v) + (C2_v v) D1(v);  third axis unused
        adv = torch.stack([self.C1_u, z, z, self.C2_u, z, z, self.C1_v, z, z, self.C2_v, z, z])
        return torch.cat([head, torch.zeros(20, dtype=dt_, device=dev), taps, adv])

    def forward(self, h):
        ch = F_pi.pi_step(h, self.param_block())
        return ch, ch

    def f_rhs(self, u, v):
        """The discovered right-hand side on stock tensor operations (reference f_rhs, bur3:154-157)."""
        f_u = self.nu_u * self.laplace_op(u) + self.C1_u * u * self.dx_op(u) + self.C2_u * v * self.dy_op(u)
        f_v = self.nu_v * self.laplace_op(v) + self.C1_v * u * self.dx_op(v) + self.C2_v * v * self.dy_op(v)
        return f_u, f_v

    def forward_rk4(self, h):
        """One classical Runge-Kutta step with ``f_rhs`` (reference: ``forward_rk4``, bur3:159-206 -- defined there, never called).
        Offered for completeness on stock tensor operations (differentiable through stock autograd, any device); the fused
        kernels implement the explicit Euler ``forward`` the reference trains with."""
        dt = self.dt
        u0, v0 = h[:, 0:1], h[:, 1:2]
        ku, kv = self.f_rhs(u0, v0)
        su, sv = ku, kv                                             # k1 + 2 k2 + 2 k3 + k4, summed in the reference's order
        for weight, frac in ((2, 2.0), (2, 2.0), (1, 1.0)):          # stage states u0 + k dt / 2, u0 + k dt / 2, u0 + k dt
            if frac == 1.0:
                ku, kv = self.f_rhs(u0 + ku * dt, v0 + kv * dt)
            else:
                ku, kv = self.f_rhs(u0 + ku * dt / frac, v0 + kv * dt / frac)
            su = su + (weight * ku if weight != 1 else ku)
            sv = sv + (weight * kv if weight != 1 else kv)
        ch = torch.cat((u0 + dt * su / 6.0, v0 + dt * sv / 6.0), dim=1)
        return ch, ch

    def init_hidden_tensor(self, prev_state):
        return prev_state.to(self.nu_u.device)


# ------------------------------------------------------------------------------------------------
# 3D IC generator: same layers, same parameters, different evaluation.  On MI355X the stock path (MIOpen
# ConvTranspose3d 8 -> 8, 5^3, at 128^3) takes 320 ms forward+backward -- 88 % of a whole 500-step training iteration
# once the rollout is fused (tools/upscaler_share.py) -- so the two transposed convolutions are evaluated as dense
# contractions on rocBLAS instead: 40 -> ~2 ms and 320 -> ~21 ms, results equal to 1e-6 (tests/test_host_logic.py).
# ------------------------------------------------------------------------------------------------
def _unfold3(x: torch.Tensor, k: int) -> torch.Tensor:
    """[1,C,D,H,W] -> [(D-k+1)*(H-k+1)*(W-k+1), C*k^3] (row = output point, column = (c, dz, dy, dx))"""
    c = x.unfold(2, k, 1).unfold(3, k, 1).unfold(4, k, 1)               # [1,C,D',H',W',k,k,k]
    d, h, w = c.shape[2:5]
    return c.permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(d * h * w, -1)


def conv_transpose3d_s2k5(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """ConvTranspose3d(k=5, stride=2, padding=2, output_padding=1) as ONE matmul on the input grid: output point
    2m+p (p = parity per axis) only sees inputs m-1, m, m+1 through tap d = 2 + p - 2e, so
    out[(p, co)](m) = sum_{ci, e in {-1,0,1}^3} x[ci](m + e) * W[ci, co, d(p, e)]  (tap 5 does not exist -> 0)."""
    assert x.dim() == 5 and x.shape[0] == 1 and tuple(weight.shape[2:]) == (5, 5, 5)
    ci, co = weight.shape[:2]
    D, H, W = x.shape[2:]
    dev = x.device
    e = torch.arange(3, device=dev) - 1                                   # -1, 0, 1
    pz = torch.arange(2, device=dev)
    d = 2 + pz[:, None] - 2 * e[None, :]                                  # [parity, e] -> tap, 5 = missing
    ok = d <= 4
    d = d.clamp(max=4)
    # Wp[(ci, ez, ey, ex), (pz, py, px, co)]
    wz = weight[:, :, d]                                                  # [ci, co, p, e, 5, 5]
    wzy = wz[:, :, :, :, d]                                               # [ci, co, pz, ez, py, ey, 5]
    wzyx = wzy[:, :, :, :, :, :, d]                                       # [ci, co, pz, ez, py, ey, px, ex]
    m = (ok[:, :, None, None, None, None] & ok[None, None, :, :, None, None] & ok[None, None, None, None, :, :])
    wzyx = wzyx * m.to(weight.dtype)
    Wp = wzyx.permute(0, 3, 5, 7, 2, 4, 6, 1).reshape(ci * 27, 8 * co)
    cols = _unfold3(torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1)), 3)    # [D*H*W, ci*27]
    out = cols @ Wp                                                       # [D*H*W, 8*co]
    out = out.reshape(D, H, W, 2, 2, 2, co).permute(6, 0, 3, 1, 4, 2, 5).reshape(1, co, 2 * D, 2 * H, 2 * W)
    return out + bias.view(1, co, 1, 1, 1)


def _conv3d_k5_slabs(x: torch.Tensor, w2d: torch.Tensor, slab: int) -> torch.Tensor:
    """'same' 5^3 cross-correlation of x [1,Ci,D,H,W] with w2d [Co, Ci*125], im2col in z-slabs of `slab` planes."""
    D = x.shape[2]
    xp = torch.nn.functional.pad(x, (2, 2, 2, 2, 2, 2))
    outs = []
    for z0 in range(0, D, slab):
        z1 = min(z0 + slab, D)
        cols = _unfold3(xp[:, :, z0:z1 + 4], 5)                          # [(z1-z0)*H*W, Ci*125]
        outs.append((cols @ w2d.t()).reshape(z1 - z0, x.shape[3], x.shape[4], -1))
    return torch.cat(outs, 0).permute(3, 0, 1, 2)[None]


def _conv3d_k5c8_hip(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """percnn_pi_conv3d_k5c8_f32: x [1,8,D,H,W] float32 on a HIP device, wt [8,5,5,5,8] = [ci][dz][dy][dx][co]"""
    from . import _lib
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().percnn_pi_conv3d_k5c8_f32(x.data_ptr(), out.data_ptr(), wt.data_ptr(),
                                                       bias.data_ptr() if bias is not None else None,
                                                       _lib.shape_arg(x.shape[2:]),
                                                       F_pi._stream()), "conv3d_k5c8")
    return out


def _hip_conv_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(weight.shape) == (8, 8, 5, 5, 5)
            and x.shape[0] == 1)


class _ConvTranspose3dS1K5(torch.autograd.Function):
    """ConvTranspose3d(k=5, stride=1, padding=2) == cross-correlation with the flipped, channel-transposed kernel.
    Forward, input gradient and weight gradient are all slab-wise im2col + matmul; the unfolded columns (1 GB per
    16 planes of 128^2 x 8 channels) are never kept between forward and backward."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ci, co = weight.shape[:2]
        ctx.save_for_backward(x, weight)
        ctx.slab = max(1, int(2 ** 28 // max(1, x.shape[3] * x.shape[4] * ci * 125)))
        if _hip_conv_ok(x, weight):                                       # hand-written kernel (csrc/pi_up3d.h)
            return _conv3d_k5c8_hip(x, weight.flip(2, 3, 4).permute(0, 2, 3, 4, 1).contiguous(), bias.contiguous())
        wf = weight.flip(2, 3, 4).transpose(0, 1).reshape(co, ci * 125)   # [co, (ci, d)]
        return _conv3d_k5_slabs(x, wf, ctx.slab) + bias.view(1, co, 1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        ci, co = weight.shape[:2]
        g = g.contiguous()
        # dL/dx = 'same' cross-correlation of g with W itself: [ci, (co, d)]
        if _hip_conv_ok(g, weight):
            gx = _conv3d_k5c8_hip(g, weight.permute(1, 2, 3, 4, 0).contiguous(), None)
        else:
            gx = _conv3d_k5_slabs(g, weight.reshape(ci, co * 125), ctx.slab)
        if _hip_conv_ok(g, weight) and _hip_conv_ok(x, weight):
            from . import _lib
            L = _lib.lib()
            ws = torch.empty(L.percnn_pi_conv3d_k5c8_wgrad_workspace_bytes(), dtype=torch.uint8, device=x.device)
            gwt = torch.empty(8, 5, 5, 5, 8, dtype=torch.float32, device=x.device)     # [ci][dz][dy][dx][co]
            with torch.cuda.device(x.device):
                _lib.check(L.percnn_pi_conv3d_k5c8_wgrad_f32(x.data_ptr(), g.data_ptr(), gwt.data_ptr(), ws.data_ptr(),
                                                             ws.numel(), _lib.shape_arg(x.shape[2:]), F_pi._stream()),
                           "conv3d_k5c8_wgrad")
            # the forward used Wt[ci][d'][co] = weight[ci][co][4 - d']
            return gx, gwt.permute(0, 4, 1, 2, 3).flip(2, 3, 4), g.sum(dim=(0, 2, 3, 4))
        # dL/dW[ci, co, d] = sum_o x[ci](o + 2 - d) g[co](o)  ->  correlate padded x columns with g, then flip the taps
        D, H, W = x.shape[2:]
        xp = torch.nn.functional.pad(x, (2, 2, 2, 2, 2, 2))
        gw = torch.zeros(co, ci * 125, dtype=x.dtype, device=x.device)
        gm = g[0].reshape(co, D, H * W)
        for z0 in range(0, D, ctx.slab):
            z1 = min(z0 + ctx.slab, D)
            cols = _unfold3(xp[:, :, z0:z1 + 4], 5)                      # [(z1-z0)*H*W, ci*125], column (ci, t): x(o + t - 2)
            gw += gm[:, z0:z1].reshape(co, -1) @ cols
        gweight = gw.reshape(co, ci, 5, 5, 5).flip(2, 3, 4).transpose(0, 1)
        return gx, gweight, g.sum(dim=(0, 2, 3, 4))


class Upscaler(nn.Module):
    """IC generator (train_2drd.py:26-41, train_3drd.py:41-56): stock torch.nn parameters (whole-model checkpoints
    load); 2D runs on the stock path (0.6 ms at 512^2), 3D evaluates its two transposed convolutions as dense
    contractions (see above).  Runs once per rollout."""

    def __init__(self, ndim: int = 2):
        super().__init__()
        if ndim == 2:
            layers = [nn.ConvTranspose2d(2, 8, 5, padding=2, stride=2, output_padding=1, bias=True), nn.Sigmoid(),
                      nn.ConvTranspose2d(8, 8, 5, padding=2, stride=2, output_padding=1, bias=True),
                      nn.Conv2d(8, 2, 1, 1, padding=0, bias=True)]
        else:
            layers = [nn.ConvTranspose3d(2, 8, 5, padding=2, stride=2, output_padding=1, bias=True), nn.Sigmoid(),
                      nn.ConvTranspose3d(8, 8, 5, padding=2, stride=1, output_padding=0, bias=True),
                      nn.Conv3d(8, 2, 1, 1, padding=0, bias=True)]
        self.convnet = nn.Sequential(*layers)

    def forward(self, h):
        if h.dim() == 5 and h.shape[0] == 1:
            ct1, act, ct2, out = self.convnet
            y = act(conv_transpose3d_s2k5(h, ct1.weight, ct1.bias))
            y = _ConvTranspose3dS1K5.apply(y, ct2.weight, ct2.bias)
            return out(y)
        return self.convnet(h)


class FrameList(list):
    """The list of frames ``RCNN.forward()`` returns (train_2drd.py:187-188) -- an ordinary list -- plus, when every step is an
    effective step, ``.stacked``: the [step+1, 2, *S] tensor the reference's callers build from it with
    ``torch.cat(tuple(output), dim=0)`` (train_2drd.py:394), as an output of the SAME autograd node.  The frames are views of
    that tensor, so ``output.stacked`` costs nothing where a cat would copy the whole trajectory (2 GiB at 512^2 x 1000) forward
    and its CatBackward slice it again backward.  Since round 5 the frames are ``functional.Frame``s: the
    reference's unchanged ``torch.cat(tuple(output), dim=0)`` returns that same tensor -- no caller edit needed for the zero-copy
    path (runs of consecutive steps; every n-th step keeps the stock cat, whose per-frame gradients the sweep can mask)."""
    stacked: Optional[torch.Tensor] = None


class RCNN(nn.Module):
    """Rollout: ``forward() -> (outputs, second_last_state)`` (train_2drd.py:162-190).

    ``outputs[0]`` is the initial state, ``outputs[k]`` the state after the k-th *effective* step
    (membership in ``effective_step`` honoured -- train_2drd.py:187); all are views of ONE
    trajectory buffer produced by a single fused rollout call.  ``second_last_state`` is the state
    after ``step-1`` steps (clone taken at loop index ``step-2`` -- train_2drd.py:182-184).
    Initial state: a fixed tensor (percnn_LO_eqn.py:158) or ``upscaler(init_state_low)``
    (train_2drd.py:150,164).
    """

    def __init__(self, cell: RCNNCell, step: int = 1, effective_step: Sequence[int] = (1,),
                 init_state: Optional[torch.Tensor] = None, upscaler: Optional[nn.Module] = None,
                 init_state_low: Optional[torch.Tensor] = None, cell_name: str = "crnn_cell",
                 cat_view: Optional[bool] = None):
        super().__init__()
        if (init_state is None) == (upscaler is None):
            raise ValueError("give either init_state or upscaler+init_state_low")
        # cat_view: may ``torch.cat(tuple(outputs), dim=0)`` return the trajectory buffer itself (no copy; the SAME tensor on every
        # call, an alias of every frame -- INTEGRATION.md 1)?  False / PERCNN_CAT_VIEW=0: the stock copying cat, for callers that
        # edit the cat result in place.  ``outputs.stacked`` is there either way.
        self.cat_view = (os.environ.get("PERCNN_CAT_VIEW", "1") not in ("0", "")) if cat_view is None else bool(cat_view)
        self.step = step
        self.effective_step = list(effective_step)
        self.cell_name = cell_name
        self.init_state = init_state
        self.init_state_low = init_state_low
        if upscaler is not None:
            self.UpconvBlock = upscaler            # registered first, as in the reference (state_dict order)
        setattr(self, cell_name, cell)

    @property
    def cell(self):
        return getattr(self, self.cell_name)

    def _block(self):
        """The cell's parameter block for ONE rollout: packed afresh where the cell caches (``RCNNCell.param_block(fresh=True)``
        -- a pack launch per rollout costs nothing and a rollout then never runs on a block made stale by a ``.data`` edit)."""
        cell = self.cell
        if isinstance(cell, RCNNCell):
            return cell.param_block(fresh=True)
        return cell.param_block()

    def trajectory(self) -> torch.Tensor:
        """[step+1, 2, *S]: every state of the rollout (what callers cat together, train_2drd.py:394)."""
        if hasattr(self, "UpconvBlock"):
            self.init_state = self.UpconvBlock(self.init_state_low)
        if hasattr(self.cell, "rollout"):                   # cells with their own kernels (Stage-1 block)
            return self.cell.rollout(self.init_state, self.step)
        return F_pi.pi_rollout(self.init_state, self._block(), self.step)

    def observe(self, t_slice=slice(None), space_stride: int = 1):
        """``torch.cat(self()[0])[t_slice][:, :, ::s, ::s(, ::s)]`` -- the tensor the reference's data loss is computed
        on (train_2drd.py:397, train_3drd.py:403) -- from ONE autograd node that never builds the dense dL/dtraj.
        Needs a dense ``effective_step``.  The full (detached) trajectory of the same rollout is kept in
        ``self.last_trajectory`` for validation-only consumers (the physics loss, train_2drd.py:405)."""
        if self.effective_step != list(range(self.step)):
            raise ValueError("observe() indexes the dense output list: effective_step must be list(range(step))")
        if hasattr(self, "UpconvBlock"):
            self.init_state = self.UpconvBlock(self.init_state_low)
        t_idx = list(range(self.step + 1))[t_slice]
        ndim = self.init_state.dim() - 2
        if hasattr(self.cell, "rollout_observe"):           # cells with their own kernels (Stage-1 block)
            pred, traj = self.cell.rollout_observe(self.init_state, self.step, t_idx, (space_stride,) * ndim)
        else:
            pred, traj = F_pi.pi_rollout_observe(self.init_state, self._block(), self.step, t_idx,
                                                 (space_stride,) * ndim)
        self.last_trajectory = traj
        return pred

    def loss_mse(self, target: Optional[torch.Tensor] = None, t_slice=slice(None), space_stride: int = 1,
                 reduction: str = "mean") -> torch.Tensor:
        """``F.mse_loss(torch.cat(self()[0])[t_slice][..., ::s, ::s], target, reduction)`` -- the reference's data loss
        (train_2drd.py:397-401) -- with the rollout as ONE autograd node and no dL/dtraj:

        * ``space_stride == 1`` (dense in space): ``target`` is None (then ``mean(traj[t_slice] ** 2)``, SURVEY 8d's loss) or
          the full-trajectory tensor [step+1, 2, *S] of which the frames in ``t_slice`` are used; the sweep forms
          ``2/N * (h_t - target_t)`` in-kernel from the state it reads anyway (``pi_rollout_sqerr``);
        * ``space_stride > 1``: the observed sub-lattice through ``observe()`` (sparse injection), ``target`` shaped like
          its result.

        The full (detached) trajectory is kept in ``self.last_trajectory``."""
        if space_stride != 1:
            pred = self.observe(t_slice, space_stride)
            t = torch.zeros_like(pred) if target is None else target
            return torch.nn.functional.mse_loss(pred, t, reduction=reduction)
        if self.effective_step != list(range(self.step)):
            raise ValueError("loss_mse() indexes the dense output list: effective_step must be list(range(step))")
        if hasattr(self.cell, "rollout"):
            traj = self.trajectory()                        # cells with their own kernels: ordinary autograd on the trajectory
            self.last_trajectory = traj.detach()
            sel = traj[t_slice]
            return torch.nn.functional.mse_loss(sel, torch.zeros_like(sel) if target is None else target[t_slice],
                                                reduction=reduction)
        if hasattr(self, "UpconvBlock"):
            self.init_state = self.UpconvBlock(self.init_state_low)
        frames = list(range(self.step + 1))[t_slice]
        loss, traj = F_pi.pi_rollout_sqerr(self.init_state, self._block(), self.step, target, frames, reduction)
        self.last_trajectory = traj
        return loss

    def ic_loss(self, mode: Optional[str] = None) -> torch.Tensor:
        """``get_ic_loss(model)`` of the reference scripts (train_2drd.py:331-338, train_3drd.py:325-332): MSE between the IC
        generator's output and the low-resolution measurement interpolated to ITS output size (bicubic in 2D, trilinear in
        3D -- the reference hard-codes (100, 100) / (48, 48, 48), the grids it trains on)."""
        if not hasattr(self, "UpconvBlock"):
            raise ValueError("ic_loss() needs an upscaler (the reference's UpconvBlock)")
        pred = self.UpconvBlock(self.init_state_low)
        ndim = pred.dim() - 2
        target = torch.nn.functional.interpolate(self.init_state_low, tuple(pred.shape[2:]),
                                                 mode=mode or ("bicubic" if ndim == 2 else "trilinear"))
        return torch.nn.functional.mse_loss(pred, target)

    def forward(self):
        if hasattr(self, "UpconvBlock"):
            self.init_state = self.UpconvBlock(self.init_state_low)
        eff = set(self.effective_step)
        frames = [0] + [k + 1 for k in range(self.step) if k in eff]
        n_out = len(frames)
        if self.step >= 2:
            frames.append(self.step - 1)                    # second_last_state rides along as one more output
        stacked = None
        if hasattr(self.cell, "rollout_frames"):            # cells with their own kernels (Stage-1 block)
            outs = self.cell.rollout_frames(self.init_state, self.step, frames, with_stacked=True)
        else:
            outs = F_pi.pi_rollout_frames(self.init_state, self._block(), self.step, frames, with_stacked=True)
        outs, stacked = outs[:-1], outs[-1]
        if self.cat_view:
            F_pi.link_frames(outs[:n_out], stacked)         # torch.cat(tuple(outputs), dim=0) -> a view of `stacked`, no copy
        outputs = FrameList(outs[:n_out])
        if stacked is not None and n_out == self.step + 1:
            outputs.stacked = stacked                       # dense effective_step: == torch.cat(tuple(outputs), dim=0), no copy
        second_last_state = outs[n_out].clone() if self.step >= 2 else []
        return outputs, second_last_state
