"""PyTorch-ROCm custom operators of the Pi-block hot path: ``torch.ops.percnn.*``.

north_star / SURVEY 8(b): the existing PeRCNN modules reach the fused kernels through registered operators with
autograd support, as a drop-in for the per-step ATen sequence of ``RCNNCell.forward``
(DataDrivenModeling/2d_gs_rd/train_2drd.py:105-121) at the call sites ``train_2drd.py:393`` (``model()``) and
``train_2drd.py:407`` (``loss.backward()``).  Registered on top of the C-ABI
(``include/percnn_pi.h``) -- the dispatcher sees real schemas, FakeTensor / meta implementations and autograd formulas,
so the operators pass ``torch.library.opcheck`` and trace under ``torch.compile(fullgraph=True)``:

    percnn::contract_block(Tensor params) -> Tensor          (+ contract_block_backward)
    percnn::pack_block(Tensor[] tensors, int hc, int ndim, float dt, float mu_up, bool sigmoid, bool contract) -> Tensor
            (+ pack_block_backward -> Tensor[]): the reference's 19 parameter tensors -> the block, one launch each way
    percnn::pi_step(Tensor h, Tensor params, str options="") -> Tensor
    percnn::pi_step_backward(Tensor h, Tensor params, Tensor g_out, str options="") -> (Tensor, Tensor)
    percnn::pi_rollout(Tensor h0, Tensor params, int steps, str options="") -> Tensor
    percnn::pi_rollout_backward(Tensor traj, Tensor params, Tensor g_traj, str options="") -> (Tensor, Tensor)
    percnn::pi_rollout_observe(Tensor h0, Tensor params, int steps, int[] t_idx, int[] strides, str options="")
            -> (Tensor pred, Tensor traj)
    percnn::pi_rollout_observe_backward(Tensor traj, Tensor params, Tensor g_pred, int[] t_idx, int[] strides,
            str options="") -> (Tensor, Tensor)

``pi_step`` / ``pi_rollout`` and their backward operators are C++ (``csrc/torch_ext.cpp``: ``TORCH_LIBRARY(percnn)`` + C++
``torch::autograd::Function``, linked against ``libpercnn_pi.so``); the others -- one call per rollout or per training
iteration -- are ``torch.library.custom_op`` over the same C-ABI.

``params`` is the packed parameter block of ``include/percnn_pi.h`` (factored, pre-contracted or advective; the kind is
encoded in its length); ``options`` carries per-call tuning overrides ("key=value,...", the keys of
``percnn_pi_set_option``) -- nothing process-wide is touched.  Every operator fails loudly on CPU tensors: there is no
CPU path.

``pi_rollout_frames`` (the reference's list-of-frames return, train_2drd.py:187-188) is NOT a registered operator: its
outputs are T+1 views of ONE trajectory buffer by design, and operators registered with the dispatcher may not return
tensors that alias each other; it stays a ``torch.autograd.Function`` (``functional.PiRolloutFramesFunction``) over the
same C-ABI calls.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import functional as F_pi

_lib_ns = "percnn"


def _opts(options: str):
    return options if options else None


# ------------------------------------------------------------------------------------------------
# factored block -> pre-contracted polynomial block (device-side contraction, csrc/pi_contract.h)
# ------------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_lib_ns}::contract_block", mutates_args=())
def contract_block(params: torch.Tensor) -> torch.Tensor:
    return F_pi.contract_fwd_hip(params.contiguous())


@contract_block.register_fake
def _(params):
    return params.new_empty((F_pi.NPOLY,))


@torch.library.custom_op(f"{_lib_ns}::contract_block_backward", mutates_args=())
def contract_block_backward(params: torch.Tensor, g_poly: torch.Tensor) -> torch.Tensor:
    return F_pi.contract_bwd_hip(params.contiguous(), g_poly.contiguous())


@contract_block_backward.register_fake
def _(params, g_poly):
    return torch.empty_like(params)


def _contract_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _contract_bwd(ctx, g):
    (params,) = ctx.saved_tensors
    return torch.ops.percnn.contract_block_backward(params, g)


contract_block.register_autograd(_contract_bwd, setup_context=_contract_setup)


# ------------------------------------------------------------------------------------------------
# the reference's parameter tensors -> packed (optionally pre-contracted) block, one launch each way
# (csrc/pi_contract.h: pi_pack_fwd_kernel / pi_pack_bwd_kernel; RCNNCell.param_block)
#   tensors = [CA|DA, CB|DB, W_laplace.weight, Wh1_u.weight, Wh1_u.bias, ..., Wh4_u.bias, Wh1_v.weight, ..., Wh4_v.bias]
# ------------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_lib_ns}::pack_block", mutates_args=())
def pack_block(tensors: List[torch.Tensor], hc: int, ndim: int, dt: float, mu_up: float, sigmoid: bool,
               contract: bool) -> torch.Tensor:
    return F_pi.pack_fwd_hip(tensors, hc, ndim, dt, mu_up, sigmoid, contract)


@pack_block.register_fake
def _(tensors, hc, ndim, dt, mu_up, sigmoid, contract):
    return tensors[2].new_empty((F_pi.NPOLY if contract else F_pi.param_count(hc),))


@torch.library.custom_op(f"{_lib_ns}::pack_block_backward", mutates_args=())
def pack_block_backward(tensors: List[torch.Tensor], g_block: torch.Tensor, hc: int, ndim: int, dt: float, mu_up: float,
                        sigmoid: bool, contract: bool) -> List[torch.Tensor]:
    return F_pi.pack_bwd_hip(tensors, g_block, hc, ndim, dt, mu_up, sigmoid, contract)


@pack_block_backward.register_fake
def _(tensors, g_block, hc, ndim, dt, mu_up, sigmoid, contract):
    return [torch.empty_like(t) for i, t in enumerate(tensors) if i != 2]


def _pack_setup(ctx, inputs, output):
    tensors, hc, ndim, dt, mu_up, sigmoid, contract = inputs
    ctx.save_for_backward(*tensors)
    ctx.meta = (hc, ndim, dt, mu_up, sigmoid, contract)


def _pack_bwd(ctx, g):
    tensors = list(ctx.saved_tensors)
    grads = torch.ops.percnn.pack_block_backward(tensors, g.contiguous(), *ctx.meta)
    return [grads[0], grads[1], None] + list(grads[2:]), None, None, None, None, None, None


pack_block.register_autograd(_pack_bwd, setup_context=_pack_setup)


# ------------------------------------------------------------------------------------------------
# one step, T-step rollout: defined in C++ (csrc/torch_ext.cpp -- TORCH_LIBRARY(percnn): HIP implementation, C++
# torch::autograd::Function registered on the Autograd key; dispatcher -> C-ABI launcher with no Python frame).  Here: loading
# the library and the FakeTensor implementations (what torch.compile / opcheck trace with).
# ------------------------------------------------------------------------------------------------
_native_loaded = False


def load_native() -> None:
    """Import csrc/percnn_torch.so (once) and attach the fake implementations of its operators.  Called at package import when
    the library has been built, and by every entry point that needs the operators otherwise -- a missing library raises."""
    global _native_loaded
    if _native_loaded:
        return
    from . import _lib
    _lib.torch_ext()

    @torch.library.register_fake(f"{_lib_ns}::pi_step")
    def _(h, params, options=""):
        return torch.empty_like(h, memory_format=torch.contiguous_format)

    @torch.library.register_fake(f"{_lib_ns}::pi_step_backward")
    def _(h, params, g_out, options=""):
        return torch.empty_like(h, memory_format=torch.contiguous_format), torch.empty_like(params)

    @torch.library.register_fake(f"{_lib_ns}::pi_rollout")
    def _(h0, params, steps, options=""):
        return h0.new_empty((steps + 1,) + tuple(h0.shape[1:]))

    @torch.library.register_fake(f"{_lib_ns}::pi_rollout_backward")
    def _(traj, params, g_traj, options=""):
        return traj.new_empty((1,) + tuple(traj.shape[1:])), torch.empty_like(params)

    _native_loaded = True


# ------------------------------------------------------------------------------------------------
# rollout + observation operator (what the reference's data loss looks at: output[0:-1:20, :, ::4, ::4],
# train_2drd.py:397; [:-1:15, :, ::2, ::2, ::2], train_3drd.py:403)
# ------------------------------------------------------------------------------------------------
def _sub(strides):
    return (slice(None),) + tuple(slice(None, None, int(s)) for s in strides)


@torch.library.custom_op(f"{_lib_ns}::pi_rollout_observe", mutates_args=())
def pi_rollout_observe(h0: torch.Tensor, params: torch.Tensor, steps: int, t_idx: List[int], strides: List[int],
                       options: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    F_pi._check_state(h0)
    params = params.contiguous()
    traj = torch.empty((steps + 1,) + tuple(h0.shape[1:]), dtype=h0.dtype, device=h0.device)
    traj[0].copy_(h0[0])
    F_pi.rollout_fwd_(traj, params, options=_opts(options))
    pred = F_pi._observe(traj, tuple(int(t) % (steps + 1) for t in t_idx), _sub(strides))
    return pred, traj


@pi_rollout_observe.register_fake
def _(h0, params, steps, t_idx, strides, options=""):
    spatial = [(n + s - 1) // s for n, s in zip(h0.shape[2:], strides)]
    return (h0.new_empty((len(t_idx), h0.shape[1]) + tuple(spatial)),
            h0.new_empty((steps + 1,) + tuple(h0.shape[1:])))


@torch.library.custom_op(f"{_lib_ns}::pi_rollout_observe_backward", mutates_args=())
def pi_rollout_observe_backward(traj: torch.Tensor, params: torch.Tensor, g_pred: torch.Tensor, t_idx: List[int],
                                strides: List[int], options: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    steps = traj.shape[0] - 1
    g_traj = torch.empty_like(traj)                    # never initialised as a whole: unobserved frames are masked
    mask = F_pi._scatter_observed(g_traj, tuple(int(t) % (steps + 1) for t in t_idx), _sub(strides), g_pred)
    g_h0, pg = F_pi.rollout_bwd(traj, g_traj, params.contiguous(), frame_mask=mask, options=_opts(options))
    return g_h0[None], pg.to(params.dtype)


@pi_rollout_observe_backward.register_fake
def _(traj, params, g_pred, t_idx, strides, options=""):
    return traj.new_empty((1,) + tuple(traj.shape[1:])), torch.empty_like(params)


def _observe_setup(ctx, inputs, output):
    _h0, params, _steps, t_idx, strides, options = inputs
    _pred, traj = output
    ctx.mark_non_differentiable(traj)
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(traj, params)
    ctx.t_idx, ctx.strides, ctx.options = list(t_idx), list(strides), options


def _observe_bwd(ctx, g_pred, g_traj):
    # `traj` is returned for inspection / logging only: the operator declares it non-differentiable (setup_context), and a
    # gradient that reaches it nevertheless -- a caller that put a loss on the second output of the raw operator -- must
    # not be dropped silently
    if g_traj is not None:
        raise RuntimeError("percnn::pi_rollout_observe: its `traj` output is not differentiable (a loss on the trajectory "
                           "itself goes through percnn::pi_rollout, or through RCNN.trajectory())")
    traj, params = ctx.saved_tensors
    g_h0, g_p = torch.ops.percnn.pi_rollout_observe_backward(traj, params, g_pred.contiguous(), ctx.t_idx, ctx.strides,
                                                             ctx.options)
    return g_h0, g_p, None, None, None, None


pi_rollout_observe.register_autograd(_observe_bwd, setup_context=_observe_setup)


import os as _os

from . import _lib as _lib_mod

if _os.path.exists(_lib_mod.TORCH_EXT_PATH) and _os.path.exists(_lib_mod.LIB_PATH):
    # a tree that has not been built yet -- or holds a library older than its sources -- still imports (percnn_amd.build()
    # lives in the package); the first entry point that needs the operators loads them and raises if it cannot
    # (ImportError / OSError: a percnn_torch.so built against another torch / ROCm -- undefined symbols at dlopen; CPU-side
    # tools that never need the operators must still be able to import the package)
    try:
        load_native()
    except (RuntimeError, ImportError, OSError):
        pass
