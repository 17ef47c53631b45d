"""percnn_amd -- MI355X-native (gfx950) fused Pi-block time stepping for PeRCNN.

Only the hot path of isds-neu/PeRCNN lives here: the per-step fixed-stencil Laplacian, the
parallel 1x1-conv branches whose Hadamard product is the reaction term, the explicit Euler
update, their adjoint, and the T-step rollout -- as hand-written HIP kernels behind a C-ABI
(``include/percnn_pi.h``), with drop-in ``RCNNCell`` / ``RCNN`` modules on top.
"""
from ._lib import build, lib, set_option, LIB_PATH  # noqa: F401
from .functional import (pi_step, pi_rollout, pack_params, contract_block, param_count, rollout_fwd_, rollout_bwd,  # noqa: F401
                         step_fwd, step_bwd, PiStepFunction, PiRolloutFunction)
from .modules import RCNNCell, RCNN, Upscaler, Stage3LambdaOmegaCell, Stage3BurgersCell, gs2d_cell, gs3d_cell, lo2d_cell, laplace_stencil  # noqa: F401

from . import slab, synthetic, physics, stage1  # noqa: F401
from .stage1 import Stage1Cell  # noqa: F401

__version__ = "0.1.0"
