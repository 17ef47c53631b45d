"""percnn_amd -- MI355X-native (gfx950) fused Pi-block time stepping for PeRCNN.

Only the hot path of isds-neu/PeRCNN lives here: the per-step fixed-stencil Laplacian, the
parallel 1x1-conv branches whose Hadamard product is the reaction term, the explicit Euler
update, their adjoint, and the T-step rollout -- as hand-written HIP kernels behind a C-ABI
(``include/percnn_pi.h``), with drop-in ``RCNNCell`` / ``RCNN`` modules on top.

Layout:  ``csrc/`` HIP kernels + the C-ABI translation units (pi_abi.hip: base block, slabs, physics residual;
pi_s1_abi.hip: Stage-1 block on the matrix cores; pi_up3d_abi.hip: 3D IC-generator contraction) ->
``libpercnn_pi.so``;  ``_lib`` ctypes binding + build;  ``ops`` the registered ``torch.ops.percnn.*`` operators;
``functional`` raw calls + autograd front-end;  ``modules`` the reference's
module interface;  ``stage1`` the Stage-1 cell;  ``slab`` multi-GPU slab decomposition;  ``physics`` physics-residual
loss;  ``synthetic`` initial states for benchmarks / tests.
"""
from ._lib import build, lib, set_option, LIB_PATH  # noqa: F401
from .functional import (pi_step, pi_rollout, pack_params, contract_block, param_count, rollout_fwd_, rollout_bwd,  # noqa: F401
                         step_fwd, step_bwd, PiStepFunction, PiRolloutFunction)
from .modules import RCNNCell, RCNN, Upscaler, Stage3LambdaOmegaCell, Stage3BurgersCell, gs2d_cell, gs3d_cell, lo2d_cell, laplace_stencil  # noqa: F401

from . import ops  # noqa: F401  (registers torch.ops.percnn.*)
from . import slab, synthetic, physics, stage1  # noqa: F401
from .stage1 import Stage1Cell  # noqa: F401

__version__ = "0.1.0"
