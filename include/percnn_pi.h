/* percnn_pi.h -- C ABI of the MI355X-native Pi-block time-stepping library (libpercnn_pi.so).
 *
 * The reference (isds-neu/PeRCNN) is pure Python/PyTorch and has NO FFI or operator-plugin
 * interface of its own: the hot path is reached through two nn.Module call signatures.  Each
 * entry point below therefore cites the reference *call* it replaces; the reference-side
 * binding a maintainer would add is the ctypes stub shown in INTEGRATION.md.  Paths are
 * relative to the reference repository root:
 *   2dgs = DataDrivenModeling/2d_gs_rd/train_2drd.py
 *   3dgs = DataDrivenModeling/3d_gs_rd/train_3drd.py
 *   lo   = ForwardSimulationOfPDEs/2d_lambda_omega/percnn_LO_eqn.py
 *
 * Conventions (all entry points)
 *   - Plain pointers and sizes only; every data pointer is DEVICE memory owned by the caller.
 *   - State layout: PyTorch-contiguous NCHW / NCDHW with N = 1, i.e. species-major planar
 *     [2][*S]; `shape` lists the spatial extents slowest-first (2D: {H, W}; 3D: {D, H, W}).
 *   - Periodic boundaries on every axis (2dgs:108-109, 3dgs:125-127); every extent >= 2.
 *   - Asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream); no host
 *     synchronisation and no allocation inside, hence hipGraph-capturable.  Re-entrant: the only
 *     process-wide mutable state is the table of tuning DEFAULTS (percnn_pi_set_option), copied once, under a
 *     lock, at the entry of every call; the *_opt entry points overlay per-call overrides on that copy, so
 *     concurrent calls (threads, models) with different options do not interact.  (Plus a lazily created
 *     internal side stream + events per device, used by the "overlap" schedules only.)
 *   - Output buffers must not alias inputs.
 *   - Return value: 0 on success, otherwise a hipError_t cast to int, or one of the negative
 *     PERCNN_PI_E* codes for argument errors.  No C++ exceptions cross this boundary.
 *
 * Parameter block `params` (device array of the compute type, percnn_pi_param_count(hc) entries)
 *   [0]  dt                       Euler step                      (2dgs:57, 3dgs:72, lo:39)
 *   [1]  coef_u  [2] coef_v       diffusion coefficients: mu_up*sigmoid(CA|CB) (2dgs:115-116) or DA|DB (lo:107-108)
 *   [3]  centre tap of W_laplace.weight (already scaled by 1/dx^2 -- 2dgs:66; never re-derived)
 *   [4 + 4*a + i]  tap of spatial axis a (0 = slowest) at offset {-2,-1,+1,+2}[i]; 12 slots
 *   [16 + s*(10*hc+1) + 10*j + k], species s in {u,v}, hidden channel j:
 *        k = 0,1,2: Wh1_s.weight[j,0], Wh1_s.weight[j,1], Wh1_s.bias[j]     (2dgs:70-71)
 *        k = 3,4,5: Wh2_s ...        k = 6,7,8: Wh3_s ...                   (2dgs:72-75)
 *        k = 9    : Wh4_s.weight[0,j]                                       (2dgs:76-77)
 *   [16 + s*(10*hc+1) + 10*hc]  Wh4_s.bias[0]
 * Gradient blocks (`param_grad`, double) use the same indexing; slots 0 and 3..15 (dt, frozen
 * stencil -- 2dgs:67) receive 0.
 *
 * Two further block kinds are selected through the `hc` argument of every entry point:
 *   hc ==  0  pre-contracted polynomial block, 36 entries: [16 + 10*s + m] = coefficient of monomial m of
 *             {1,u,v,u^2,uv,v^2,u^3,u^2 v,u v^2,v^3} in the reaction term of species s (the expansion of
 *             Wh4(Wh1*Wh2*Wh3) the reference prints at 3dgs:442-468; also the Stage-3 lambda-omega cell,
 *             DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-3/fine_tuning_LO_[10%noise,41x51x51].py:149-152)
 *   hc == -1  advective polynomial block, 60 entries (not in slab mode): [0..35] as above,
 *             [36 + 4*a + i] first-derivative tap of axis a at offset {-2,-1,+1,+2}[i],
 *             [48 + 6*s + 2*a + {0,1}] = (cu, cv) of the term (cu*u + cv*v) * D_a(h_s) in species s
 *             (2D Burgers Stage-3 f_rhs, DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-3/
 *             fine_tuning_[5%noise,41x51x51].py:154-157)
 */
#ifndef PERCNN_PI_H
#define PERCNN_PI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PERCNN_PI_ABI_VERSION 3   /* 2: percnn_pi_halo_ring gained `peer`; percnn_pi_peer_*, *_opt entry points.
                                    3: step_bwd_rows, bwd_rows_finish, rollout_bwd_top, pack_fwd_guard, host_words_*, persist_status,
                                       PERCNN_PI_EASYNC; debug_plan out[14] is a bit field; round 5: persist_fence */

#define PERCNN_PI_SWEEP_ONLY 1    /* flags of percnn_pi_slab_step_bwd_*: adjoint state + diffusion-coefficient
                                   * gradients only; branch gradients come from percnn_pi_slab_wgrad_* later */
#define PERCNN_PI_NO_RESET 2      /* the workspace's partial rows already hold sums of earlier launches of this sweep */
#define PERCNN_PI_NO_FINISH 4     /* leave this launch's sums in the partial rows; a later launch without this flag
                                   * (and with NO_RESET) reduces everything into param_grad */

#define PERCNN_PI_EINVAL   (-1)  /* bad ndim / hc / shape / NULL pointer          */
#define PERCNN_PI_EWORKSPACE (-2) /* workspace smaller than *_workspace_bytes says */
#define PERCNN_PI_ETOOLARGE (-3)  /* grid beyond the 32-bit byte offsets of the step kernels: a 2D field (+ 4 rows) or a 3D
                                   * plane of 4 GiB or more per species (e.g. 32768^2 float32); nothing was launched */
#define PERCNN_PI_EASYNC (-4)     /* an EARLIER call's persistent tile sweep (option "tile_persist", launched with
                                   * "persist_handshake=0") aborted on the device: the outputs of that call are invalid.  Reported
                                   * once, by the first entry point called after the abort; nothing was launched by this call.
                                   * percnn_pi_persist_status tells where.  (With the handshake -- the default -- an aborted sweep is
                                   * re-run launch by launch inside the same call and no error is ever reported.) */

/* ABI version of the loaded library (== PERCNN_PI_ABI_VERSION it was built with). */
int percnn_pi_abi_version(void);

/* Number of entries of the parameter / gradient block for `hc` hidden channels:
 * 16 + 2*(10*hc+1).  Mirrors the parameter census of RCNNCell.__init__ (2dgs:46-90). */
size_t percnn_pi_param_count(int hc);

/* Factored block (hc >= 1) -> pre-contracted polynomial block (36 entries, "hc = 0"): the cubic the three 1x1 branches
 * and the 1x1 aggregation of 2dgs:115-116 multiply out to (the expansion 3dgs:442-468 prints), float64 arithmetic rounded
 * once; header entries [0,16) are copied.  _bwd is its exact chain rule: g_poly = dL/d(poly block) (36 entries) ->
 * g_params = dL/d(factored block) (percnn_pi_param_count(hc) entries, overwritten).  One single-workgroup launch each. */
int percnn_pi_contract_fwd_f32(const float *params, int hc, float *poly, void *stream);
int percnn_pi_contract_fwd_f64(const double *params, int hc, double *poly, void *stream);
int percnn_pi_contract_bwd_f32(const float *params, int hc, const float *g_poly, float *g_params, void *stream);
int percnn_pi_contract_bwd_f64(const double *params, int hc, const double *g_poly, double *g_params, void *stream);


/* ---- parameter packing in one launch (replaces the tensor-op assembly of RCNNCell's parameters) ----------------------
 * The reference keeps its parameters as 18 trainable tensors + the frozen stencil (RCNNCell.__init__, train_2drd.py:46-90).
 * percnn_pi_pack_fwd_* gathers them into the packed block of this header -- `contract` != 0: into its pre-contracted
 * 36-entry form (== percnn_pi_contract_fwd_* of the factored block, bit for bit) -- with ONE single-workgroup launch;
 * percnn_pi_pack_bwd_* maps dL/d(block) back onto the 18 tensors (contraction chain rule, sigmoid derivative).
 *   c[0], c[1]   CA, CB when `sigmoid` (coefficient = mu_up * sigmoid(C), train_2drd.py:115), else the raw DA, DB
 *   w            W_laplace.weight, 5^ndim taps (frozen in the reference: no gradient is produced for it)
 *   branch[16]   Wh1_u.weight, Wh1_u.bias, Wh2_u.weight, ..., Wh4_u.bias, then the same eight tensors of species v
 * All pointers are device pointers to contiguous tensors of the entry point's element type; in `grads` NULL slots are
 * skipped and `w` is ignored. */
typedef struct percnn_pi_param_ptrs {
    const void* c[2];
    const void* w;
    const void* branch[16];
} percnn_pi_param_ptrs;
int percnn_pi_pack_fwd_f32(const percnn_pi_param_ptrs* params, int hc, int ndim, double dt, double mu_up, int sigmoid,
                           int contract, float* block, void* stream);
int percnn_pi_pack_fwd_f64(const percnn_pi_param_ptrs* params, int hc, int ndim, double dt, double mu_up, int sigmoid,
                           int contract, double* block, void* stream);
/* ... with the CONDITIONING GUARD of the pre-contracted form: the reference evaluates Wh4(Wh1*Wh2*Wh3) literally
 * (train_2drd.py:115-116); the expanded cubic the `contract` block holds cancels its monomials only to rounding, so its
 * per-step noise relative to the state is eps * A,  A = |dt| * max_s sum_m |c[s][m]| phi_m(u_max, v_max) / max(u_max, v_max).
 * The launch writes {A, seq} (A first, then seq with release semantics) to `host_slot`, two doubles of HOST-MAPPED memory
 * (percnn_pi_host_words_alloc), whether it packs the contracted or the factored block -- the caller reads them without
 * synchronising and packs the factored block while A is above its bound (float32: 10, float64: 1e4; RCNNCell.param_block). */
int percnn_pi_pack_fwd_guard_f32(const percnn_pi_param_ptrs* params, int hc, int ndim, double dt, double mu_up, int sigmoid,
                                 int contract, float* block, double u_max, double v_max, double* host_slot, double seq,
                                 void* stream);
int percnn_pi_pack_fwd_guard_f64(const percnn_pi_param_ptrs* params, int hc, int ndim, double dt, double mu_up, int sigmoid,
                                 int contract, double* block, double u_max, double v_max, double* host_slot, double seq,
                                 void* stream);
/* `bytes` of zeroed host memory mapped into the device's address space (hipHostMalloc, coherent): written by kernels with
 * system-scope stores, read by the host with plain loads.  Returns 0 or a hipError_t. */
int percnn_pi_host_words_alloc(void** p, size_t bytes);
int percnn_pi_host_words_free(void* p);
int percnn_pi_pack_bwd_f32(const percnn_pi_param_ptrs* params, const percnn_pi_param_ptrs* grads, int hc, int ndim, double dt,
                           double mu_up, int sigmoid, int contract, const float* g_block, void* stream);
int percnn_pi_pack_bwd_f64(const percnn_pi_param_ptrs* params, const percnn_pi_param_ptrs* grads, int hc, int ndim, double dt,
                           double mu_up, int sigmoid, int contract, const double* g_block, void* stream);

/* Bytes of scratch the backward entry points need for a grid of this shape
 * (two adjoint ping-pong states + per-workgroup gradient partials). elem_size = 4 or 8. */
size_t percnn_pi_bwd_workspace_bytes(int hc, int ndim, const int64_t *shape, int elem_size);

/* Bytes of scratch percnn_pi_rollout_bwd_* needs for a T-step rollout: the adjoint trajectory
 * (T+1 frames of [2][*S]) + per-workgroup gradient partials. */
size_t percnn_pi_rollout_bwd_workspace_bytes(int hc, int ndim, const int64_t *shape, int T, int elem_size);

/* Tuning / diagnostic options: process-wide DEFAULTS (see the *_opt entry points at the end of this header for per-call
 * overrides).  Every setting computes the same values (state fields bit-identical,
 * gradient sums to reduction round-off); defaults are what measured fastest on MI355X:
 *   "block"        workgroup size of the per-step direct kernels (64..256, multiple of 64; default 256)
 *   "vec"          1 = one point per lane instead of 16 bytes per lane
 *   "wgrad_blocks" grid cap of the time-parallel gradient kernels
 *   "tile"         2D temporally blocked kernels: 0 never, 1 (default) below 1 M points, 2 whenever the shape allows
 *   "tile_k"       time steps per tile launch: 2, 4 (default), 8 (pre-contracted blocks only)
 *   "tile_nt"      threads per tile workgroup: 256, 512 (default), 1024
 *   "tile_by"      tile height: 8, 16, 32, 0 (default) = 8 while 32x8 tiles fit one per CU, 16 while 32x32 tiles would leave
 *                  more than half the CUs idle, else 32
 *   "tile_xcd"     1 (default) = XCD-aware block -> tile map
 *   "stream3d"     3D plane-streaming kernels: 0 never, 1 (default) by size, 2 whenever W == 64 * lanes' vector width
 *   "zc"           planes per workgroup of the plane-streaming kernels (default 8; the adjoint uses twice that)
 *   "fuse_wgrad"   parameter gradients reduced inside the sweep launches instead of one time-parallel pass:
 *                  0 never, 1 wherever a fused flavour exists, 2 (default) float32 pre-contracted blocks on the direct
 *                  / plane-streaming kernels
 *   "tile_fuse"    1 (default): the pre-contracted tile sweep with 32x32 tiles reduces the 20 coefficient moments itself
 *                  and stores only every K-th adjoint frame (no separate moments pass; float32: register accumulators,
 *                  float64: per-lane accumulators in LDS updated with ds_add_f64); 0: split schedule
 *   "tile_persist" 2 (default): where it applies (float32 pre-contracted blocks, whole 32x32 tiles, 16 .. #CUs tiles, >= 8
 *                  steps, frame masks up to 4096 frames) the whole tile sweep of percnn_pi_rollout_bwd_* runs as ONE launch of resident workgroups
 *                  that keep the adjoint tile in LDS and hand their halos over through device memory (512^2: 218 -> 235 k
 *                  steps/s).  One workgroup fills a CU's LDS, so the grid is resident unless ANOTHER kernel holds whole CUs:
 *                  calls on other streams of the process are detected and use one launch per K steps; a hand-over that waits
 *                  longer than 2 s fills its halo with NaNs instead of hanging.  PROCESSES THAT SHARE ONE GPU MUST SET 0.
 *                  1 = the same through hipLaunchCooperativeKernel (residency guaranteed, +0.4 ms per launch); 0 = off
 *   "tile_wide"    3 (default): float32 pre-contracted blocks past 512^2 use 32x40 / 40x40 tiles while that keeps the grid in
 *                  one resident round; 0 never; 1 / 2 force a shape
 *   "rz"           direct 3D kernels, pre-contracted blocks: consecutive planes per workgroup pass that share their plane
 *                  neighbours in registers (1, 2, 4; default 0 = by grid size: large grids 4 forward / 2 backward)
 *   "block_small"  1 (default): 128-thread workgroups for the direct kernels on grids below ~1 M points
 *   "slab_local_index"   1 (default): native slab rollouts of ONE rank (ring == NULL) resolve the periodic wrap by index inside the
 *                        step launches -- no face copies, no recomputed halo planes, the halo planes of the frames are neither read
 *                        nor written; 0: by face copies into the halo planes (the launches of a multi-rank run minus its
 *                        transport).  Per call: bit 1 of the `overlap` argument of percnn_pi_slab_rollout_* selects the copies.
 *   "slab_wide_adjoint"  1: the native slab backward over an RCCL ring exchanges once per two adjoint steps (default 0:
 *                  measured slower, an ncclGroup costs per operation)
 *   "lane_x"       direct kernels: log2 of the 16-byte chunks a wave takes from one row (2..6), or 7 = flat (a workgroup
 *                  takes consecutive chunks of the plane across row ends); default 0 = the decomposition with the most
 *                  useful lanes, weighted by segment length and row-pitch alignment (grids that are not a power of two
 *                  wide: 48^3 +14 %, 144^3 .. 200^3 +20-30 %); -1 = the earlier rule (next power of two >= chunks per row)
 *   "fwd_blocks", "xcd_window"   forward direct kernel: bounded persistent grid / windowed XCD remap (measured: no gain; off)
 *   "l2_tile_kb"   direct 3D kernels: the rows of a plane are processed in y-tiles of this many KiB (both species; default
 *                  128, 0 = whole planes) and the workgroups march along axis 0 tile by tile, so the five planes a tile's
 *                  stencil reads stay in the XCD's L2 when whole planes do not fit (e.g. 384^3); "l2_tile_min_kb" (default
 *                  1536) = size of four neighbour planes x two species from which the tiling is applied
 *   "bwd_cpl"      direct adjoint kernel: 16-byte chunks per lane (1..16, default 2) once >= 512 workgroups remain
 *   "overlap", "overlap_chunk"  run the time-parallel gradient pass of finished chunks on a side stream under the sweep
 *   "skip_wgrad"   diagnostics: adjoint sweep only, parameter gradients of the branches come back as zeros
 *   "lds_pad"      diagnostics: extra dynamic LDS per workgroup (limits workgroups per CU)
 *   "tile_persist" 2D float32 pre-contracted blocks on whole 32 x 32 tiles, 16 .. #CUs tiles: the whole tile sweep of a rollout
 *                  backward as ONE launch of resident workgroups that hand their halos to each other (2 = plain launch, default;
 *                  1 = hipLaunchCooperativeKernel; 0 = one launch per four steps).  RESIDENCY IS CHECKED, NOT ASSUMED: every
 *                  workgroup answers a roll call when it starts, a hand-over wait is bounded ("persist_first_timeout_ms",
 *                  default 100, for the first one; "persist_timeout_ms", default 2000, later), and a workgroup whose wait runs out
 *                  aborts the whole launch -- no output of an aborted launch is ever written.
 *   "persist_handshake"  1 (default): the entry point waits on a host-mapped word (no stream synchronisation) until the launch
 *                  reports "all resident" -- normally microseconds after the kernel starts, i.e. the call returns once the stream
 *                  has reached the sweep -- or "aborted"; after an abort the launch-per-group sweep is enqueued by the same call,
 *                  a one-line warning goes to stderr and the device keeps the launch-per-group path until "persist_reset".  0: no
 *                  wait; an abort is reported by the next entry point as PERCNN_PI_EASYNC.
 *   "persist_split"  1 (default): the persistent sweep computes the halo-independent part of every step while the granules
 *                  travel (pi_adj2d_persist_split_kernel); 0: round 3's kernel
 *   "persist_small"  1 (default): grids in the 32 x 8-tile regime (below ~300^2, ragged ones included: the reference's own
 *                  100^2, train_2drd.py:597-636) run their sweep as one resident launch as well (pi_adj2d_persist_small_kernel;
 *                  its granule outbox is part of percnn_pi_rollout_bwd_workspace_bytes); 0: one launch per four steps
 *                  Round 5: the FORWARD of that regime too (pi_fwd2d_persist_small_kernel, 32 x 8 tiles, T >= 32; gated by
 *                  "fwd_persist" as well; scratch from the per-device granule outbox of the resident forward); round 6: 32 x 16 tiles
 *                  too and ragged grids of 32 x 32 tiles too (with "fwd_small_half").  2: up to 2 x "fwd_persist_per_cu" workgroups per CU
 *                  (tests and experiments only)
 *   "fwd_small_pause"  -1 .. 200, default -1 (24 up to 64 tiles, 28 above, 52 for 32-row tiles): units of 64 clocks the small-tile resident forward waits between publishing its tile
 *                  and the first request of its ring (granules asked for too early come back stale)
 *   "fwd_small_half"   0 | 1, default 1: the small-tile resident forward works its tiles on two-point half-strips (512 lanes at
 *                  32 x 8, 640 at 32 x 16, 1024 at 32 x 32) and hands them over as 16-byte granules of two values; 0: round 5's kernels
 *   "adj_small_half"   0 | 1, default 1: the same for the small-tile resident sweep
 *   "adj_small_pause"  -1 .. 200, default -1 (20 up to 64 tiles, 28 above): the same for the small-tile resident sweep
 *   "fwd_persist"  1 (default): the FORWARD rollout of a grid the persistent sweep takes (float32 pre-contracted block, whole
 *                  32 x 32 tiles, 16 .. #CUs of them, T >= 32) runs as one launch of resident workgroups too
 *                  (pi_fwd2d_persist_kernel: the launch-per-group kernel's trajectory bit for bit; residency check, abort and
 *                  fallback as "tile_persist"; the library keeps 256 B + 24 KiB per tile of device scratch per device for its
 *                  granule outbox, allocated at the first such call); 0: one launch per four steps
 *   "fwd_persist_f64"  1 (default): ... and of float64 pre-contracted blocks (lambda-omega, percnn_LO_eqn.py:12) on 16-byte
 *                  granules {lo32, tag, hi32, tag}; its scratch is 256 B + 48 KiB per tile; 0: one launch per four steps
 *   "adj_persist_f64"  1 (default): the float64 tile SWEEP as one resident launch too (pi_adj2d_persist_split_kernel<double>:
 *                  16-byte granules, the 20 moment sums in [20][256] LDS rows shared by two lanes; needs "persist_split" = 1);
 *                  0: one fused launch per four steps
 *   "fwd_persist_per_cu"  1 (default) or 2: two of its 77 KB workgroups fit a CU, so grids of up to 2 x #CUs tiles can run the
 *                  resident forward (576^2 .. 704^2: -9 .. -16 % per forward step, profiles/r04_forward_persistent.txt).  Not the
 *                  default: with every CU doubly booked any other kernel that holds LDS makes the launch abort, and an abort
 *                  switches the resident launches off for the device until "persist_reset"
 *   "brick_xny"    3D brick kernels, XCD regions: 0 (default) = sized by what an L2 holds (three plane groups x rows x arrays
 *                  touched <= 2.5 MiB; among the maps that fit, the smallest halo share), 1 = contiguous plane ranges, 2 / 4 / 8 =
 *                  force that many row strips, -1 = round 4's rule for forward steps of >= 8 M points.  Pure placement.
 *   "brick_wide"   1 (default): 3D rows of 65 .. 128 sixteen-byte chunks on 512-lane bricks where they win; 0: direct kernels
 *   "res3d"        1 (default): 3D float32 pre-contracted blocks on grids of whole 16 x 16 x 32 blocks with at least 7/8 of the CUs
 *                  busy and at most one block per CU (224 .. 256 blocks: 128^3, 112 x 128^2), T >= 16: the whole reverse sweep of a
 *                  rollout backward (train_3drd.py:408) as ONE launch of resident workgroups with the adjoint state in LDS
 *                  (pi_adj3d_resident_kernel, round 6; dL/dh0 is the brick sweep's bit for bit; residency check, abort -> the
 *                  launch-per-step bricks and "persist_reset" as "tile_persist", which also gates it; 21 MB of the per-device
 *                  granule scratch); 0: one brick launch per step; 2: every grid of whole blocks that fits the device (tests)
 *   "peer_wire_us"  MEASUREMENT AID, default 0: every put over the peer mailboxes holds its arrival flag back this many
 *                  microseconds -- a stand-in for the link time of an xGMI hop when a ring runs through one device's own mailbox
 *                  (examples/slab_delay_ring.cpp, profiles/r06_slab_injected_wire.txt)
 *   "persist_reset"  (any value) re-arm the persistent sweep after an abort
 * Returns 0, or PERCNN_PI_EINVAL for an unknown key / bad value. */
int percnn_pi_set_option(const char *key, long value);

/* State of the persistent tile sweep in this process: info[0] launches so far, info[1] launches that aborted, info[2] 1 if the
 * current device has been switched to the launch-per-group path by an abort, info[3] / info[4] group / tile of the last abort,
 * info[5] the state word of the most recent launch as the device left it (0 not started yet, 1 every workgroup resident, 2
 * aborted; -1: no launch yet).  `info` holds at least 8 longs. */
int percnn_pi_persist_status(long *info);
/* For hosts that set "persist_handshake" = 0 (no wait at enqueue time) and pass the outputs of a rollout to code outside this
 * library: synchronises `stream`, then returns 0, or PERCNN_PI_EASYNC -- once -- if a resident launch aborted since the last
 * entry point looked (its outputs are invalid: re-run the call; the device is on the launch-per-group path from then on until
 * "persist_reset").  The library's own entry points make the same check when they are entered. */
int percnn_pi_persist_fence(void *stream);
/* Diagnostics for tests of that abort path: `blocks` workgroups that each hold `lds_bytes` of a CU's LDS for `ms` milliseconds
 * on `stream` (a stand-in for "another kernel holds whole CUs"). */
int percnn_pi_debug_hog(int blocks, int lds_bytes, int ms, void *stream);

/* ---- one Pi-block step ------------------------------------------------------------------
 * Replaces RCNNCell.forward(h) (2dgs:105-121, 3dgs:123-139, lo:98-112): periodic pad,
 * W_laplace conv of each species, three parallel 1x1 convs 2->hc multiplied element-wise,
 * 1x1 conv hc->1, Euler update.  h, out: [2][*S]. */
int percnn_pi_step_fwd_f32(const float *h, float *out, const float *params, int hc, int ndim,
                           const int64_t *shape, void *stream);
int percnn_pi_step_fwd_f64(const double *h, double *out, const double *params, int hc, int ndim,
                           const int64_t *shape, void *stream);

/* Adjoint of one step; replaces the autograd backward of the ops above (triggered at
 * 2dgs:407, 3dgs:408, lo:373).
 *   h          state the step was applied to                      [2][*S]
 *   g_out      dL/d(step output)                                  [2][*S]
 *   g_inject   optional dL/dh arriving from other consumers of h, added to the result (may be NULL)
 *   g_in       dL/dh (output)                                     [2][*S]
 *   param_grad ACCUMULATED (+=) gradient block, double[percnn_pi_param_count(hc)]
 *   workspace  percnn_pi_bwd_workspace_bytes(...) bytes of device scratch */
int percnn_pi_step_bwd_f32(const float *h, const float *g_out, const float *g_inject, float *g_in,
                           double *param_grad, void *workspace, size_t workspace_bytes,
                           const float *params, int hc, int ndim, const int64_t *shape, void *stream);
int percnn_pi_step_bwd_f64(const double *h, const double *g_out, const double *g_inject, double *g_in,
                           double *param_grad, void *workspace, size_t workspace_bytes,
                           const double *params, int hc, int ndim, const int64_t *shape, void *stream);

/* ---- T-step rollout ---------------------------------------------------------------------
 * Replaces the time loop of RCNN.forward() (2dgs:162-190, 3dgs:186-214, lo:169-218).
 * traj: [T+1][2][*S]; frame 0 holds the initial state on entry, frames 1..T are written
 * (frame k = cell applied k times), i.e. exactly torch.cat(tuple(outputs), dim=0) of the
 * reference's callers (2dgs:394). */
int percnn_pi_rollout_fwd_f32(float *traj, const float *params, int hc, int ndim,
                              const int64_t *shape, int T, void *stream);
int percnn_pi_rollout_fwd_f64(double *traj, const double *params, int hc, int ndim,
                              const int64_t *shape, int T, void *stream);

/* Backward of the rollout over a trajectory produced by percnn_pi_rollout_fwd_*: a sequential
 * reverse sweep t = T..1 that stores the adjoint trajectory in `workspace`, then ONE time-parallel
 * reduction over all (step, point) pairs for the branch-weight gradients.
 *   workspace   percnn_pi_rollout_bwd_workspace_bytes(...) bytes of device scratch
 *   g_traj      dL/dtraj, [T+1][2][*S]
 *   frame_mask  optional HOST array of T+1 bytes; frame k of g_traj is read only where
 *               frame_mask[k] != 0 (sparse losses such as the strided data loss 2dgs:397-402);
 *               NULL = every frame carries gradient
 *   g_h0        dL/d(initial state) (output)                      [2][*S]
 *   param_grad  ACCUMULATED gradient block, double[percnn_pi_param_count(hc)] */
int percnn_pi_rollout_bwd_f32(const float *traj, const float *g_traj, const unsigned char *frame_mask,
                              float *g_h0, double *param_grad, void *workspace, size_t workspace_bytes,
                              const float *params, int hc, int ndim, const int64_t *shape, int T,
                              void *stream);
int percnn_pi_rollout_bwd_f64(const double *traj, const double *g_traj, const unsigned char *frame_mask,
                              double *g_h0, double *param_grad, void *workspace, size_t workspace_bytes,
                              const double *params, int hc, int ndim, const int64_t *shape, int T,
                              void *stream);

/* ---- slab-decomposed step (one rank of a 1-D domain decomposition along spatial axis 0) ----
 * No reference counterpart: the reference is single-GPU (each script pins one device, 2dgs:14).
 * All state-shaped arrays are LOCAL slabs with `halo` planes (even, >= 2) on each side of axis 0:
 *   [2][n0 + 2*halo][rest];  `shape` = LOCAL interior shape {n0, ...}; other axes stay periodic.
 * Forward: planes [skip, n0+2*halo-skip) of `h` must be valid (filled by the caller's halo
 * exchange or by a previous call with skip-2); planes [skip+2, n0+2*halo-skip-2) of `out` are
 * written.  skip = 0, 2, ..., halo-2 lets the caller take halo/2 steps per exchange (wide halos,
 * the outer planes being recomputed redundantly); the last of them writes exactly the interior.
 * Backward: `g_out` needs 2 valid planes next to the interior; `h`, `g_inject` (nullable) and
 * `g_in` are touched on the interior planes [halo, halo+n0) only; param_grad sums over them. */
int percnn_pi_slab_step_fwd_f32(const float *h, float *out, const float *params, int hc, int ndim,
                                const int64_t *shape, int halo, int skip, void *stream);
int percnn_pi_slab_step_fwd_f64(const double *h, double *out, const double *params, int hc, int ndim,
                                const int64_t *shape, int halo, int skip, void *stream);
int percnn_pi_slab_step_bwd_f32(const float *h, const float *g_out, const float *g_inject, float *g_in,
                                double *param_grad, void *workspace, size_t workspace_bytes,
                                const float *params, int hc, int ndim, const int64_t *shape, int halo,
                                int flags, void *stream);
int percnn_pi_slab_step_bwd_f64(const double *h, const double *g_out, const double *g_inject, double *g_in,
                                double *param_grad, void *workspace, size_t workspace_bytes,
                                const double *params, int hc, int ndim, const int64_t *shape, int halo,
                                int flags, void *stream);

/* 'same' 5x5x5 cross-correlation 8 -> 8 channels with zero padding on a [8][D][H][W] float32 field -- the contraction
 * of the 3D IC generator's second layer (ConvTranspose3d(8, 8, 5, padding=2), train_3drd.py:45-52): its forward and its
 * input gradient are this operation with differently arranged weights.
 *   out[co](p) = bias[co] + sum_{ci, dz, dy, dx} weights[(((ci*5 + dz)*5 + dy)*5 + dx)*8 + co] * in[ci](p + (dz,dy,dx) - 2)
 * bias may be NULL; shape = {D, H, W}; in and out must not alias. */
int percnn_pi_conv3d_k5c8_f32(const float* in, float* out, const float* weights, const float* bias, const int64_t* shape,
                              void* stream);

/* weight gradient of the same contraction: g_weights[(((ci*5+dz)*5+dy)*5+dx)*8 + co] = sum_p in[ci](p+d-2) * g_out[co](p)
 * (overwritten); workspace: percnn_pi_conv3d_k5c8_wgrad_workspace_bytes() bytes of device memory. */
size_t percnn_pi_conv3d_k5c8_wgrad_workspace_bytes(void);
int percnn_pi_conv3d_k5c8_wgrad_f32(const float* in, const float* g_out, float* g_weights, void* workspace,
                                    size_t workspace_bytes, const int64_t* shape, void* stream);

/* ---- native slab rollouts (multi-GPU): the whole T-step loop, halo exchanges included, in ONE call ---------------
 * The ring is described by plain function pointers so that the library needs no link-time dependency on RCCL: the host
 * side passes the addresses of ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv of the librccl it already uses
 * (percnn_amd.slab does this with ctypes for the librccl PyTorch loaded) plus its communicator and neighbour ranks --
 * or a percnn_pi_peer_ring (below), in which case the library needs nothing from RCCL at all.
 * ring == NULL: single rank, the periodic wrap is done with device-to-device copies.
 * overlap != 0: the step that produces a frame about to be exchanged computes the two faces first, the exchange runs
 * on an internal side stream (ordered by events) while the planes in between are computed.
 * Layout: local padded trajectories [T+1][2][n0_local + 2*halo][...]; shape = local interior shape. */
/* Second transport: PEER MAILBOXES (csrc/pi_peer.h).  xGMI peers are load/store addressable, so a face can travel as plain
 * stores into a mailbox that lives in the neighbour's fine-grained device memory, announced by an epoch flag -- two small
 * kernels per exchange on the compute stream instead of one ncclGroup call (~19-25 us each on MI355X).  Every rank
 * allocates one mailbox (percnn_pi_peer_box_alloc), exports its hipIpc handle, maps the mailboxes of its two ring
 * neighbours (percnn_pi_peer_box_open; a rank that is its own neighbour passes its own mailbox) and fills this struct;
 * the library advances `epoch` by one per exchange -- all ranks of a ring must issue the same sequence of exchanges.
 * slot_bytes >= 2 species x width planes of the widest exchange (rounded up to 16 B per species). */
typedef struct percnn_pi_peer_ring {
    void* my_box;               /* this rank's mailbox */
    void* prev_box;             /* mapped mailbox of the previous / next rank on the ring */
    void* next_box;
    size_t slot_bytes;          /* the capacity all three mailboxes were allocated with */
    uint64_t epoch;             /* exchanges issued so far (in/out) */
    uint64_t timeout_ticks;     /* bounded wait of a take, in 10 ns ticks; 0 = default: PERCNN_PEER_TIMEOUT_S seconds
                                 * (environment, default 300).  A take that times out records the exchange number in the
                                 * mailbox (percnn_pi_peer_box_status) and fills its halo planes with NaNs; so does every
                                 * later take of that mailbox. */
} percnn_pi_peer_ring;

typedef struct percnn_pi_halo_ring {
    void* comm;                 /* ncclComm_t */
    int prev, next;             /* neighbour ranks on the ring (periodic) */
    int dtype_f32, dtype_f64;   /* ncclFloat32 / ncclFloat64 enum values of that RCCL */
    int (*group_start)(void);
    int (*group_end)(void);
    int (*send)(const void* buf, size_t count, int dtype, int peer, void* comm, void* stream);
    int (*recv)(void* buf, size_t count, int dtype, int peer, void* comm, void* stream);
    percnn_pi_peer_ring* peer;  /* non-NULL: exchange through the peer mailboxes, the RCCL members above are not used */
    void* stage;                /* optional device scratch of stage_bytes for the RCCL path: both species of a face are packed into */
    size_t stage_bytes;         /* ONE message per direction (2 sends + 2 receives per exchange instead of 4 + 4); needs
                                 * 8 x width x plane elements; NULL / too small: per-species messages straight from the slab */
} percnn_pi_halo_ring;

/* diagnostics (host only, no device work): the block decomposition the direct kernels would use for `shape` under `options`
 * ("key=value,..." or NULL): out[6] = {log2 lanes along x (-1 = flat), x blocks per row, row groups per plane, virtual blocks,
 * planes per pass of the adjoint, workgroup size} */
int percnn_pi_debug_blockmap(int ndim, const int64_t* shape, int elem_size, const char* options, int* out);
/* Host-only: which kernel family a rollout of this problem takes (the library's own dispatch rules, for 16-byte-aligned
 * buffers).  out[15] = {forward family, adjoint family, 1 if the parameter gradients are reduced inside the sweep launches,
 * time steps per forward launch, per adjoint launch, planes per pass forward, adjoint, lanes per brick workgroup or 0,
 * 2D tile width, tile height, lanes per tile workgroup of the adjoint sweep (0: no tile kernels), the same three of the
 * forward, bit 0: the tile sweep of a long rollout without frame mask runs as ONE launch of resident workgroups (option
 * tile_persist; asks the current device for its CU count -- 0 without a device), bit 1: so does its forward (fwd_persist)};
 * families: 0 direct step kernels,
 * 1 2D tile kernels, 2 3D plane streaming, 3 3D brick kernels, 4 advective block.  (The lanes are those of the launch-per-group
 * tile kernels; the resident launches of the 8- and 16-row regimes run twice as many on half-strips: fwd_small_half, adj_small_half.) */
int percnn_pi_debug_plan(int hc, int ndim, const int64_t* shape, int elem_size, const char* options, int* out);

size_t percnn_pi_peer_box_bytes(size_t slot_bytes);                 /* size of a mailbox allocation */
int percnn_pi_peer_box_alloc(void** box, size_t slot_bytes);        /* fine-grained device memory on the current device, zeroed */
int percnn_pi_peer_box_free(void* box);
int percnn_pi_peer_box_export(void* box, void* handle64);           /* hipIpcMemHandle_t (64 bytes) of a mailbox */
int percnn_pi_peer_box_open(const void* handle64, void** mapped);   /* map a neighbour's mailbox (another process) */
int percnn_pi_peer_box_close(void* mapped);
int percnn_pi_peer_box_status(const void* box, uint64_t* error_epoch, void* stream);   /* synchronises `stream`; 0 = no take timed out */
/* one ring exchange of `width` planes per side of a local slab [2][n0 + 2*halo][...] (shape = local interior shape) */
int percnn_pi_peer_exchange_f32(float* slab, int ndim, const int64_t* shape, int halo, int width,
                                percnn_pi_peer_ring* ring, void* stream);
int percnn_pi_peer_exchange_f64(double* slab, int ndim, const int64_t* shape, int halo, int width,
                                percnn_pi_peer_ring* ring, void* stream);

size_t percnn_pi_halo_ring_bytes(void);   /* sizeof(percnn_pi_halo_ring) of the library build: bindings check their layout */
/* overlap: bit 0 = faces first, exchange on a side stream; bit 1 (ring == NULL only) = periodic wrap by face copies into the
 * halo planes instead of by index (option "slab_local_index") */
int percnn_pi_slab_rollout_fwd_f32(float* traj, const float* params, int hc, int ndim, const int64_t* shape, int halo,
                                   int T_steps, const percnn_pi_halo_ring* ring, int overlap, void* stream);
int percnn_pi_slab_rollout_fwd_f64(double* traj, const double* params, int hc, int ndim, const int64_t* shape, int halo,
                                   int T_steps, const percnn_pi_halo_ring* ring, int overlap, void* stream);
/* adj: caller-provided local adjoint trajectory (same layout as traj, contents irrelevant on entry; adj[0] holds
 * dL/d(frame 0) on return); g_traj: dL/dtraj in the same padded layout (halo planes ignored; only with option "slab_wide_adjoint" = 1 -- one
 * exchange per two adjoint steps over an RCCL ring, halo >= 4 -- they serve as receive buffers and hold the neighbours'
 * values on return); param_grad: double[np],
 * ACCUMULATED (local sums of this rank; the caller all-reduces them); workspace: percnn_pi_bwd_workspace_bytes(). */
int percnn_pi_slab_rollout_bwd_f32(const float* traj, const float* g_traj, float* adj, double* param_grad,
                                   void* workspace, size_t workspace_bytes, const float* params, int hc, int ndim,
                                   const int64_t* shape, int halo, int T_steps, const percnn_pi_halo_ring* ring,
                                   int overlap, void* stream);
int percnn_pi_slab_rollout_bwd_f64(const double* traj, const double* g_traj, double* adj, double* param_grad,
                                   void* workspace, size_t workspace_bytes, const double* params, int hc, int ndim,
                                   const int64_t* shape, int halo, int T_steps, const percnn_pi_halo_ring* ring,
                                   int overlap, void* stream);

/* Plane-range variants of the slab steps, for communication / computation overlap: compute only the padded plane
 * indices [lo, hi) (2 <= lo < hi <= n0_local + 2*halo - 2) of the output -- e.g. first the faces a neighbour is waiting
 * for, then, while they travel, the planes in between.  The union of the ranges of one step must be what the
 * un-split call computes; values are bit-identical to it.  Backward: split launches of ONE sweep share their gradient
 * sums through the workspace (PERCNN_PI_NO_RESET / PERCNN_PI_NO_FINISH). */
int percnn_pi_slab_step_fwd_range_f32(const float* h, float* h_next, const float* params, int hc, int ndim,
                                      const int64_t* shape, int halo, int lo, int hi, void* stream);
int percnn_pi_slab_step_fwd_range_f64(const double* h, double* h_next, const double* params, int hc, int ndim,
                                      const int64_t* shape, int halo, int lo, int hi, void* stream);
int percnn_pi_slab_step_bwd_range_f32(const float* h, const float* g_out, const float* g_inject, float* g_in,
                                      double* param_grad, void* workspace, size_t workspace_bytes, const float* params,
                                      int hc, int ndim, const int64_t* shape, int halo, int lo, int hi, int flags,
                                      void* stream);
int percnn_pi_slab_step_bwd_range_f64(const double* h, const double* g_out, const double* g_inject, double* g_in,
                                      double* param_grad, void* workspace, size_t workspace_bytes, const double* params,
                                      int hc, int ndim, const int64_t* shape, int halo, int lo, int hi, int flags,
                                      void* stream);

/* Time-parallel gradient reduction over the INTERIOR of local slab trajectories (same padded layout):
 * traj frames 0..T-1 and adjoint frames 1..T ([T+1][2][n0+2*halo][rest] each), accumulated into param_grad.
 * Pairs with percnn_pi_slab_step_bwd_* called with PERCNN_PI_SWEEP_ONLY. */
int percnn_pi_slab_wgrad_f32(const float *traj, const float *adj, double *param_grad, void *workspace,
                             size_t workspace_bytes, const float *params, int hc, int ndim, const int64_t *shape,
                             int halo, int T, void *stream);
int percnn_pi_slab_wgrad_f64(const double *traj, const double *adj, double *param_grad, void *workspace,
                             size_t workspace_bytes, const double *params, int hc, int ndim, const int64_t *shape,
                             int halo, int T, void *stream);

/* ---- physics residual of the trajectory (the loss consumer next to the path) -------
 * Replaces loss_generator.get_phy_Loss (2dgs:270-329, 3dgs:287-323, lo:283-341): a 5x5(x5) Laplacian
 * convolution over all frames plus a permute/reshape/Conv1d time difference.  ONE launch, frame-parallel:
 *   resid[f][s](x) = coef_s * Lap(traj[f])_s + r_s(traj[f]) - (traj[f+1][s] - traj[f][s]) / dt,  f < nframes
 * `params` is a PRE-CONTRACTED block (36 entries, "hc = 0") holding the TRUE equation's constants
 * (e.g. Gray-Scott Du, Dv, f, k -- 2dgs:321-327), not the model's.  traj: [nframes+1][2][*S].
 * _bwd: g_state[f] = (d resid[f] / d traj[f])^T g_resid[f]; the part w.r.t. traj[f+1] is -g_resid[f]/dt,
 * pointwise, and is left to the caller. */
int percnn_pi_residual_fwd_f32(const float *traj, float *resid, const float *params, int ndim,
                               const int64_t *shape, int nframes, void *stream);
int percnn_pi_residual_fwd_f64(const double *traj, double *resid, const double *params, int ndim,
                               const int64_t *shape, int nframes, void *stream);
int percnn_pi_residual_bwd_f32(const float *traj, const float *g_resid, float *g_state, const float *params,
                               int ndim, const int64_t *shape, int nframes, void *stream);
int percnn_pi_residual_bwd_f64(const double *traj, const double *g_resid, double *g_state, const double *params,
                               int ndim, const int64_t *shape, int nframes, void *stream);

/* The residual LOSS and its gradient without the residual itself (round 3).  Replaces loss_gen (2dgs:340-353, 3dgs:333-346,
 * lo:343-357):  L = MSE(f_u, 0) + MSE(f_v, 0) over resid[0 .. nframes).  weighted != 0 reproduces the reference's padding
 * (2 cells low, 3 high -> an (N+1)^d evaluation grid in which index 0 of every axis counts twice):
 *   L = sum_{f,s,x} w(x) resid^2 / (nframes * prod(N_a + 1)),  w = 2^(number of zero coordinates of x);
 * weighted == 0: the plain mean over the periodic grid.  traj: [nframes+1][2][*S] at least.
 *   _sqloss     : loss_out[0] (device, compute type) = L.  One pass over the trajectory (8 / 16 B per point and frame), the
 *                 per-block sums in double, fixed-order final sum.  workspace >= percnn_pi_residual_sqloss_workspace_bytes().
 *   _sqloss_bwd : g_traj[0 .. nout_frames) = g_loss * dL/dtraj (nout_frames >= nframes + 1; frames the loss does not see are
 *                 zeroed), g_loss = DEVICE pointer to the upstream scalar gradient (NULL = 1).  Two launches: the scaled
 *                 residual into `scratch` (nframes frames), then its adjoint incl. the -G/dt part of the later frame.
 *                 (Was: residual, 6 element-wise autograd nodes over the whole trajectory, zero-fill, adjoint, a division and
 *                 a subtraction over the whole trajectory: lambda-omega 512^2 x 400 iteration 16.1 ms.) */
size_t percnn_pi_residual_sqloss_workspace_bytes(void);
int percnn_pi_residual_sqloss_f32(const float *traj, const float *params, int ndim, const int64_t *shape, int nframes,
                                  int weighted, float *loss_out, void *workspace, size_t workspace_bytes, void *stream);
int percnn_pi_residual_sqloss_f64(const double *traj, const double *params, int ndim, const int64_t *shape, int nframes,
                                  int weighted, double *loss_out, void *workspace, size_t workspace_bytes, void *stream);
int percnn_pi_residual_sqloss_bwd_f32(const float *traj, const float *g_loss, const float *params, int ndim,
                                      const int64_t *shape, int nframes, int nout_frames, int weighted, float *scratch,
                                      float *g_traj, void *stream);
int percnn_pi_residual_sqloss_bwd_f64(const double *traj, const double *g_loss, const double *params, int ndim,
                                      const int64_t *shape, int nframes, int nout_frames, int weighted, double *scratch,
                                      double *g_traj, void *stream);

/* ---- per-call tuning overrides -----------------------------------------------------------------------------------
 * The four calls above that training / inference loops issue, with an `options` string "key=value,key=value" (keys of
 * percnn_pi_set_option; NULL or "" = the process defaults).  The overrides apply to THIS call only and are never
 * written back, so two models (or two threads) can run with different settings.  Returns PERCNN_PI_EINVAL for a
 * malformed string, an unknown key or a bad value.  Same reference calls as their namesakes: RCNNCell.forward
 * (2dgs:105-121), its autograd backward (2dgs:407), RCNN.forward's loop (2dgs:162-190) and its backward. */
int percnn_pi_step_fwd_opt_f32(const float *h, float *out, const float *params, int hc, int ndim,
                               const int64_t *shape, const char *options, void *stream);
int percnn_pi_step_fwd_opt_f64(const double *h, double *out, const double *params, int hc, int ndim,
                               const int64_t *shape, const char *options, void *stream);
int percnn_pi_step_bwd_opt_f32(const float *h, const float *g_out, const float *g_inject, float *g_in,
                               double *param_grad, void *workspace, size_t workspace_bytes, const float *params,
                               int hc, int ndim, const int64_t *shape, const char *options, void *stream);
int percnn_pi_step_bwd_opt_f64(const double *h, const double *g_out, const double *g_inject, double *g_in,
                               double *param_grad, void *workspace, size_t workspace_bytes, const double *params,
                               int hc, int ndim, const int64_t *shape, const char *options, void *stream);
/* percnn_pi_rollout_bwd_opt_* with the gradient of the LAST frame in a buffer of its own: frame T of `g_traj` is never read,
 * `g_top` ([2][*S]) is used instead (frame_mask, if given, must select frame T).  For callers whose per-frame gradients are
 * slices of one buffer except the newest one -- what autograd hands a node that produced T consecutive frames of a step loop
 * (train_2drd.py:169-188) when the last of them also feeds the next node. */
int percnn_pi_rollout_bwd_top_f32(const float *traj, const float *g_traj, const float *g_top, const unsigned char *frame_mask,
                                  float *g_h0, double *param_grad, void *workspace, size_t workspace_bytes,
                                  const float *params, int hc, int ndim, const int64_t *shape, int T, const char *options,
                                  void *stream);
int percnn_pi_rollout_bwd_top_f64(const double *traj, const double *g_traj, const double *g_top, const unsigned char *frame_mask,
                                  double *g_h0, double *param_grad, void *workspace, size_t workspace_bytes,
                                  const double *params, int hc, int ndim, const int64_t *shape, int T, const char *options,
                                  void *stream);
/* One adjoint step whose parameter-gradient sums STAY in the workspace's partial rows (a reference-style loop,
 * train_2drd.py:169-188, differentiated by autograd node by node: T calls of this, ONE percnn_pi_bwd_rows_finish_* at the end
 * instead of a reset + reduction launch per step).  flags: PERCNN_PI_NO_RESET when the rows already hold sums of earlier calls
 * of the same pass (the first call of a pass zeroes them).  The workspace must be the same buffer, and (hc, shape) the same
 * problem, for all calls of one pass.  Arguments otherwise as percnn_pi_step_bwd_*. */
int percnn_pi_step_bwd_rows_f32(const float *h, const float *g_out, const float *g_inject, float *g_in, void *workspace,
                                size_t workspace_bytes, const float *params, int hc, int ndim, const int64_t *shape,
                                int flags, void *stream);
int percnn_pi_step_bwd_rows_f64(const double *h, const double *g_out, const double *g_inject, double *g_in, void *workspace,
                                size_t workspace_bytes, const double *params, int hc, int ndim, const int64_t *shape,
                                int flags, void *stream);
/* param_grad (double[percnn_pi_param_count(hc)]) += the sums those calls left in the rows (one launch). */
int percnn_pi_bwd_rows_finish_f32(void *workspace, size_t workspace_bytes, int hc, int ndim, const int64_t *shape,
                                  double *param_grad, void *stream);
int percnn_pi_bwd_rows_finish_f64(void *workspace, size_t workspace_bytes, int hc, int ndim, const int64_t *shape,
                                  double *param_grad, void *stream);
int percnn_pi_rollout_fwd_opt_f32(float *traj, const float *params, int hc, int ndim, const int64_t *shape,
                                  int T, const char *options, void *stream);
int percnn_pi_rollout_fwd_opt_f64(double *traj, const double *params, int hc, int ndim, const int64_t *shape,
                                  int T, const char *options, void *stream);
int percnn_pi_rollout_bwd_opt_f32(const float *traj, const float *g_traj, const unsigned char *frame_mask,
                                  float *g_h0, double *param_grad, void *workspace, size_t workspace_bytes,
                                  const float *params, int hc, int ndim, const int64_t *shape, int T,
                                  const char *options, void *stream);
int percnn_pi_rollout_bwd_opt_f64(const double *traj, const double *g_traj, const unsigned char *frame_mask,
                                  double *g_h0, double *param_grad, void *workspace, size_t workspace_bytes,
                                  const double *params, int hc, int ndim, const int64_t *shape, int T,
                                  const char *options, void *stream);

/* ---- squared-error losses differentiated INSIDE the sweep (round 3) -------------------------------------------------------
 * Replaces, for the reference's dense squared-error losses (train_2drd.py:397-407 MSE against data; SURVEY 8d mean(traj^2)),
 * the sequence  loss = f(torch.cat(outputs)) ; loss.backward()  -- which materialises dL/dtraj (as large as the trajectory)
 * only for the sweep to read it back -- by a sweep that forms  dL/dh_t = a * (h_t - target_t)  from the state it loads anyway
 * (24 instead of 32 bytes per point and step; no dL/dtraj buffer).
 *   L = scale' * sum_{t : frame_mask[t]} sum_x (traj_t - target_t)^2 ,   a = 2 * scale' (pass scale = a), times *dev_scale
 *   (device scalar of the compute type, nullable: the gradient autograd hands the loss -- no host synchronisation).
 * target == NULL: target = 0 (mode 1: the sweep reads nothing extra).  frame_mask: HOST array of T_steps + 1 bytes, NULL =
 * every frame.  Workspace, params, options, error codes as percnn_pi_rollout_bwd_opt_*; PERCNN_PI_EINVAL also for block
 * kinds without the in-kernel form (advective blocks): materialise the gradient and call percnn_pi_rollout_bwd_* then. */
int percnn_pi_rollout_bwd_sqerr_f32(const float *traj, const float *target, const unsigned char *frame_mask, double scale,
                                    const float *dev_scale, float *g_h0, double *param_grad, void *workspace,
                                    size_t workspace_bytes, const float *params, int hc, int ndim, const int64_t *shape,
                                    int T_steps, const char *options, void *stream);
int percnn_pi_rollout_bwd_sqerr_f64(const double *traj, const double *target, const unsigned char *frame_mask, double scale,
                                    const double *dev_scale, double *g_h0, double *param_grad, void *workspace,
                                    size_t workspace_bytes, const double *params, int hc, int ndim, const int64_t *shape,
                                    int T_steps, const char *options, void *stream);
/* The loss value itself: out[0] = scale * sum_{f < nframes : frame_mask[f]} sum_x (traj_f - target_f)^2 in ONE streaming
 * pass (float64 accumulation); workspace >= 8 KiB, 8-byte aligned; one launch per run of consecutive selected frames (any number of runs). */
int percnn_pi_traj_sqerr_f32(const float *traj, const float *target, const unsigned char *frame_mask, int nframes, int ndim,
                             const int64_t *shape, double scale, float *out, void *workspace, size_t workspace_bytes,
                             void *stream);
int percnn_pi_traj_sqerr_f64(const double *traj, const double *target, const unsigned char *frame_mask, int nframes, int ndim,
                             const int64_t *shape, double scale, double *out, void *workspace, size_t workspace_bytes,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PERCNN_PI_H */
