/* C-ABI of the Stage-1 Pi-block -- part of libpercnn_pi.so.
 *
 * Replaces, for one time step / a T-step rollout and its backward, the body of
 *   RCNNCell.forward   DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/rcnn_Burgers_[...].py:143-178
 *                      DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-1/rcnn_LO_[...].py:141-172
 *   RCNN.forward loop  rcnn_Burgers_[...].py:277-303 (T sequential cell calls) and the autograd backward of both
 * i.e.  h_next[s] = h[s] + dt * ( coef_s * Lap(h[s]) + Wh4_s( Wh1_s(h) * Wh2_s(h) * Wh3_s(h) ) ),
 * Wh{1,2,3}_s = 5x5 periodic convolutions 2 -> 16 with bias, Wh4_s = 1x1 convolution 16 -> 1 with bias.
 * float32 only (both reference scripts set the default dtype to float32), 2D periodic grids, 2 species.
 *
 * Parameter block (device array of PERCNN_PI_S1_PARAMS floats):
 *   [0] dt   [1] coef_u = nu_up*sigmoid(CA)   [2] coef_v   [3] Laplacian centre tap (already / dx^2)
 *   [4+i]  Laplacian tap at offset {-2,-1,+1,+2}[i] along axis 0 (rows),  [8+i] along axis 1,  [12..15] unused
 *   [16 + ((s*3+k)*16 + j)*52 + kk]   branch k (0..2) of species s (0 = u, 1 = v), hidden channel j (0..15):
 *        kk = c*25 + dy*5 + dx  ->  Wh{k+1}_s.weight[j, c, dy, dx]   (cross-correlation, tap offset (dy-2, dx-2))
 *        kk = 50                ->  Wh{k+1}_s.bias[j]
 *        kk = 51                ->  0 (padding to a multiple of the MFMA K = 4)
 *   [16+4992 + s*16 + j]  Wh4_s.weight[0, j, 0, 0]        [16+4992+32 + s]  Wh4_s.bias[0]
 * The gradient block (double[PERCNN_PI_S1_PARAMS]) uses the same indexing; slots 0 and 3..15 and kk = 51 stay 0.
 *
 * Conventions as in percnn_pi.h: caller owns all buffers, planar [2][H][W] states, asynchronous on `stream`
 * (a hipStream_t), no allocation / host synchronisation inside, returns 0 or hipError_t / PERCNN_PI_E*.
 */
#ifndef PERCNN_PI_STAGE1_H
#define PERCNN_PI_STAGE1_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PERCNN_PI_S1_PARAMS 5042

/* number of floats in the parameter block (= PERCNN_PI_S1_PARAMS) */
size_t percnn_pi_s1_param_count(void);

/* tuning / diagnostics (results stay within the documented tolerance): "etile" = 1 (default) hands the input-gradient
 * contributions between consecutive launches of the backward sweep as 8x8 footprint tiles when H and W are multiples
 * of 4, 0 = always per-tap planes;  "skip_wgrad" = 1 runs the adjoint sweep only (parameter gradients come back as
 * zeros; used to time the sweep kernel alone);  "persist" = 1 (default): rollouts of >= "persist_min_steps" (8) steps on
 * grids of whole 4x4 patches whose (patch, species) tasks all fit on the device at once run as ONE launch of resident waves
 * each way (round 5; results bit for bit those of one launch per step; the residency guard, the handshake on the roll
 * call, PERCNN_PI_EASYNC and the abort -> launch-per-step fallback are those of percnn_pi.h's resident launches, option
 * keys persist_* there), 0 = one launch per step;  "pause_fwd" / "pause_adj" = s_sleep units between publishing a step's
 * granules and requesting the neighbours'.  Returns PERCNN_PI_EINVAL for an unknown key. */
int percnn_pi_s1_set_option(const char* key, long value);

/* one time step; shape = {H, W}, H and W >= 8;  h and h_next: [2][H][W], must not alias */
int percnn_pi_s1_step_fwd_f32(const float* h, float* h_next, const float* params, const int64_t* shape, void* stream);

/* T steps: traj [T+1][2][H][W], frame 0 holds the initial state, frames 1..T are written */
int percnn_pi_s1_rollout_fwd_f32(float* traj, const float* params, const int64_t* shape, int T_steps, void* stream);

/* workspace of the rollout backward (adjoint trajectory + per-tap scatter planes + gradient partials) */
size_t percnn_pi_s1_rollout_bwd_workspace_bytes(const int64_t* shape, int T_steps);

/* backward of the T-step rollout:  g_traj = dL/dtraj [T+1][2][H][W] (frame_mask: host bytes, T+1 entries, 0 = frame
 * carries no gradient and is not read; NULL = all frames);  g_h0 [2][H][W] = dL/d(frame 0);
 * param_grad: double[PERCNN_PI_S1_PARAMS] (overwritten) */
int percnn_pi_s1_rollout_bwd_f32(const float* traj, const float* g_traj, const unsigned char* frame_mask, float* g_h0,
                                 double* param_grad, void* workspace, size_t workspace_bytes, const float* params,
                                 const int64_t* shape, int T_steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
