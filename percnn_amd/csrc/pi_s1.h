// Stage-1 Pi-block: three 5x5 conv branches 2 -> 16 per species, Hadamard product, 1x1
// contraction, FD Laplacian, explicit Euler.  float32, 2D periodic.
//   reference: DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-1/rcnn_Burgers_[...].py:54-178
//              DataDrivenDiscoveryOfPDEs/2D_Lambda_Omega_eqn/stage-1/rcnn_LO_[...].py:53-172
//
// Here the branch evaluation IS a dense contraction (K = 2*25 taps + bias = 51 -> 52), so it runs on the
// matrix cores: v_mfma_f32_16x16x4_f32 with  M = 16 hidden channels, N = 16 grid points (a 4x4 patch),
// K = 4 taps per instruction.  Orientation "channels x points":
//   A (weights)  lane l holds W[j = l&15][kk = 4q + (l>>4)]          -- resident in VGPRs for the whole launch
//   B (im2col)   lane l holds h_c(y+dy, x+dx) of point l&15, tap kk = 4q + (l>>4)   -- one ds_read_b32 from the
//                wave's 8x8x2 window in LDS (periodic wrap resolved once, when the window is staged)
//   D            lane l holds channels j = 4*(l>>4) + r (r = 0..3) of point l&15
// so the product of the three branches, the Wh4 contraction and the Euler update are per-lane VALU work plus two
// cross-lane adds.  One wave = one (patch, species) task; a wave needs no other wave => no workgroup barriers.
// f32 MFMA is an exact k-ordered fmaf chain, so the forward is bit-identical to oracle/pi_oracle.c's
// pi_oracle_s1_step_fwd_f32.
#pragma once
#include <hip/hip_runtime.h>
#include "pi_device.h"

namespace pi {
namespace s1 {

using f4 = __attribute__((ext_vector_type(4))) float;

constexpr int HC = 16;
constexpr int KK = 52;                 // 50 taps + bias + 1 zero pad
constexpr int NKS = KK / 4;            // MFMA k-steps per branch
constexpr int OFF_W = 16;
constexpr int OFF_W4 = OFF_W + 6 * HC * KK;
constexpr int OFF_B4 = OFF_W4 + 2 * HC;
constexpr int NP = OFF_B4 + 2;         // 5042
constexpr int WIN = 8;                 // 4x4 patch + radius-2 halo
constexpr int WAVES = 4;               // independent waves per workgroup

struct Geom {
    int H, W;
    int px;        // patches per row = ceil(W / 4)
    int npatch;    // ceil(H/4) * ceil(W/4)
    long n;        // H * W
};

__device__ __forceinline__ int wrap1(int i, int n)
{
    i = i < 0 ? i + n : i;
    return i >= n ? i - n : i;
}

// writes by some lanes of this wave must be visible to reads by other lanes of the same wave
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float xor_add(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

// per-lane LDS offset (in floats, inside one wave's window [2][8][8]) of the B operand of k-step q
__device__ __forceinline__ int tap_offset(int q, int grp, int py, int px)
{
    int kk = 4 * q + grp;
    kk = kk < 50 ? kk : 0;                         // bias / pad slots never read the window
    const int c = kk >= 25 ? 1 : 0, rem = kk - 25 * c;
    const int dy = rem / 5, dx = rem - 5 * dy;
    return c * (WIN * WIN) + (py + dy) * WIN + (px + dx);
}

// stage the 8x8 window (both species) of the patch at (y0, x0): 128 values, 2 per lane
struct Window {
    float v[2];
    __device__ __forceinline__ void load(const float* __restrict__ h, const Geom& g, int y0, int x0, int lane)
    {
        const int wy = lane >> 3, wx = lane & 7;
        const int row = wrap1(y0 + wy - 2, g.H), col = wrap1(x0 + wx - 2, g.W);
        v[0] = h[(long)row * g.W + col];
        v[1] = h[g.n + (long)row * g.W + col];
    }
    __device__ __forceinline__ void store(float* win, int lane) const
    {
        win[lane] = v[0];
        win[WIN * WIN + lane] = v[1];
    }
};

// branch pre-activations of species s at the 16 points of the patch: acc[k] in D layout
__device__ __forceinline__ void branches(const float (&a)[3][NKS], const float* win, const int (&toff)[NKS], int grp,
                                         f4 (&acc)[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NKS; ++q) {
        float b = win[toff[q]];
        if (q == NKS - 1) b = grp == 2 ? 1.0f : (grp == 3 ? 0.0f : b);      // kk = 50: bias column, kk = 51: pad
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k][q], b, acc[k], 0, 0, 0);
    }
}

__device__ __forceinline__ void load_branch_weights(const float* __restrict__ P, int s, int lane, float (&a)[3][NKS])
{
    const int j = lane & 15, grp = lane >> 4;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < NKS; ++q) a[k][q] = P[OFF_W + ((s * 3 + k) * HC + j) * KK + 4 * q + grp];
}

// star Laplacian of species plane `w` (a [8][8] window) at patch point (py, px); oracle order: centre, axis 0, axis 1
template <int FLIP>
__device__ __forceinline__ float win_star(const float* w, const float* __restrict__ P, int py, int px)
{
    const int c = (py + 2) * WIN + (px + 2);
    float lap = P[P_C0] * w[c];
    constexpr int offs[4] = {-2, -1, 1, 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) lap = fma_(P[P_TAPS + i], w[c + FLIP * offs[i] * WIN], lap);
#pragma unroll
    for (int i = 0; i < 4; ++i) lap = fma_(P[P_TAPS + 4 + i], w[c + FLIP * offs[i]], lap);
    return lap;
}

// ------------------------------------------------------------------------------------------------
// forward: out = h + dt * (coef_s * Lap(h_s) + b4 + sum_j w4_j * prod_k (W_kj * h + b_kj))
// grid = (workgroups, 2 species); each wave walks patches with stride gridDim.x * WAVES
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void s1_fwd_kernel(const float* __restrict__ h, float* __restrict__ out,
                                                            const float* __restrict__ P, Geom g)
{
    __shared__ float lds[WAVES][2 * WIN * WIN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];

    int patch = blockIdx.x * WAVES + wv;
    const int stride = gridDim.x * WAVES;
    if (patch >= g.npatch) return;

    Window wnd;
    wnd.load(h, g, (patch / g.px) * 4, (patch % g.px) * 4, lane);

    float a[3][NKS];
    load_branch_weights(P, s, lane, a);
    int toff[NKS];
#pragma unroll
    for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);
    float w4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w4[r] = P[OFF_W4 + s * HC + 4 * grp + r];
    const float b4 = P[OFF_B4 + s], coef = P[P_COEF + s], dt = P[P_DT];

    for (; patch < g.npatch; patch += stride) {
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        wave_sync();                                   // earlier reads of the window are done
        wnd.store(win, lane);
        wave_sync();
        if (patch + stride < g.npatch) {               // next window travels while the matrix cores work
            const int pn = patch + stride;
            wnd.load(h, g, (pn / g.px) * 4, (pn % g.px) * 4, lane);
        }
        f4 acc[3];
        branches(a, win, toff, grp, acc);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fma_(w4[r], (acc[0][r] * acc[1][r]) * acc[2][r], t);
        t = xor_add(t, 16);                            // (c0 + c1), (c2 + c3)
        t = xor_add(t, 32);                            // + the other pair
        const float rr = t + b4;
        const float* ws = win + s * (WIN * WIN);
        const float lap = win_star<1>(ws, P, py, px);
        const float res = coef * lap + rr;
        const float upd = res * dt;
        const int y = y0 + py, x = x0 + px;
        if (grp == 0 && y < g.H && x < g.W) out[s * g.n + (long)y * g.W + x] = ws[(py + 2) * WIN + (px + 2)] + upd;
    }
}


// ------------------------------------------------------------------------------------------------
// backward sweep, one launch per time step t (t = T .. 1, then a gather-only launch for t = 0):
//   phase 1   a_t[s](y) = inj_t + a_{t+1} + dt*coef_s*LapT(a_{t+1}[s])(y) + sum_{s',d} D_{t+1}[s'][c=s][d][y - d]
//             (gathers the per-tap scatter planes the previous launch wrote)            -> adj_out
//   phase 2   C = branches(h_{t-1}) (MFMA), G[s,k,j](x) = dt*a_t[s](x)*w4_j*prod_{k' != k} C_{k'j}(x),
//             D_t[s][kk][x] = sum_{k,j} W[s,k][j][kk] * G[s,k,j](x)  (MFMA: M = 64 taps (50 used), N = 16 points,
//             K = 48 channels; the K index is permuted so that G is consumed in the D layout it was produced in)
//                                                                                      -> D_out
// A wave owns (patch, species); it only consumes what the PREVIOUS launch produced => no intra-launch sync.
// Parameter gradients are not touched here: s1_wgrad_kernel reduces them over all steps in one launch.
// ------------------------------------------------------------------------------------------------
constexpr int NTAP = 50;

__global__ __launch_bounds__(64 * WAVES) void s1_adj_kernel(const float* __restrict__ h_prev,
                                                            const float* __restrict__ inj,
                                                            const float* __restrict__ adj_next,
                                                            const float* __restrict__ D_next,
                                                            float* __restrict__ adj_out, float* __restrict__ D_out,
                                                            const float* __restrict__ P, Geom g)
{
    __shared__ float lds[WAVES][3 * WIN * WIN];          // h window (2 species) + adjoint window (own species)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];
    float* awin = win + 2 * WIN * WIN;

    int patch = blockIdx.x * WAVES + wv;
    const int stride = gridDim.x * WAVES;
    if (patch >= g.npatch) return;

    const bool phase2 = D_out != nullptr;
    float a[3][NKS];
    float wT[4][3][4];                                   // [M-tile][branch][r]: W[s][k][4*grp + r][16*mt + (lane & 15)]
    int toff[NKS];
    float w4[4];
    if (phase2) {
        load_branch_weights(P, s, lane, a);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = 16 * mt + pt;
                    wT[mt][k][r] = kk < NTAP ? P[OFF_W + ((s * 3 + k) * HC + 4 * grp + r) * KK + kk] : 0.f;
                }
#pragma unroll
        for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);
#pragma unroll
        for (int r = 0; r < 4; ++r) w4[r] = P[OFF_W4 + s * HC + 4 * grp + r];
    }
    const float coef = P[P_COEF + s], dt = P[P_DT];

    for (; patch < g.npatch; patch += stride) {
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        const int y = y0 + py, x = x0 + px;
        const bool inside = y < g.H && x < g.W;
        const long pidx = (long)(inside ? y : 0) * g.W + (inside ? x : 0);

        wave_sync();
        if (phase2) {
            Window wnd;
            wnd.load(h_prev, g, y0, x0, lane);
            wnd.store(win, lane);
        }
        float at = inj ? inj[s * g.n + pidx] : 0.f;
        if (adj_next) {
            {   // adjoint window of the own species: 64 values, one per lane
                const int row = wrap1(y0 + (lane >> 3) - 2, g.H), col = wrap1(x0 + (lane & 7) - 2, g.W);
                awin[lane] = adj_next[s * g.n + (long)row * g.W + col];
            }
            // scatter planes: 50 (s', d) terms per point, 13 per lane group
            float gsum = 0.f;
#pragma unroll
            for (int m = 0; m < NKS; ++m) {
                const int i = 4 * m + grp;
                if (i < NTAP) {
                    const int sp = i >= 25 ? 1 : 0, d = i - 25 * sp;
                    const int dy = d / 5, dx = d - 5 * dy;
                    const int row = wrap1(y - dy + 2, g.H), col = wrap1(x - dx + 2, g.W);
                    gsum += D_next[(long)(sp * NTAP + s * 25 + d) * g.n + (long)row * g.W + col];
                }
            }
            gsum = xor_add(gsum, 16);
            gsum = xor_add(gsum, 32);
            wave_sync();
            const float lapT = win_star<-1>(awin, P, py, px);
            at += awin[(py + 2) * WIN + (px + 2)] + fma_(dt * coef, lapT, gsum);
        } else {
            wave_sync();
        }
        if (!inside) at = 0.f;
        if (grp == 0 && inside) adj_out[s * g.n + pidx] = at;
        if (!phase2) continue;

        f4 acc[3];
        branches(a, win, toff, grp, acc);
        const float ga = at * dt;
        float G[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gw = ga * w4[r];
            G[0][r] = gw * (acc[1][r] * acc[2][r]);
            G[1][r] = gw * (acc[0][r] * acc[2][r]);
            G[2][r] = gw * (acc[0][r] * acc[1][r]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f4 d = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wT[mt][k][r], G[k][r], d, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * mt + 4 * grp + r;
                if (kk < NTAP && inside) D_out[(long)(s * NTAP + kk) * g.n + pidx] = d[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// parameter gradients of ALL time steps in one launch (time-parallel, like the conv-wgrad of the base block):
// tasks = (t, patch) for species blockIdx.y; per task the branches are recomputed (39 MFMA), G is transposed
// through LDS and  dW[s,k][j][kk] += sum_x G[k][j][x] * col[kk][x]  runs as 48 MFMA (M = 16 channels,
// N = 64 taps (51 used), K = 16 points).  Accumulators stay in registers over all tasks of the wave; the four
// waves of a workgroup are summed in LDS and written as ONE float row per (workgroup, species);
// s1_reduce_kernel sums the rows in double, in a fixed order.
// Float row: [k][j][64 taps] (kk = 50: bias).  The 18 scalar-like sums (w4[16], b4, coef) cancel heavily, so they
// are accumulated in double per lane and travel in a second, double row.
// ------------------------------------------------------------------------------------------------
constexpr int ROW = 3 * HC * 64;
constexpr int ROWD = HC + 2;
constexpr int GT_LD = 20;                              // padded leading dimension of the transposed G tile

__global__ __launch_bounds__(64 * WAVES) void s1_wgrad_kernel(const float* __restrict__ traj,
                                                              const float* __restrict__ adj,
                                                              float* __restrict__ partials,
                                                              double* __restrict__ partials_d,
                                                              const float* __restrict__ P, Geom g, int T)
{
    __shared__ __attribute__((aligned(16))) float lds[WAVES][2 * WIN * WIN + 3 * HC * GT_LD];
    __shared__ float rowsum[ROW];
    __shared__ double rowsum_d[ROWD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];
    float* gt = win + 2 * WIN * WIN;

    float a[3][NKS];
    load_branch_weights(P, s, lane, a);
    int toff[NKS];
#pragma unroll
    for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);
    float w4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w4[r] = P[OFF_W4 + s * HC + 4 * grp + r];
    const float dt = P[P_DT];
    // B operand of the wgrad GEMM: lane (k-slot grp, n = pt) reads col[kk = 16*nt + pt] of point (py', px') = (grp, q')
    int tb[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int kk = 16 * nt + pt;
        kk = kk < NTAP ? kk : 0;
        const int c = kk >= 25 ? 1 : 0, rem = kk - 25 * c;
        const int dy = rem / 5, dx = rem - 5 * dy;
        tb[nt] = c * (WIN * WIN) + (grp + dy) * WIN + dx;
    }

    f4 wacc[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wacc[k][nt] = f4{0.f, 0.f, 0.f, 0.f};
    double acc_w4[4] = {0., 0., 0., 0.}, acc_b4 = 0., acc_cf = 0.;

    const long ntask = (long)T * g.npatch;
    const long frame = 2 * g.n;
    for (long task = (long)blockIdx.x * WAVES + wv; task < ntask; task += (long)gridDim.x * WAVES) {
        const int t = (int)(task / g.npatch) + 1, patch = (int)(task % g.npatch);
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        const int y = y0 + py, x = x0 + px;
        const bool inside = y < g.H && x < g.W;
        wave_sync();
        Window wnd;
        wnd.load(traj + (t - 1) * frame, g, y0, x0, lane);
        wnd.store(win, lane);
        const float ga = inside ? adj[t * frame + s * g.n + (long)y * g.W + x] * dt : 0.f;
        wave_sync();

        f4 acc[3];
        branches(a, win, toff, grp, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p12 = acc[0][r] * acc[1][r];
            acc_w4[r] += (double)(ga * (p12 * acc[2][r]));
            const float gw = ga * w4[r];
            const int j = 4 * grp + r;
            gt[(0 * HC + j) * GT_LD + pt] = gw * (acc[1][r] * acc[2][r]);
            gt[(1 * HC + j) * GT_LD + pt] = gw * (acc[0][r] * acc[2][r]);
            gt[(2 * HC + j) * GT_LD + pt] = gw * p12;
        }
        if (grp == 0) {
            acc_b4 += (double)ga;
            acc_cf += (double)(ga * win_star<1>(win + s * (WIN * WIN), P, py, px));
        }
        wave_sync();
        // A operand: lane (m = j = pt, k-slot grp) holds G[k][j][points 4*grp .. 4*grp+3]
        f4 ga4[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ga4[k] = *reinterpret_cast<const f4*>(gt + (k * HC + pt) * GT_LD + 4 * grp);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float b = win[tb[nt] + q];
                if (nt == 3) b = pt < 2 ? b : (pt == 2 ? 1.0f : 0.0f);          // kk = 50: bias, kk > 50: nothing
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    wacc[k][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga4[k][q], b, wacc[k][nt], 0, 0, 0);
            }
    }

    // sum over the 16 points held by the lanes of a group (w4) / by group 0 (b4, coef)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) acc_w4[r] += __shfl_xor(acc_w4[r], m, 64);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { acc_b4 += __shfl_xor(acc_b4, m, 64); acc_cf += __shfl_xor(acc_cf, m, 64); }

    for (int w = 0; w < WAVES; ++w) {
        if (wv == w) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (k * HC + 4 * grp + r) * 64 + 16 * nt + pt;
                        rowsum[idx] = (w == 0 ? 0.f : rowsum[idx]) + wacc[k][nt][r];
                    }
            if (pt == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = 4 * grp + r;
                    rowsum_d[idx] = (w == 0 ? 0. : rowsum_d[idx]) + acc_w4[r];
                }
            }
            if (lane == 0) {
                rowsum_d[HC] = (w == 0 ? 0. : rowsum_d[HC]) + acc_b4;
                rowsum_d[HC + 1] = (w == 0 ? 0. : rowsum_d[HC + 1]) + acc_cf;
            }
        }
        __syncthreads();
    }
    float* row = partials + ((long)blockIdx.x * 2 + s) * ROW;
    for (int i = threadIdx.x; i < ROW; i += blockDim.x) row[i] = rowsum[i];
    if (threadIdx.x < ROWD) partials_d[((long)blockIdx.x * 2 + s) * ROWD + threadIdx.x] = rowsum_d[threadIdx.x];
}

// one thread per gradient slot: fixed-order double sum over the workgroup rows
__global__ void s1_reduce_kernel(const float* __restrict__ partials, const double* __restrict__ partials_d, int nrows,
                                 double* __restrict__ pg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NP) return;
    int s = 0, idx = -1, didx = -1;
    if (i == P_COEF || i == P_COEF + 1) { s = i - P_COEF; didx = HC + 1; }
    else if (i >= OFF_B4) { s = i - OFF_B4; didx = HC; }
    else if (i >= OFF_W4) { s = (i - OFF_W4) / HC; didx = (i - OFF_W4) % HC; }
    else if (i >= OFF_W) {
        const int e = i - OFF_W, kk = e % KK, kj = e / KK;         // kj = (s*3 + k)*16 + j
        s = kj / (3 * HC);
        if (kk <= NTAP) idx = (kj % (3 * HC)) * 64 + kk;
    }
    double sum = 0.0;
    if (idx >= 0)
        for (int b = 0; b < nrows; ++b) sum += (double)partials[((long)b * 2 + s) * ROW + idx];
    if (didx >= 0)
        for (int b = 0; b < nrows; ++b) sum += partials_d[((long)b * 2 + s) * ROWD + didx];
    pg[i] = sum;
}

}  // namespace s1
}  // namespace pi
