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
// f32 MFMA is an exact k-ordered fmaf chain, so the forward is reproducible bit for bit by a scalar fmaf loop over
// kk = 0..51 (that is what the parity tests compare against).
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

#ifdef PI_S1_TIMING
// debug build only: 100 MHz stamps of {species 0, wave 0} (slot 0) and {species 1, last wave} (slot 1) of every workgroup
__device__ long long s1_stamps[2 * 512 * 8];
#define S1_STAMP(i, dep) do { asm volatile("" ::"v"(dep)); \
        const int which_ = (blockIdx.y == 0 && threadIdx.x == 0) ? 0 : ((blockIdx.y == 1 && threadIdx.x == 64 * (WAVES - 1)) ? 1 : -1); \
        if (which_ >= 0 && blockIdx.x < 512) s1_stamps[(which_ * 512 + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define S1_STAMP(i, dep) do { } while (0)
#endif

struct Geom {
    int H, W;
    int px;        // patches per row = ceil(W / 4)
    int npy;       // patch rows = ceil(H / 4)
    int npatch;    // npy * px
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

// The species' three weight matrices [3][16][52] (10 KB) are staged in LDS by the whole workgroup with coalesced
// 16-byte loads (a per-lane gather straight from global touches 64 cache lines per instruction and dominated the
// launch at small grids), then each wave fills its resident A operands from LDS.
constexpr int WMAT = 3 * HC * KK;

__device__ __forceinline__ void stage_weights(const float* __restrict__ P, int s, float* wl)
{
    const float4* src = reinterpret_cast<const float4*>(P + OFF_W + s * WMAT);      // OFF_W, WMAT: multiples of 4
    float4* dst = reinterpret_cast<float4*>(wl);
    for (int i = threadIdx.x; i < WMAT / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

__device__ __forceinline__ void load_branch_weights(const float* wl, int lane, float (&a)[3][NKS])
{
    const int j = lane & 15, grp = lane >> 4;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < NKS; ++q) a[k][q] = wl[(k * HC + j) * KK + 4 * q + grp];
}

// the scalars of the block, fetched at the very top of a kernel: every launch starts with cold caches, so a load issued
// after the weight-staging barrier would cost a second ~1 us round trip
struct Consts {
    float dt, coef, b4, c0, taps[8], w4[4];
    __device__ __forceinline__ void load(const float* __restrict__ P, int s, int grp)
    {
        dt = P[P_DT]; coef = P[P_COEF + s]; b4 = P[OFF_B4 + s]; c0 = P[P_C0];
#pragma unroll
        for (int i = 0; i < 8; ++i) taps[i] = P[P_TAPS + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) w4[r] = P[OFF_W4 + s * HC + 4 * grp + r];
    }
};

// star Laplacian of species plane `w` (a [8][8] window) at patch point (py, px); summation order: centre, axis 0, axis 1
template <int FLIP>
__device__ __forceinline__ float win_star(const float* w, const Consts& k, int py, int px)
{
    const int c = (py + 2) * WIN + (px + 2);
    float lap = k.c0 * w[c];
    constexpr int offs[4] = {-2, -1, 1, 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) lap = fma_(k.taps[i], w[c + FLIP * offs[i] * WIN], lap);
#pragma unroll
    for (int i = 0; i < 4; ++i) lap = fma_(k.taps[4 + i], w[c + FLIP * offs[i]], lap);
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
    __shared__ __attribute__((aligned(16))) float wl[WMAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];

    int patch = blockIdx.x * WAVES + wv;
    const int stride = gridDim.x * WAVES;
    S1_STAMP(0, lane);
    Window wnd;
    if (patch < g.npatch) wnd.load(h, g, (patch / g.px) * 4, (patch % g.px) * 4, lane);
    Consts k;
    k.load(P, s, grp);
    stage_weights(P, s, wl);
    if (patch >= g.npatch) return;

    float a[3][NKS];
    load_branch_weights(wl, lane, a);
    S1_STAMP(1, a[2][NKS - 1]);
    int toff[NKS];
#pragma unroll
    for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);

    for (; patch < g.npatch; patch += stride) {
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        wave_sync();                                   // earlier reads of the window are done
        wnd.store(win, lane);
        wave_sync();
        if (patch + stride < g.npatch) {               // next window travels while the matrix cores work
            const int pn = patch + stride;
            wnd.load(h, g, (pn / g.px) * 4, (pn % g.px) * 4, lane);
        }
        S1_STAMP(2, wnd.v[0]);
        f4 acc[3];
        branches(a, win, toff, grp, acc);
        S1_STAMP(3, acc[2][3]);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fma_(k.w4[r], (acc[0][r] * acc[1][r]) * acc[2][r], t);
        t = xor_add(t, 16);                            // (c0 + c1), (c2 + c3)
        t = xor_add(t, 32);                            // + the other pair
        const float rr = t + k.b4;
        const float* ws = win + s * (WIN * WIN);
        const float lap = win_star<1>(ws, k, py, px);
        const float res = k.coef * lap + rr;
        const float upd = res * k.dt;
        const int y = y0 + py, x = x0 + px;
        if (grp == 0 && y < g.H && x < g.W) out[s * g.n + (long)y * g.W + x] = ws[(py + 2) * WIN + (px + 2)] + upd;
        S1_STAMP(4, upd);
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

// everything a wave reads from global memory for one (patch, species) task of the sweep -- issued in one go, for the
// first task BEFORE the weights are staged and for task i+1 while task i is on the matrix cores: every launch starts
// with cold caches (the inputs were written by the previous launch, mostly on other XCDs), so each dependent round
// trip costs ~1.5 us
// Two hand-over formats between consecutive launches:
//   ETILE (H, W multiples of 4 -- the reference's 100^2): each wave folds its 50 x 16 tap contributions into ONE 8x8
//         tile per input channel (the patch's footprint), 512 B per wave, written / read as whole cache lines; a point
//         then collects from the 4 tiles covering it (x 2 source species).  8x less hand-over traffic, no partial lines.
//   planes (any shape): D[s'][kk][n], one plane per tap, gathered with the tap's shift.
template <bool ETILE>
struct AdjIn {
    Window wnd;          // h_{t-1}, both species
    float aw;            // a_{t+1}[s] window, one value per lane
    float inj;           // dL/dtraj[t][s] at the own point
    float d[ETILE ? 2 : NKS];   // hand-over terms of the own lane group
    __device__ __forceinline__ void load(const float* __restrict__ h_prev, const float* __restrict__ inj_p,
                                         const float* __restrict__ adj_next, const float* __restrict__ D_next,
                                         const Geom& g, int patch, int s, int lane)
    {
        const int grp = lane >> 4, pt = lane & 15;
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        const int y = y0 + (pt >> 2), x = x0 + (pt & 3);
        const bool inside = y < g.H && x < g.W;
        if (h_prev) wnd.load(h_prev, g, y0, x0, lane);
        inj = (inj_p && inside) ? inj_p[s * g.n + (long)y * g.W + x] : 0.f;
        aw = 0.f;
        if (adj_next) {
            const int row = wrap1(y0 + (lane >> 3) - 2, g.H), col = wrap1(x0 + (lane & 7) - 2, g.W);
            aw = adj_next[s * g.n + (long)row * g.W + col];
            if constexpr (ETILE) {
                // lane group = (source species, own / neighbouring patch row); both patch columns per lane
                const int py = pt >> 2, px = pt & 3, sp = grp >> 1;
                const int dyy = (grp & 1) ? (py >= 2 ? 1 : -1) : 0;
                const int qy = wrap1(patch / g.px + dyy, g.npy), wyq = py - 4 * dyy + 2;
#pragma unroll
                for (int xs = 0; xs < 2; ++xs) {
                    const int dxx = xs ? (px >= 2 ? 1 : -1) : 0;
                    const int qx = wrap1(patch % g.px + dxx, g.px), wxq = px - 4 * dxx + 2;
                    d[xs] = D_next[(((long)sp * g.npatch + qy * g.px + qx) * 2 + s) * (WIN * WIN) + wyq * WIN + wxq];
                }
            } else
#pragma unroll
            for (int m = 0; m < NKS; ++m) {
                const int i = 4 * m + grp;
                const int ii = i < NTAP ? i : 0;
                const int sp = ii >= 25 ? 1 : 0, dd = ii - 25 * sp;
                const int dy = dd / 5, dx = dd - 5 * dy;
                const int row2 = wrap1(y - dy + 2, g.H), col2 = wrap1(x - dx + 2, g.W);
                const float v = D_next[(long)(sp * NTAP + s * 25 + dd) * g.n + (long)row2 * g.W + col2];
                d[m] = i < NTAP ? v : 0.f;
            }
        }
    }
};

template <bool ETILE>
__global__ __launch_bounds__(64 * WAVES) void s1_adj_kernel(const float* __restrict__ h_prev,
                                                            const float* __restrict__ inj,
                                                            const float* __restrict__ adj_next,
                                                            const float* __restrict__ D_next,
                                                            float* __restrict__ adj_out, float* __restrict__ D_out,
                                                            const float* __restrict__ P, Geom g)
{
    __shared__ float lds[WAVES][3 * WIN * WIN + (ETILE ? NTAP * 16 : 0)];   // h window, adjoint window, tap tile
    __shared__ __attribute__((aligned(16))) float wl[WMAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];
    float* awin = win + 2 * WIN * WIN;
    float* dl = awin + WIN * WIN;

    int patch = blockIdx.x * WAVES + wv;
    const int stride = gridDim.x * WAVES;
    const bool phase2 = D_out != nullptr;
    if (phase2) S1_STAMP(0, lane);
    AdjIn<ETILE> cur;
    if (patch < g.npatch) cur.load(h_prev, inj, adj_next, D_next, g, patch, s, lane);
    Consts k;
    k.load(P, s, grp);
    if (phase2) stage_weights(P, s, wl);
    if (patch >= g.npatch) return;

    float a[3][NKS];
    int toff[NKS];
    if (phase2) {
        load_branch_weights(wl, lane, a);
#pragma unroll
        for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);
    }
    if (phase2) S1_STAMP(1, a[2][NKS - 1]);

    for (; patch < g.npatch; patch += stride) {
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        const int y = y0 + py, x = x0 + px;
        const bool inside = y < g.H && x < g.W;
        const long pidx = (long)(inside ? y : 0) * g.W + (inside ? x : 0);

        wave_sync();
        if (phase2) cur.wnd.store(win, lane);
        awin[lane] = cur.aw;
        float gsum = 0.f;
#pragma unroll
        for (int m = 0; m < (ETILE ? 2 : NKS); ++m) gsum += cur.d[m];
        float at = cur.inj;
        const bool more = patch + stride < g.npatch;
        if (more) cur.load(h_prev, inj, adj_next, D_next, g, patch + stride, s, lane);
        wave_sync();
        if (adj_next) {
            gsum = xor_add(gsum, 16);
            gsum = xor_add(gsum, 32);
            const float lapT = win_star<-1>(awin, k, py, px);
            at += awin[(py + 2) * WIN + (px + 2)] + fma_(k.dt * k.coef, lapT, gsum);
        }
        if (!inside) at = 0.f;
        if (grp == 0 && inside) adj_out[s * g.n + pidx] = at;
        if (phase2) S1_STAMP(2, at);
        if (!phase2) continue;

        f4 acc[3];
        branches(a, win, toff, grp, acc);
        S1_STAMP(3, acc[2][3]);
        const float ga = at * k.dt;
        float G[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gw = ga * k.w4[r];
            G[0][r] = gw * (acc[1][r] * acc[2][r]);
            G[1][r] = gw * (acc[0][r] * acc[2][r]);
            G[2][r] = gw * (acc[0][r] * acc[1][r]);
        }
        // input-gradient GEMM; its A operand W[s][k][4*grp + r][16*mt + pt] comes straight from the staged weights
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f4 d = f4{0.f, 0.f, 0.f, 0.f};
            const int kk_a = 16 * mt + pt;
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float w = kk_a < NTAP ? wl[(k * HC + 4 * grp + r) * KK + kk_a] : 0.f;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w, G[k][r], d, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * mt + 4 * grp + r;
                if constexpr (ETILE) {
                    if (kk < NTAP) dl[kk * 16 + pt] = d[r];
                } else {
                    if (kk < NTAP && inside) D_out[(long)(s * NTAP + kk) * g.n + pidx] = d[r];
                }
            }
            if (mt == 3) S1_STAMP(4, d[3]);
        }
        if constexpr (ETILE) {
            // fold the 50 x 16 tap contributions into the patch's 8x8 footprint per input channel: window position
            // (wy, wx) collects tap (dy, dx) of source point (wy - dy, wx - dx) when that point is inside the patch
            wave_sync();
            const int wy = lane >> 3, wx = lane & 7;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float e = 0.f;
#pragma unroll
                for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) {
                        const int qy = wy - dy, qx = wx - dx;
                        const bool ok = (unsigned)qy < 4u && (unsigned)qx < 4u;
                        const float v = dl[(c * 25 + dy * 5 + dx) * 16 + (ok ? qy * 4 + qx : 0)];
                        e += ok ? v : 0.f;
                    }
                D_out[(((long)s * g.npatch + patch) * 2 + c) * (WIN * WIN) + lane] = e;
                if (c == 1) S1_STAMP(5, e);
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
    __shared__ __attribute__((aligned(16))) float rowsum[ROW];          // first the staged weights, at the end the row sums
    __shared__ double rowsum_d[ROWD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];
    float* gt = win + 2 * WIN * WIN;

    static_assert(WMAT <= ROW, "weights are staged in the row-sum buffer");
    float a[3][NKS];
    Consts k;
    k.load(P, s, grp);
    stage_weights(P, s, rowsum);
    load_branch_weights(rowsum, lane, a);
    __syncthreads();                                     // rowsum is reused for the final reduction
    int toff[NKS];
#pragma unroll
    for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);
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
    const long tstride = (long)gridDim.x * WAVES;
    Window wnd;
    float a_raw = 0.f;
    auto fetch = [&](long task) {                      // global reads of one task: h_{t-1} window + a_t at the own point
        const int t = (int)(task / g.npatch) + 1, patch = (int)(task % g.npatch);
        const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
        const int y = y0 + py, x = x0 + px;
        wnd.load(traj + (t - 1) * frame, g, y0, x0, lane);
        a_raw = (y < g.H && x < g.W) ? adj[t * frame + s * g.n + (long)y * g.W + x] : 0.f;
    };
    long task = (long)blockIdx.x * WAVES + wv;
    if (task < ntask) fetch(task);
    for (; task < ntask; task += tstride) {
        wave_sync();
        wnd.store(win, lane);
        const float ga = a_raw * k.dt;
        if (task + tstride < ntask) fetch(task + tstride);        // travels while this task is on the matrix cores
        wave_sync();

        f4 acc[3];
        branches(a, win, toff, grp, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p12 = acc[0][r] * acc[1][r];
            acc_w4[r] += (double)(ga * (p12 * acc[2][r]));
            const float gw = ga * k.w4[r];
            const int j = 4 * grp + r;
            gt[(0 * HC + j) * GT_LD + pt] = gw * (acc[1][r] * acc[2][r]);
            gt[(1 * HC + j) * GT_LD + pt] = gw * (acc[0][r] * acc[2][r]);
            gt[(2 * HC + j) * GT_LD + pt] = gw * p12;
        }
        if (grp == 0) {
            acc_b4 += (double)ga;
            acc_cf += (double)(ga * win_star<1>(win + s * (WIN * WIN), k, py, px));
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
// blockDim = (64 slots, 4 row lanes): lane r sums rows r, r+4, ...; the four partial sums are combined in a fixed order
__global__ void s1_reduce_kernel(const float* __restrict__ partials, const double* __restrict__ partials_d, int nrows,
                                 double* __restrict__ pg)
{
    __shared__ double part[4][64];
    const int i = blockIdx.x * 64 + threadIdx.x, rl = threadIdx.y;
    const bool live = i < NP;
    int s = 0, idx = -1, didx = -1;
    if (!live) { }
    else if (i == P_COEF || i == P_COEF + 1) { s = i - P_COEF; didx = HC + 1; }
    else if (i >= OFF_B4) { s = i - OFF_B4; didx = HC; }
    else if (i >= OFF_W4) { s = (i - OFF_W4) / HC; didx = (i - OFF_W4) % HC; }
    else if (i >= OFF_W) {
        const int e = i - OFF_W, kk = e % KK, kj = e / KK;         // kj = (s*3 + k)*16 + j
        s = kj / (3 * HC);
        if (kk <= NTAP) idx = (kj % (3 * HC)) * 64 + kk;
    }
    double sum = 0.0;
    if (live && idx >= 0) {
#pragma unroll 4
        for (int b = rl; b < nrows; b += 4) sum += (double)partials[((long)b * 2 + s) * ROW + idx];
    }
    if (live && didx >= 0) {
#pragma unroll 4
        for (int b = rl; b < nrows; b += 4) sum += partials_d[((long)b * 2 + s) * ROWD + didx];
    }
    part[rl][threadIdx.x] = sum;
    __syncthreads();
    if (live && rl == 0) pg[i] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ================================================================================================================
// Resident rollouts (round 5).  At the reference's 100^2 a launch per step is its boundary (2.8 us) plus one cold round trip for
// the operands the previous launch wrote (1.8 us) around 0.6 us (forward) / 1.3 us (sweep) of matrix-pipe work.  Here ONE
// launch walks all time steps: every wave owns one (patch, species) task for the whole rollout, and what a task needs from the
// tasks around it travels as data-tagged 8-byte granules {tag = step count, value} through an outbox in global memory
// (agent-scope write-through stores / loads, the idiom of the 2D resident kernels, pi_tile2d.h): a consumer polls until the tag
// is the step it waits for.  Two outbox halves by step parity suffice: a producer can only be one step ahead of its consumers
// (its next step needs THEIR output of this one).  No workgroup barrier after the weights are staged -- waves stay independent.
// Residency (every task must be on the machine at once) is checked by a roll call into a host-mapped word and bounded waits;
// an aborted launch is recomputed launch by launch (pi_s1_abi.hip).  Arithmetic = the per-step kernels' own functions, same
// order: trajectory and adjoint frames bit for bit theirs.  Shapes: H, W multiples of 4 (whole patches).
// ================================================================================================================
struct ResArgs {
    unsigned long long* outbox;     // two parity halves of granules
    unsigned* sync;                 // [0] roll call, [1] abort flag
    int* host;                      // host-mapped {roll call complete, step, task, aborted}
    unsigned long long timeout_ticks, first_timeout_ticks;     // 100 MHz ticks
    int nwg;                        // workgroups of the launch
    int pause;                      // s_sleep(1) units between publish and first request
    int masked;                     // sweep: only the frames whose bit is set in `frames` carry a dL/dtraj
    unsigned frames[128];           // (T < 4096)
};
typedef __attribute__((address_space(1))) unsigned long long res_gu64;

__device__ __forceinline__ void res_put(res_gu64* p, unsigned tag, float v)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float res_value(unsigned long long x) { return __builtin_bit_cast(float, (unsigned)x); }

// the wave's N granules per lane of one hand-over; false: timed out or somebody aborted
template <int N>
__device__ __forceinline__ bool res_poll(res_gu64* half, const unsigned (&gi)[N], unsigned tag, unsigned long long (&gx)[N],
                                         const ResArgs& ra, bool first)
{
#pragma unroll
    for (int q = 0; q < N; ++q) gx[q] = __hip_atomic_load(half + gi[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    const unsigned long long bound = first ? ra.first_timeout_ticks : ra.timeout_ticks;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < N; ++q) ok &= (unsigned)(gx[q] >> 32) == tag;
        if (__all(ok)) return true;
        if (wall_clock64() - t0 > bound || __hip_atomic_load(ra.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            return false;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int q = 0; q < N; ++q)                         // only what has not arrived yet is asked for again
            if ((unsigned)(gx[q] >> 32) != tag) gx[q] = __hip_atomic_load(half + gi[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void res_roll_call(const ResArgs& ra)
{
    if (threadIdx.x == 0) {
        const unsigned n = __hip_atomic_fetch_add(ra.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)ra.nwg && ra.host) __hip_atomic_store(ra.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the first wave to give up tells the others (abort flag) and the host (status slot)
__device__ __forceinline__ void res_abort(const ResArgs& ra, int step, int task, int lane)
{
    if (lane == 0 && __hip_atomic_exchange(ra.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ra.host) {
        __hip_atomic_store(ra.host + 1, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(ra.host + 2, task, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(ra.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// granules per parity half of the forward: [2 species][npatch][16 points]
__host__ __device__ inline size_t res_fwd_half(const Geom& g) { return (size_t)2 * g.npatch * 16; }
// ... of the sweep: adjoint values [2][npatch][16] | footprint tiles [2 producer species][npatch][2 channels][64]
__host__ __device__ inline size_t res_adj_half(const Geom& g) { return (size_t)2 * g.npatch * 16 + (size_t)2 * g.npatch * 2 * (WIN * WIN); }

// frames 1 .. T of `traj` from frame 0; grid = (ceil(npatch / WAVES), 2): one task per wave
__global__ __launch_bounds__(64 * WAVES) void s1_fwd_persist_kernel(float* __restrict__ traj, int T, const float* __restrict__ P,
                                                                    Geom g, ResArgs ra)
{
    __shared__ float lds[WAVES][2 * WIN * WIN];
    __shared__ __attribute__((aligned(16))) float wl[WMAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];
    res_roll_call(ra);
    const int patch = blockIdx.x * WAVES + wv;
    const bool active = patch < g.npatch;
    const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
    Window wnd;
    if (active) wnd.load(traj, g, y0, x0, lane);
    Consts k;
    k.load(P, s, grp);
    stage_weights(P, s, wl);
    if (!active) return;

    float a[3][NKS];
    load_branch_weights(wl, lane, a);
    int toff[NKS];
#pragma unroll
    for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);

    // where this lane's two window values (both species) come from, and where its own point goes
    unsigned gi[2];
    {
        const int row = wrap1(y0 + (lane >> 3) - 2, g.H), col = wrap1(x0 + (lane & 7) - 2, g.W);
        const unsigned src = (unsigned)(((row >> 2) * g.px + (col >> 2)) * 16 + (row & 3) * 4 + (col & 3));
        gi[0] = src;
        gi[1] = (unsigned)g.npatch * 16u + src;
    }
    const size_t half = res_fwd_half(g);
    res_gu64* outbox = (res_gu64*)ra.outbox;
    const unsigned mine = (unsigned)((s * g.npatch + patch) * 16 + pt);
    const long frame = 2 * g.n;
    const long own = s * g.n + (long)(y0 + py) * g.W + (x0 + px);

    for (int t = 0; t < T; ++t) {
        wave_sync();
        wnd.store(win, lane);
        wave_sync();
        f4 acc[3];
        branches(a, win, toff, grp, acc);
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) r = fma_(k.w4[i], (acc[0][i] * acc[1][i]) * acc[2][i], r);
        r = xor_add(r, 16);
        r = xor_add(r, 32);
        const float rr = r + k.b4;
        const float* ws = win + s * (WIN * WIN);
        const float lap = win_star<1>(ws, k, py, px);
        const float res = k.coef * lap + rr;
        const float upd = res * k.dt;
        const float nv = ws[(py + 2) * WIN + (px + 2)] + upd;
        const unsigned tag = (unsigned)t + 1u;
        res_gu64* hf = outbox + (size_t)(tag & 1u) * half;
        if (grp == 0) {
            if (t + 1 < T) res_put(hf + mine, tag, nv);      // the neighbours wait for this; the frame can follow
            traj[(long)(t + 1) * frame + own] = nv;
        }
        if (t + 1 == T) break;
        for (int w = 0; w < ra.pause; ++w) __builtin_amdgcn_s_sleep(1);
        unsigned long long gx[2];
        if (!res_poll<2>(hf, gi, tag, gx, ra, t == 0)) { res_abort(ra, t, 2 * patch + s, lane); return; }
        wnd.v[0] = res_value(gx[0]);
        wnd.v[1] = res_value(gx[1]);
    }
}

// the adjoint sweep t = T .. 0 (s1_adj_kernel<ETILE = true>'s two phases per step) in one launch: adjoint frames 1 .. T into
// `adj` (the time-parallel weight-gradient kernel reads them afterwards), dL/dh0 into g_h0.  What a step hands to the next --
// the adjoint values of the 8x8 window and the folded footprint tiles -- travels as granules; h_{t-1} and dL/dtraj[t] are
// plain loads, requested one step ahead.
__global__ __launch_bounds__(64 * WAVES) void s1_adj_persist_kernel(const float* __restrict__ traj, const float* __restrict__ g_traj,
                                                                    float* __restrict__ adj, float* __restrict__ g_h0, int T,
                                                                    const float* __restrict__ P, Geom g, ResArgs ra)
{
    __shared__ float lds[WAVES][3 * WIN * WIN + NTAP * 16];   // h window, adjoint window, tap tile
    __shared__ __attribute__((aligned(16))) float wl[WMAT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.y;
    const int grp = lane >> 4, pt = lane & 15, py = pt >> 2, px = pt & 3;
    float* win = lds[wv];
    float* awin = win + 2 * WIN * WIN;
    float* dl = awin + WIN * WIN;
    res_roll_call(ra);
    const int patch = blockIdx.x * WAVES + wv;
    const bool active = patch < g.npatch;
    const int y0 = (patch / g.px) * 4, x0 = (patch % g.px) * 4;
    const long frame = 2 * g.n;
    const long pidx = (long)(y0 + py) * g.W + (x0 + px);
    auto has_inj = [&](int t) { return !ra.masked || ((ra.frames[t >> 5] >> (t & 31)) & 1u); };
    Window wnd;
    float inj = 0.f;
    if (active) {
        if (T > 0) wnd.load(traj + (long)(T - 1) * frame, g, y0, x0, lane);
        if (has_inj(T)) inj = g_traj[(long)T * frame + s * g.n + pidx];
    }
    Consts k;
    k.load(P, s, grp);
    stage_weights(P, s, wl);
    if (!active) return;

    float a[3][NKS];
    int toff[NKS];
    load_branch_weights(wl, lane, a);
#pragma unroll
    for (int q = 0; q < NKS; ++q) toff[q] = tap_offset(q, grp, py, px);

    const unsigned nA = (unsigned)(2 * g.npatch * 16);
    unsigned gi[3];
    {
        const int row = wrap1(y0 + (lane >> 3) - 2, g.H), col = wrap1(x0 + (lane & 7) - 2, g.W);
        gi[0] = (unsigned)((s * g.npatch + (row >> 2) * g.px + (col >> 2)) * 16 + (row & 3) * 4 + (col & 3));
        // lane group = (producer species, own / neighbouring patch row); both patch columns per lane (AdjIn<true>::load)
        const int sp = grp >> 1;
        const int dyy = (grp & 1) ? (py >= 2 ? 1 : -1) : 0;
        const int qy = wrap1(patch / g.px + dyy, g.npy), wyq = py - 4 * dyy + 2;
#pragma unroll
        for (int xs = 0; xs < 2; ++xs) {
            const int dxx = xs ? (px >= 2 ? 1 : -1) : 0;
            const int qx = wrap1(patch % g.px + dxx, g.px), wxq = px - 4 * dxx + 2;
            gi[1 + xs] = nA + (unsigned)(((sp * g.npatch + qy * g.px + qx) * 2 + s) * (WIN * WIN) + wyq * WIN + wxq);
        }
    }
    const size_t half = res_adj_half(g);
    res_gu64* outbox = (res_gu64*)ra.outbox;
    const unsigned mineA = (unsigned)((s * g.npatch + patch) * 16 + pt);
    const unsigned mineD = nA + (unsigned)(((s * g.npatch + patch) * 2) * (WIN * WIN) + lane);

    for (int t = T; t >= 0; --t) {
        const unsigned kdone = (unsigned)(T - t);           // what step t + 1 produced carries tag kdone, what this one produces kdone + 1
        float aw = 0.f, gsum = 0.f;
        if (t < T) {
            unsigned long long gx[3];
            if (!res_poll<3>(outbox + (size_t)(kdone & 1u) * half, gi, kdone, gx, ra, t == T - 1)) {
                res_abort(ra, t, 2 * patch + s, lane);
                return;
            }
            aw = res_value(gx[0]);
            gsum = res_value(gx[1]) + res_value(gx[2]);
        }
        wave_sync();
        if (t > 0) wnd.store(win, lane);
        awin[lane] = aw;
        float at = inj;
        if (t > 0) {                                        // next step's plain operands travel under this step's arithmetic
            inj = has_inj(t - 1) ? g_traj[(long)(t - 1) * frame + s * g.n + pidx] : 0.f;
            if (t > 1) wnd.load(traj + (long)(t - 2) * frame, g, y0, x0, lane);
        }
        wave_sync();
        if (t < T) {
            gsum = xor_add(gsum, 16);
            gsum = xor_add(gsum, 32);
            const float lapT = win_star<-1>(awin, k, py, px);
            at += awin[(py + 2) * WIN + (px + 2)] + fma_(k.dt * k.coef, lapT, gsum);
        }
        if (t == 0) {
            if (grp == 0) g_h0[s * g.n + pidx] = at;
            break;
        }
        res_gu64* hf = outbox + (size_t)((kdone + 1u) & 1u) * half;
        if (grp == 0) {
            res_put(hf + mineA, kdone + 1u, at);
            adj[(long)t * frame + s * g.n + pidx] = at;
        }

        f4 acc[3];
        branches(a, win, toff, grp, acc);
        const float ga = at * k.dt;
        float G[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gw = ga * k.w4[r];
            G[0][r] = gw * (acc[1][r] * acc[2][r]);
            G[1][r] = gw * (acc[0][r] * acc[2][r]);
            G[2][r] = gw * (acc[0][r] * acc[1][r]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f4 d = f4{0.f, 0.f, 0.f, 0.f};
            const int kk_a = 16 * mt + pt;
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float w = kk_a < NTAP ? wl[(kb * HC + 4 * grp + r) * KK + kk_a] : 0.f;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w, G[kb][r], d, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * mt + 4 * grp + r;
                if (kk < NTAP) dl[kk * 16 + pt] = d[r];
            }
        }
        wave_sync();
        const int wy = lane >> 3, wx = lane & 7;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float e = 0.f;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) {
                    const int qy = wy - dy, qx = wx - dx;
                    const bool ok = (unsigned)qy < 4u && (unsigned)qx < 4u;
                    const float v = dl[(c * 25 + dy * 5 + dx) * 16 + (ok ? qy * 4 + qx : 0)];
                    e += ok ? v : 0.f;
                }
            res_put(hf + mineD + (unsigned)c * (WIN * WIN), kdone + 1u, e);
        }
        for (int w = 0; w < ra.pause; ++w) __builtin_amdgcn_s_sleep(1);
    }
}

}  // namespace s1
}  // namespace pi
