// pi_tile2d.h -- temporally blocked 2D Pi-block kernels (gfx950).
//
// A 512^2 two-species state is 2 MiB: one step moves 4 MiB and ~80 MFLOP, i.e. well under a
// microsecond of chip time, while a dependent kernel boundary costs 1.5-1.9 us and a cold first
// load another ~1 us.  One launch per step is therefore latency-bound (measured 3.3-4.2 us/step).
// These kernels advance K steps per launch: a workgroup stages its BX x BY tile plus a 2K-wide
// periodic halo of the state in LDS (160 KiB/CU on CDNA4), recomputes the shrinking halo ring
// redundantly (region (B+4(K-m-1))^2 at sub-step m) and still writes EVERY intermediate frame of
// its own tile to the trajectory -- the API returns all steps (train_2drd.py:187-188).
//   * stencil reads: conflict-free ds_read_b32 (consecutive lanes -> consecutive x)
//   * global traffic: 16-byte coalesced loads of the window, 16-byte coalesced stores of the tile
//   * one __syncthreads per sub-step (ping-pong state buffers)
//   * per-point arithmetic and its order are IDENTICAL to the direct kernels (bit-equal results)
// The adjoint flavour keeps the adjoint state in LDS and streams h_{t-1}, dL/dout_{t-1} pointwise
// from HBM (they need no neighbours); diffusion-coefficient gradients are accumulated in registers
// across the K sub-steps and reduced once per launch.
#pragma once
#include "pi_device.h"

namespace pi {

template <int K, int BX, int BY>
struct Tile {
    static constexpr int LX = BX + 4 * K, LY = BY + 4 * K;
    static constexpr int PLANE = LX * LY;
    static constexpr int region_w(int m) { return LX - 4 * (m + 1); }
    static constexpr int region_h(int m) { return LY - 4 * (m + 1); }
    static constexpr int region_n(int m) { return region_w(m) * region_h(m); }
};

struct TileGeom {
    int H, W;          // grid
    long ss;           // species stride = H*W
    int tiles_x;       // W / BX
};

__device__ __forceinline__ int wrap1(int g, int n) { return g < 0 ? g + n : (g >= n ? g - n : g); }

// stage the (LY x LX) periodic window of both species of `src` into buf[2][LY][LX]
template <typename T, int K, int BX, int BY, int NT>
__device__ __forceinline__ void tile_load(const T* __restrict__ src, const TileGeom& g, int ty0, int tx0, T* buf)
{
    using TL = Tile<K, BX, BY>;
    constexpr int VEC = vec_width<T>::value;
    constexpr int LXV = TL::LX / VEC;
    constexpr int N = 2 * TL::LY * LXV;
    for (int i = threadIdx.x; i < N; i += NT) {
        const int s = i / (TL::LY * LXV);
        const int r = i - s * (TL::LY * LXV);
        const int ly = r / LXV, c = r - ly * LXV;
        const int gy = wrap1(ty0 - 2 * K + ly, g.H);
        const int gx = wrap1(tx0 - 2 * K + c * VEC, g.W);
        const Pack<T, VEC> p = ld<T, VEC>(src + s * g.ss + (long)gy * g.W + gx);
        st<T, VEC>(buf + s * TL::PLANE + ly * TL::LX + c * VEC, p);
    }
}

// write the BX x BY centre of buf to frame `dst`
template <typename T, int K, int BX, int BY, int NT>
__device__ __forceinline__ void tile_store(const T* buf, T* __restrict__ dst, const TileGeom& g, int ty0, int tx0)
{
    using TL = Tile<K, BX, BY>;
    constexpr int VEC = vec_width<T>::value;
    constexpr int BXV = BX / VEC;
    constexpr int N = 2 * BY * BXV;
    for (int i = threadIdx.x; i < N; i += NT) {
        const int s = i / (BY * BXV);
        const int r = i - s * (BY * BXV);
        const int y = r / BXV, c = r - y * BXV;
        const Pack<T, VEC> p = ld<T, VEC>(buf + s * TL::PLANE + (2 * K + y) * TL::LX + 2 * K + c * VEC);
        st<T, VEC>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC, p);
    }
}

// radius-2 star on an LDS plane; same tap order as pi::star (axis 0 = y first, then x)
template <typename T, int LX, int FLIP>
__device__ __forceinline__ T lds_star(const T* pl, int ly, int lx, const T* __restrict__ P)
{
    const T* c = pl + ly * LX + lx;
    T lap = P[P_C0] * c[0];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        lap = fma_(P[P_TAPS + t], c[k * LX], lap);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        lap = fma_(P[P_TAPS + 4 + t], c[k], lap);
    }
    return lap;
}

// ------------------------------------------------------------------------------------------------
// forward: frames t+1 .. t+K from frame t
// ------------------------------------------------------------------------------------------------
template <typename T, int HC, int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void fwd_substep(T* cur, T* nxt, const T* __restrict__ P)
{
    using TL = Tile<K, BX, BY>;
    constexpr int RW = TL::region_w(M), RN = TL::region_n(M), O = 2 * (M + 1);
    constexpr int PT = (RN + NT - 1) / NT;
    T u[PT], v[PT], lap[2][PT];
    int off[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        int idx = threadIdx.x + q * NT;
        if (idx >= RN) idx = RN - 1;                       // tail lanes recompute the last point (never stored twice differently)
        const int ry = idx / RW, rx = idx - ry * RW;
        off[q] = (ry + O) * TL::LX + rx + O;
        u[q] = cur[off[q]];
        v[q] = cur[TL::PLANE + off[q]];
        lap[0][q] = lds_star<T, TL::LX, +1>(cur, ry + O, rx + O, P);
        lap[1][q] = lds_star<T, TL::LX, +1>(cur + TL::PLANE, ry + O, rx + O, P);
    }
    const T dt = P[P_DT];
    // species / hidden-channel loops stay rolled (small I$-resident body, scalars prefetched)
#pragma clang loop unroll(disable)
    for (int s = 0; s < 2; ++s) {
        T rr[PT];
        if constexpr (HC == POLY) {
            const T* c = P + P_W + 10 * s;
#pragma unroll
            for (int q = 0; q < PT; ++q) rr[q] = poly_r(c, u[q], v[q]);
        } else {
            const T* W = P + P_W + s * species_block(HC);
#pragma unroll
            for (int q = 0; q < PT; ++q) rr[q] = W[10 * HC];
            W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
            for (int j = 0; j < HC; ++j) {
                const W10<T> c = nx;
                if (j + 1 < HC) nx = load_w10(W + 10 * (j + 1));
#pragma unroll
                for (int q = 0; q < PT; ++q) {
                    const T a1 = fma_(c.w[0], u[q], fma_(c.w[1], v[q], c.w[2]));
                    const T a2 = fma_(c.w[3], u[q], fma_(c.w[4], v[q], c.w[5]));
                    const T a3 = fma_(c.w[6], u[q], fma_(c.w[7], v[q], c.w[8]));
                    rr[q] = fma_(c.w[9], (a1 * a2) * a3, rr[q]);
                }
            }
        }
        const T coef = P[P_COEF + s];
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            const T lp = s == 0 ? lap[0][q] : lap[1][q];
            const T res = coef * lp + rr[q];
            const T inc = res * dt;
            nxt[s * TL::PLANE + off[q]] = (s == 0 ? u[q] : v[q]) + inc;
        }
    }
}

template <typename T, int HC, int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void fwd_substeps(T* b0, T* b1, T* __restrict__ frames, long frame_stride, const TileGeom& g,
                                             int ty0, int tx0, const T* __restrict__ P)
{
    T* cur = (M & 1) ? b1 : b0;
    T* nxt = (M & 1) ? b0 : b1;
    fwd_substep<T, HC, K, BX, BY, NT, M>(cur, nxt, P);
    lds_barrier();                                         // do not drain the previous frame's global stores
    tile_store<T, K, BX, BY, NT>(nxt, frames + (long)(M + 1) * frame_stride, g, ty0, tx0);
    if constexpr (M + 1 < K) fwd_substeps<T, HC, K, BX, BY, NT, M + 1>(b0, b1, frames, frame_stride, g, ty0, tx0, P);
}

template <typename T, int HC, int K, int BX, int BY, int NT>
__global__ void __launch_bounds__(NT)
pi_fwd2d_tile_kernel(T* __restrict__ frames /* frame t; t+1..t+K are written */, long frame_stride,
                     const T* __restrict__ P, TileGeom g)
{
    using TL = Tile<K, BX, BY>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw);
    T* b1 = b0 + 2 * TL::PLANE;
    const int tile = blockIdx.x;
    const int ty0 = (tile / g.tiles_x) * BY, tx0 = (tile % g.tiles_x) * BX;
    tile_load<T, K, BX, BY, NT>(frames, g, ty0, tx0, b0);
    __syncthreads();
    fwd_substeps<T, HC, K, BX, BY, NT, 0>(b0, b1, frames, frame_stride, g, ty0, tx0, P);
}

// ------------------------------------------------------------------------------------------------
// adjoint sweep: adj frames t-1 .. t-K from adj frame t (adjoint state lives in LDS)
//   hframes : trajectory; hframes + (t-1-m)*frame_stride is the state sub-step m linearises about
//   gframes : dL/dtraj;   gframes + (t-1-m)*frame_stride is injected at sub-step m if inj_mask bit m
//   aframes : adjoint trajectory; frame t is read, frames t-1..t-K are written
// ------------------------------------------------------------------------------------------------
template <typename T, int HC, int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void adj_substep(T* cur, T* nxt, const T* __restrict__ hfr, const T* __restrict__ gfr,
                                            const TileGeom& g, int ty0, int tx0, const T* __restrict__ P,
                                            T (&acc_c)[2])
{
    using TL = Tile<K, BX, BY>;
    constexpr int RW = TL::region_w(M), RN = TL::region_n(M), O = 2 * (M + 1);
    constexpr int PT = (RN + NT - 1) / NT;
    T u[PT], v[PT], ju[PT], jv[PT], gc[2][PT], dl[2][PT];
    int off[PT];
    bool own[PT];
    const T dt = P[P_DT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        int idx = threadIdx.x + q * NT;
        const bool live = idx < RN;
        if (!live) idx = RN - 1;
        const int ry = idx / RW, rx = idx - ry * RW;
        const int ly = ry + O, lx = rx + O;
        off[q] = ly * TL::LX + lx;
        own[q] = live && ly >= 2 * K && ly < 2 * K + BY && lx >= 2 * K && lx < 2 * K + BX;
        const int gy = wrap1(ty0 - 2 * K + ly, g.H), gx = wrap1(tx0 - 2 * K + lx, g.W);
        const long e = (long)gy * g.W + gx;
        u[q] = hfr[e];
        v[q] = hfr[g.ss + e];
        ju[q] = gfr ? gfr[e] : T(0);
        jv[q] = gfr ? gfr[g.ss + e] : T(0);
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int ly = off[q] / TL::LX, lx = off[q] - ly * TL::LX;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            gc[s][q] = cur[s * TL::PLANE + off[q]];
            dl[s][q] = lds_star<T, TL::LX, -1>(cur + s * TL::PLANE, ly, lx, P) * dt;
        }
        if (own[q]) {
            acc_c[0] += dl[0][q] * u[q];
            acc_c[1] += dl[1][q] * v[q];
        }
    }
    T du[PT], dv[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) du[q] = dv[q] = T(0);
#pragma clang loop unroll(disable)
    for (int s = 0; s < 2; ++s) {
        T gr[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) gr[q] = (s == 0 ? gc[0][q] : gc[1][q]) * dt;
        if constexpr (HC == POLY) {
            const T* c = P + P_W + 10 * s;
#pragma unroll
            for (int q = 0; q < PT; ++q) {
                T ru, rv;
                poly_dr(c, u[q], v[q], ru, rv);
                du[q] = fma_(gr[q], ru, du[q]);
                dv[q] = fma_(gr[q], rv, dv[q]);
            }
        } else {
            const T* W = P + P_W + s * species_block(HC);
            W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
            for (int j = 0; j < HC; ++j) {
                const W10<T> c = nx;
                if (j + 1 < HC) nx = load_w10(W + 10 * (j + 1));
#pragma unroll
                for (int q = 0; q < PT; ++q) {
                    const T a1 = fma_(c.w[0], u[q], fma_(c.w[1], v[q], c.w[2]));
                    const T a2 = fma_(c.w[3], u[q], fma_(c.w[4], v[q], c.w[5]));
                    const T a3 = fma_(c.w[6], u[q], fma_(c.w[7], v[q], c.w[8]));
                    const T p12 = a1 * a2;
                    const T gw = gr[q] * c.w[9];
                    const T q1 = gw * (a2 * a3), q2 = gw * (a1 * a3), q3 = gw * p12;
                    du[q] = fma_(q1, c.w[0], fma_(q2, c.w[3], fma_(q3, c.w[6], du[q])));
                    dv[q] = fma_(q1, c.w[1], fma_(q2, c.w[4], fma_(q3, c.w[7], dv[q])));
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const T tu = P[P_COEF + 0] * dl[0][q] + du[q];
        const T tv = P[P_COEF + 1] * dl[1][q] + dv[q];
        T ou = gc[0][q] + tu, ov = gc[1][q] + tv;
        if (gfr) { ou += ju[q]; ov += jv[q]; }
        nxt[off[q]] = ou;
        nxt[TL::PLANE + off[q]] = ov;
    }
}

template <typename T, int HC, int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void adj_substeps(T* b0, T* b1, const T* __restrict__ hbase, const T* __restrict__ gbase,
                                             T* __restrict__ abase, long frame_stride, unsigned inj_mask,
                                             T* __restrict__ g_h0, int steps_to_zero, const TileGeom& g, int ty0,
                                             int tx0, const T* __restrict__ P, T (&acc_c)[2])
{
    T* cur = (M & 1) ? b1 : b0;
    T* nxt = (M & 1) ? b0 : b1;
    const long fo = -(long)(M + 1) * frame_stride;         // frame t-1-M relative to frame t
    adj_substep<T, HC, K, BX, BY, NT, M>(cur, nxt, hbase + fo, (inj_mask >> M) & 1u ? gbase + fo : nullptr, g, ty0,
                                         tx0, P, acc_c);
    lds_barrier();
    // the adjoint of frame 0 is the caller's dL/dh0 output
    T* dst = (M + 1 == steps_to_zero && g_h0) ? g_h0 : abase + fo;
    tile_store<T, K, BX, BY, NT>(nxt, dst, g, ty0, tx0);
    if constexpr (M + 1 < K)
        adj_substeps<T, HC, K, BX, BY, NT, M + 1>(b0, b1, hbase, gbase, abase, frame_stride, inj_mask, g_h0,
                                                  steps_to_zero, g, ty0, tx0, P, acc_c);
}

template <typename T, int HC, int K, int BX, int BY, int NT>
__global__ void __launch_bounds__(NT)
pi_adj2d_tile_kernel(const T* __restrict__ hframe_t, const T* __restrict__ gframe_t, T* __restrict__ aframe_t,
                     long frame_stride, unsigned inj_mask, T* __restrict__ g_h0, int steps_to_zero,
                     double* __restrict__ partials, int np, const T* __restrict__ P, TileGeom g)
{
    using TL = Tile<K, BX, BY>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw);
    T* b1 = b0 + 2 * TL::PLANE;
    const int tile = blockIdx.x;
    const int ty0 = (tile / g.tiles_x) * BY, tx0 = (tile % g.tiles_x) * BX;
    tile_load<T, K, BX, BY, NT>(aframe_t, g, ty0, tx0, b0);
    __syncthreads();
    T acc_c[2] = {T(0), T(0)};
    adj_substeps<T, HC, K, BX, BY, NT, 0>(b0, b1, hframe_t, gframe_t, aframe_t, frame_stride, inj_mask, g_h0,
                                          steps_to_zero, g, ty0, tx0, P, acc_c);
    // diffusion-coefficient gradients of this tile over the K sub-steps: one reduction per launch
    __syncthreads();
    T* red = b0;                                           // state buffers are dead now
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const T r = wave_sum_to_last(acc_c[s]);
        if (lane == REDUCE_LANE) red[wave * 2 + s] = r;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        T sum = T(0);
        for (int w = 0; w < NT / WAVE; ++w) sum += red[w * 2 + threadIdx.x];
        partials[(long)blockIdx.x * np + P_COEF + threadIdx.x] += (double)sum;
    }
}

}  // namespace pi
