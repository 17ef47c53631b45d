// pi_adv.h -- "advective polynomial" step kernels (gfx950): Stage-3 physics-based cells.
//
//   rhs_s = coef_s Lap(h_s) + r_s(u,v) + sum_a (cu_{s,a} u + cv_{s,a} v) D_a(h_s),   next = h + dt * rhs
// with D_a the 4th-order central first derivative along axis a (taps in the block).  Reference: 2D Burgers
// Stage-3 f_rhs / forward (DataDrivenDiscoveryOfPDEs/2D_Burgers_eqn/Stage-3/fine_tuning_[5%noise,41x51x51].py:
// 154-157, 209-221): three circular 5x5 convolutions per species and ~10 elementwise launches per step,
// float64, 100^2 grids -- entirely launch-bound there; here one fused launch per step, and one for the
// adjoint (adjoint state + all 30 gradient sums).  Block layout ("hc = -1", 60 entries): [0..35] as the
// pre-contracted block; [36+4a+i] derivative taps; [48+6s+2a+{0,1}] = (cu, cv).  One point per lane: the
// reference's grids are tiny, the win is fusion over T, not bandwidth.
#pragma once
#include "pi_device.h"
#include "pi_kernels.h"

namespace pi {

constexpr int ADV = -2;                 // template tag
constexpr int NADV = 60, A_DTAPS = 36, A_ADV = 48;

// value of field f (one species plane set) at the point shifted by `off` along axis a (periodic)
template <int NDIM>
__device__ __forceinline__ long shifted(const Geom& g, int i0, int i1, int x, int a, int off)
{
    int j0 = i0, j1 = i1, jx = x;
    if (a == 0) j0 = wrap_near(i0 + off, g.n0);
    else if (NDIM == 3 && a == 1) j1 = wrap_near(i1 + off, g.n1);
    else jx = wrap_near(x + off, g.W);
    return (long)j0 * g.s0 + (NDIM == 3 ? (long)j1 * g.W : 0) + jx;
}

template <typename T, int NDIM>
__global__ void __launch_bounds__(256)
pi_adv_fwd_kernel(const T* __restrict__ h, T* __restrict__ out, const T* __restrict__ A, Geom g)
{
    const long n = (long)g.rows * g.W;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int i0, i1, x;
    long e;
    chunk_coords<NDIM>(g, p, g.W, 1, i0, i1, x, e);
    const Pack<T, 1> cu{{h[e]}}, cv{{h[g.ss + e]}};
    const T u = cu.v[0], v = cv.v[0];
    const T dt = A[P_DT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const T* f = h + s * g.ss;
        T lap[1];
        star<T, NDIM, 1, +1>(f, A, g, i0, i1, x, e, s == 0 ? cu : cv, lap);
        const T rr = poly_r(A + P_W + 10 * s, u, v);
        T adv = T(0);
#pragma unroll
        for (int a = 0; a < NDIM; ++a) {
            T d = A[A_DTAPS + 4 * a] * f[shifted<NDIM>(g, i0, i1, x, a, -2)];
            d = fma_(A[A_DTAPS + 4 * a + 1], f[shifted<NDIM>(g, i0, i1, x, a, -1)], d);
            d = fma_(A[A_DTAPS + 4 * a + 2], f[shifted<NDIM>(g, i0, i1, x, a, +1)], d);
            d = fma_(A[A_DTAPS + 4 * a + 3], f[shifted<NDIM>(g, i0, i1, x, a, +2)], d);
            const T c = fma_(A[A_ADV + 6 * s + 2 * a], u, A[A_ADV + 6 * s + 2 * a + 1] * v);
            adv = fma_(c, d, adv);
        }
        const T res = A[P_COEF + s] * lap[0] + (rr + adv);
        const T inc = res * dt;
        out[s * g.ss + e] = f[e] + inc;
    }
}

// adjoint of one step + every gradient sum (2 diffusion coefficients, 20 moments, 12 advection coefficients)
template <typename T, int NDIM>
__global__ void __launch_bounds__(256)
pi_adv_bwd_kernel(const T* __restrict__ h, const T* __restrict__ G, const T* __restrict__ inj, T* __restrict__ Gp,
                  double* __restrict__ partials, const T* __restrict__ A, Geom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);          // [nwaves][NADV]
    const int nwaves = blockDim.x / WAVE, wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    for (int i = threadIdx.x; i < nwaves * NADV; i += blockDim.x) red[i] = 0.0;
    __syncthreads();
    double* myred = red + wave * NADV;
    const long n = (long)g.rows * g.W;
    const T dt = A[P_DT];
    const long stride = (long)gridDim.x * blockDim.x;
    const long iters = (n + stride - 1) / stride;
    for (long it = 0; it < iters; ++it) {
        const long praw = (long)blockIdx.x * blockDim.x + threadIdx.x + it * stride;
        const bool valid = praw < n;
        const long p = valid ? praw : n - 1;
        int i0, i1, x;
        long e;
        chunk_coords<NDIM>(g, p, g.W, 1, i0, i1, x, e);
        const T live = valid ? T(1) : T(0);
        const T u = h[e], v = h[g.ss + e];
        const T u2 = u * u, uv = u * v, v2 = v * v;
        const T phi[10] = {T(1), u, v, u2, uv, v2, u2 * u, u2 * v, u * v2, v2 * v};
        T dsp[2] = {T(0), T(0)}, advT[2] = {T(0), T(0)}, dl[2], gcen[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const T* Gs = G + s * g.ss;
            const T* hs = h + s * g.ss;
            const T* c = A + P_W + 10 * s;
            const Pack<T, 1> gc{{Gs[e]}};
            gcen[s] = gc.v[0];
            T lg[1];
            star<T, NDIM, 1, -1>(Gs, A, g, i0, i1, x, e, gc, lg);
            dl[s] = lg[0] * dt;
            const T gr = (gc.v[0] * dt) * live;
            double acc[1 + 10 + 2 * NDIM];
            acc[0] = (double)((dl[s] * live) * hs[e]);
            acc[1] = (double)gr;
#pragma unroll
            for (int m = 1; m < 10; ++m) acc[1 + m] = (double)(gr * phi[m]);
            T ru, rv;
            poly_dr(c, u, v, ru, rv);
            dsp[0] = fma_(gr, ru, dsp[0]);
            dsp[1] = fma_(gr, rv, dsp[1]);
#pragma unroll
            for (int a = 0; a < NDIM; ++a) {
                const T cu = A[A_ADV + 6 * s + 2 * a], cv = A[A_ADV + 6 * s + 2 * a + 1];
                T d = A[A_DTAPS + 4 * a] * hs[shifted<NDIM>(g, i0, i1, x, a, -2)];
                d = fma_(A[A_DTAPS + 4 * a + 1], hs[shifted<NDIM>(g, i0, i1, x, a, -1)], d);
                d = fma_(A[A_DTAPS + 4 * a + 2], hs[shifted<NDIM>(g, i0, i1, x, a, +1)], d);
                d = fma_(A[A_DTAPS + 4 * a + 3], hs[shifted<NDIM>(g, i0, i1, x, a, +2)], d);
                const T gd = gr * d;
                acc[11 + 2 * a] = (double)(gd * u);
                acc[12 + 2 * a] = (double)(gd * v);
                dsp[0] = fma_(gd, cu, dsp[0]);
                dsp[1] = fma_(gd, cv, dsp[1]);
                // transposed derivative stencil: sum_i tap_i * m(x - off_i e_a),  m = (cu u + cv v) * G_s * dt
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int off = -(i < 2 ? i - 2 : i - 1);
                    const long en = shifted<NDIM>(g, i0, i1, x, a, off);
                    const T m = fma_(cu, h[en], cv * h[g.ss + en]) * (Gs[en] * dt);
                    advT[s] = fma_(A[A_DTAPS + 4 * a + i], m, advT[s]);
                }
            }
            // wave-level sums -> this wave's LDS row (fp64: several of these sums cancel heavily)
#pragma unroll
            for (int k = 0; k < 1 + 10 + 2 * NDIM; ++k) acc[k] = wave_sum_to_last(acc[k]);
            if (lane == REDUCE_LANE) {
                myred[P_COEF + s] += acc[0];
#pragma unroll
                for (int m = 0; m < 10; ++m) myred[P_W + 10 * s + m] += acc[1 + m];
#pragma unroll
                for (int a = 0; a < NDIM; ++a) {
                    myred[A_ADV + 6 * s + 2 * a] += acc[11 + 2 * a];
                    myred[A_ADV + 6 * s + 2 * a + 1] += acc[12 + 2 * a];
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const T t = A[P_COEF + s] * dl[s] + (dsp[s] + advT[s]);
                T o = gcen[s] + t;
                if (inj) o += inj[s * g.ss + e];
                Gp[s * g.ss + e] = o;
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NADV; idx += blockDim.x) {
        double sum = 0.0;
        for (int w = 0; w < nwaves; ++w) sum += red[w * NADV + idx];
        partials[(long)blockIdx.x * NADV + idx] += sum;
    }
}

}  // namespace pi
