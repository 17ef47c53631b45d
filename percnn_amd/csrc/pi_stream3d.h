// pi_stream3d.h -- plane-streaming ("2.5-D") 3D Pi-block kernels for gfx950.
//
// The direct 3D kernel issues 11 sixteen-byte loads per species per lane (13-point star): it runs
// at ~2.7 TB/s effective, bound by the vector-L1/TA path, not by HBM.  Here a workgroup owns a strip
// of TY full-width grid rows and marches along axis 0 (z):
//   * one 64-lane wave == one full row of W = 64*VEC points -> every global access is one fully
//     coalesced row segment, periodic wrap in x is just LDS column arithmetic;
//   * z-neighbours live in a 5-deep REGISTER queue (each plane is fetched once per lane);
//   * y/x-neighbours of the current plane come from LDS: the TY rows written by the waves plus two
//     halo rows on each side, double-buffered -> ONE workgroup barrier per plane;
//   * next plane / next halo rows are requested before the current plane is computed (software
//     prefetch hides L2/HBM latency at the 1-2 waves/SIMD a 128^3 grid offers).
// Arithmetic and its order are identical to pi::star + pi_fwd_kernel / pi_bwd_kernel (bit-equal).
#pragma once
#include "pi_device.h"
#include "pi_kernels.h"

namespace pi {

template <typename T, int VEC, int TY>
struct Strip {
    static constexpr int W = WAVE * VEC;
    static constexpr int ROWS = TY + 4;
    static constexpr int PLANE = ROWS * W;              // one species, one buffer
};

// in-plane (y then x) part of the star for the lane's VEC points of row `r` (LDS row index, halo offset 2)
template <typename T, int VEC, int TY, int FLIP>
__device__ __forceinline__ void inplane_taps(const T* pl, int r, int x0, const T* __restrict__ P, T (&lap)[VEC])
{
    using S = Strip<T, VEC, TY>;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        const Pack<T, VEC> nb = ld<T, VEC>(pl + (r + k) * S::W + x0);
        const T w = P[P_TAPS + 4 + t];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, nb.v[i], lap[i]);
    }
    T win[VEC + 4];
    const T* row = pl + r * S::W;
    if constexpr (VEC == 1) {
        win[0] = row[(x0 - 2) & (S::W - 1)];
        win[1] = row[(x0 - 1) & (S::W - 1)];
        win[2] = row[x0];
        win[3] = row[(x0 + 1) & (S::W - 1)];
        win[4] = row[(x0 + 2) & (S::W - 1)];
    } else {
        const Pack<T, 2> l = ld<T, 2>(row + ((x0 - 2) & (S::W - 1)));
        const Pack<T, VEC> c = ld<T, VEC>(row + x0);
        const Pack<T, 2> rr = ld<T, 2>(row + ((x0 + VEC) & (S::W - 1)));
        win[0] = l.v[0]; win[1] = l.v[1];
#pragma unroll
        for (int i = 0; i < VEC; ++i) win[2 + i] = c.v[i];
        win[VEC + 2] = rr.v[0]; win[VEC + 3] = rr.v[1];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        const T w = P[P_TAPS + 8 + t];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, win[2 + i + k], lap[i]);
    }
}

// One plane-iteration with compile-time ring phase J.  Slots: plane z-2+k lives in q[s][(J+k) % R]
// (R = 8: planes z-2 .. z+5, i.e. three planes of look-ahead beyond the stencil); the halo-row and
// adjoint-operand rings have 4 slots (planes z .. z+3).  Rotating INDICES instead of registers is what
// keeps the prefetches in flight: moving a register that a load has not filled yet forces vmcnt(0).
template <typename T, int HC, int VEC, int TY, bool ADJ>
struct Stream3D {
    using S = Strip<T, VEC, TY>;
    static constexpr int FLIP = ADJ ? -1 : +1;
    static constexpr int R = 8, RH = 4;
    Pack<T, VEC> q[2][R];
    Pack<T, VEC> hp[2][RH];
    Pack<T, VEC> ph[2][ADJ ? RH : 1], pj[2][ADJ ? RH : 1];
    double acc_c[2];        // heavily cancelling sums: fp64
    // adjoint, float32 poly mode: the 20 coefficient moments sum dt*a[s]*phi_m(h) accumulate per lane over the whole
    // z-march (one cross-lane reduction per launch) -- replaces the separate pi_moments_kernel pass over both trajectories
    static constexpr bool MOM = ADJ && HC == POLY && sizeof(T) == 4;
    T mom[MOM ? 2 : 1][MOM ? 10 : 1];
    int with_mom;
    // uniform state
    const T* f; T* out; const T* h; const T* inj; const T* P; T* lds;
    Geom g; int hc, wy, x0, y, hy, hrow, z1;
    long rowoff, hrowoff;
    int zq, zh;             // wrapped plane indices of the next queue / halo-row request

    __device__ __forceinline__ const T* plane(const T* base, int s, int zz) const
    {
        return base + s * g.ss + g.off + (long)zz * g.s0;
    }
    __device__ __forceinline__ int next_plane(int zz) const
    {
        ++zz;
        return (g.wrap0 && zz == g.n0) ? 0 : zz;
    }

    template <int J>
    __device__ __forceinline__ void iter(int z, int cur)
    {
        T* buf = lds + cur * 2 * S::PLANE;
        constexpr int C = (J + 2) % R;                  // slot of plane z
        constexpr int HS = J % RH;                      // slot of plane z in the 4-deep rings
        // 1) publish plane z of my row and of my halo row, then re-arm the halo slot (plane z+4)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st<T, VEC>(buf + s * S::PLANE + (2 + wy) * S::W + x0, q[s][C]);
            st<T, VEC>(buf + s * S::PLANE + hrow * S::W + x0, hp[s][HS]);
        }
        const bool more_h = z + RH < z1;
        if (more_h) {
#pragma unroll
            for (int s = 0; s < 2; ++s) hp[s][HS] = ld<T, VEC>(plane(f, s, zh) + hrowoff);
        }
        const long e = (long)z * g.s0 + rowoff;         // interior-relative (z in [0,n0): never wrapped)
        lds_barrier();                                   // LDS only: global requests stay in flight
        // 2) stencil: centre, axis 0 (register queue), then axes 1, 2 (LDS)
        T lap[2][VEC];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) lap[s][i] = P[P_C0] * q[s][C].v[i];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                constexpr int dummy = 0; (void)dummy;
                const int k = FLIP * (t < 2 ? t - 2 : t - 1);
                const T w = P[P_TAPS + t];
                const Pack<T, VEC>& nb = q[s][(J + 2 + k + R) % R];
#pragma unroll
                for (int i = 0; i < VEC; ++i) lap[s][i] = fma_(w, nb.v[i], lap[s][i]);
            }
            inplane_taps<T, VEC, TY, FLIP>(buf + s * S::PLANE, 2 + wy, x0, P, lap[s]);
        }
        // plane z-2 is dead now: its slot receives plane z+6
        if (z + 6 <= z1 + 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) q[s][J % R] = ld<T, VEC>(plane(f, s, zq) + rowoff);
        }
        zq = next_plane(zq);
        const Pack<T, VEC> cu = q[0][C], cv = q[1][C];
        const T dt = P[P_DT];
        // 3) reaction + update
        if constexpr (!ADJ) {
#pragma clang loop unroll(disable)
            for (int s = 0; s < 2; ++s) {
                T rr[VEC];
                if constexpr (HC == POLY) {
                    const T* c = P + P_W + 10 * s;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) rr[i] = poly_r(c, cu.v[i], cv.v[i]);
                } else {
                    const T* W = P + P_W + s * species_block(hc);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) rr[i] = W[10 * hc];
                    W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
                    for (int j = 0; j < hc; ++j) {
                        const W10<T> c = nx;
                        if (j + 1 < hc) nx = load_w10(W + 10 * (j + 1));
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const T a1 = fma_(c.w[0], cu.v[i], fma_(c.w[1], cv.v[i], c.w[2]));
                            const T a2 = fma_(c.w[3], cu.v[i], fma_(c.w[4], cv.v[i], c.w[5]));
                            const T a3 = fma_(c.w[6], cu.v[i], fma_(c.w[7], cv.v[i], c.w[8]));
                            rr[i] = fma_(c.w[9], (a1 * a2) * a3, rr[i]);
                        }
                    }
                }
                const T coef = P[P_COEF + s];
                Pack<T, VEC> o;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const T hs = s == 0 ? cu.v[i] : cv.v[i];
                    const T lp = s == 0 ? lap[0][i] : lap[1][i];
                    const T res = coef * lp + rr[i];
                    const T inc = res * dt;
                    o.v[i] = hs + inc;
                }
                st<T, VEC>(out + s * g.ss + g.off + e, o);
            }
        } else {
            const Pack<T, VEC> hu = ph[0][HS], hv = ph[1][HS];
            Pack<T, VEC> ju, jv;
            if (inj) { ju = pj[0][HS]; jv = pj[1][HS]; }
            if (more_h) {                                // re-arm the operand slots (plane z+4)
                const long en = g.off + (long)(z + RH) * g.s0 + rowoff;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    ph[s][HS] = ld<T, VEC>(h + s * g.ss + en);
                    if (inj) pj[s][HS] = ld<T, VEC>(inj + s * g.ss + en);
                }
            }
            T du[VEC], dv[VEC], dl[2][VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                du[i] = dv[i] = T(0);
                dl[0][i] = lap[0][i] * dt;
                dl[1][i] = lap[1][i] * dt;
                acc_c[0] += (double)(dl[0][i] * hu.v[i]);
                acc_c[1] += (double)(dl[1][i] * hv.v[i]);
            }
#pragma clang loop unroll(disable)
            for (int s = 0; s < 2; ++s) {
                T gr[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) gr[i] = (s == 0 ? cu.v[i] : cv.v[i]) * dt;
                if constexpr (HC == POLY) {
                    const T* c = P + P_W + 10 * s;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        T ru, rv;
                        poly_dr(c, hu.v[i], hv.v[i], ru, rv);
                        du[i] = fma_(gr[i], ru, du[i]);
                        dv[i] = fma_(gr[i], rv, dv[i]);
                    }
                    if constexpr (MOM) {
                        if (with_mom) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                const T uu = hu.v[i], vv = hv.v[i];
                                const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
                                const T phi[10] = {T(1), uu, vv, u2, uv, v2, u2 * uu, u2 * vv, uu * v2, v2 * vv};
                                T* ms = s == 0 ? mom[0] : mom[MOM ? 1 : 0];
                                ms[0] += gr[i];
#pragma unroll
                                for (int q = 1; q < 10; ++q) ms[q] = fma_(gr[i], phi[q], ms[q]);
                            }
                        }
                    }
                } else {
                    const T* W = P + P_W + s * species_block(hc);
                    W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
                    for (int j = 0; j < hc; ++j) {
                        const W10<T> c = nx;
                        if (j + 1 < hc) nx = load_w10(W + 10 * (j + 1));
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const T a1 = fma_(c.w[0], hu.v[i], fma_(c.w[1], hv.v[i], c.w[2]));
                            const T a2 = fma_(c.w[3], hu.v[i], fma_(c.w[4], hv.v[i], c.w[5]));
                            const T a3 = fma_(c.w[6], hu.v[i], fma_(c.w[7], hv.v[i], c.w[8]));
                            const T p12 = a1 * a2;
                            const T gw = gr[i] * c.w[9];
                            const T q1 = gw * (a2 * a3), q2 = gw * (a1 * a3), q3 = gw * p12;
                            du[i] = fma_(q1, c.w[0], fma_(q2, c.w[3], fma_(q3, c.w[6], du[i])));
                            dv[i] = fma_(q1, c.w[1], fma_(q2, c.w[4], fma_(q3, c.w[7], dv[i])));
                        }
                    }
                }
            }
            Pack<T, VEC> ou, ov;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const T tu = P[P_COEF + 0] * dl[0][i] + du[i];
                const T tv = P[P_COEF + 1] * dl[1][i] + dv[i];
                ou.v[i] = cu.v[i] + tu;
                ov.v[i] = cv.v[i] + tv;
                if (inj) { ou.v[i] += ju.v[i]; ov.v[i] += jv.v[i]; }
            }
            st<T, VEC>(out + g.off + e, ou);
            st<T, VEC>(out + g.ss + g.off + e, ov);
        }
        zh = next_plane(zh);
    }

    template <int J>
    __device__ __forceinline__ void run8(int z, int& cur)
    {
        if (z + J < z1) {
            iter<J>(z + J, cur);
            cur ^= 1;
            if constexpr (J + 1 < R) run8<J + 1>(z, cur);
        }
    }
};

// FWD: out = step(f);  ADJ: out = adjoint step of f (=G) about state h (+ inj), dcoef partials
template <typename T, int HC, int VEC, int TY, bool ADJ>
__global__ void __launch_bounds__(WAVE * TY)
pi_stream3d_kernel(const T* __restrict__ f,        // stencil-read field: state (fwd) or adjoint G (adj)
                   T* __restrict__ out,             // next state (fwd) or adjoint of the previous state (adj)
                   const T* __restrict__ h,         // adj only: state the step was applied to
                   const T* __restrict__ inj,       // adj only, nullable
                   double* __restrict__ partials,   // adj only
                   const T* __restrict__ P, Geom g, int zc, int hc_rt, int with_mom)
{
    using S = Strip<T, VEC, TY>;
    using K = Stream3D<T, HC, VEC, TY, ADJ>;
    static_assert(TY == 4, "one halo row per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    K k;
    k.lds = reinterpret_cast<T*>(smem_raw);                        // [2 buffers][2 species][ROWS][W]
    k.f = f; k.out = out; k.h = h; k.inj = inj; k.P = P; k.g = g;
    k.hc = HC > 0 ? HC : hc_rt;
    k.wy = threadIdx.x / WAVE;
    const int lane = threadIdx.x % WAVE;
    k.x0 = lane * VEC;
    const int ytiles = g.n1 / TY;
    const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
    const int yt = id % ytiles, zci = id / ytiles;
    k.y = yt * TY + k.wy;
    const int z0 = zci * zc;
    k.z1 = min(z0 + zc, g.n0);
    // every wave also fetches one of the 4 halo rows (TY == 4 waves <-> 4 halo rows)
    k.hy = wrap(yt * TY + (k.wy < 2 ? k.wy - 2 : TY + k.wy - 2), g.n1);
    k.hrow = k.wy < 2 ? k.wy : TY + k.wy;
    k.rowoff = (long)k.y * S::W + k.x0;
    k.hrowoff = (long)k.hy * S::W + k.x0;
    k.acc_c[0] = k.acc_c[1] = 0.0;
    k.with_mom = with_mom;
    if constexpr (K::MOM) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 10; ++q) k.mom[s][q] = T(0);
    }

    // prologue: planes z0-2 .. z0+5 -> slots 0..7; halo rows / operands of planes z0 .. z0+3 -> slots 0..3
    int zz = g.wrap0 ? wrap(z0 - 2, g.n0) : z0 - 2;
#pragma unroll
    for (int j = 0; j < K::R; ++j) {
        if (z0 - 2 + j <= k.z1 + 1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) k.q[s][j] = ld<T, VEC>(k.plane(f, s, zz) + k.rowoff);
        }
        if (j >= 2 && j < 2 + K::RH && z0 + (j - 2) < k.z1) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                k.hp[s][j - 2] = ld<T, VEC>(k.plane(f, s, zz) + k.hrowoff);
                if constexpr (ADJ) {
                    const long en = g.off + (long)(z0 + j - 2) * g.s0 + k.rowoff;
                    k.ph[s][j - 2] = ld<T, VEC>(h + s * g.ss + en);
                    if (inj) k.pj[s][j - 2] = ld<T, VEC>(inj + s * g.ss + en);
                }
            }
        }
        if (j == 2 + K::RH - 1) k.zh = k.next_plane(zz);           // plane z0+4
        zz = k.next_plane(zz);
    }
    k.zq = zz;                                                     // plane z0+6

    int cur = 0;
    for (int z = z0; z < k.z1; z += K::R) k.template run8<0>(z, cur);

    if constexpr (ADJ) {
        // diffusion-coefficient gradients of this strip over its planes: one reduction per launch
        __syncthreads();
        double* red = reinterpret_cast<double*>(k.lds);
        constexpr int NS = K::MOM ? 22 : 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const double r = wave_sum_to_last(k.acc_c[s]);
            if (lane == REDUCE_LANE) red[k.wy * NS + s] = r;
        }
        if constexpr (K::MOM) {
            if (with_mom) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int q = 0; q < 10; ++q) {
                        const double r = wave_sum_to_last((double)k.mom[s][q]);
                        if (lane == REDUCE_LANE) red[k.wy * NS + 2 + 10 * s + q] = r;
                    }
            }
        }
        __syncthreads();
        const int nsum = (K::MOM && with_mom) ? 22 : 2;
        if ((int)threadIdx.x < nsum) {
            double sum = 0.0;
            for (int w = 0; w < TY; ++w) sum += red[w * NS + threadIdx.x];
            const int slot = threadIdx.x < 2 ? P_COEF + (int)threadIdx.x : P_W + (int)threadIdx.x - 2;
            partials[(long)blockIdx.x * nparams(HC == POLY ? 0 : k.hc) + slot] += sum;
        }
    }
}

}  // namespace pi
