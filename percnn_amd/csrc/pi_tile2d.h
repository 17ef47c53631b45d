// pi_tile2d.h -- temporally blocked 2D Pi-block kernels (gfx950).
//
// A 512^2 two-species state is 2 MiB: one step moves 4 MiB and ~80 MFLOP, i.e. well under a
// microsecond of chip time, while a dependent kernel boundary costs 1.5-1.9 us and a cold first
// load another ~1 us.  One launch per step is therefore latency-bound (measured 3.3-4.2 us/step).
// These kernels advance K steps per launch: a workgroup stages its BX x BY tile plus a 2K-wide
// periodic halo of the state in LDS (160 KiB/CU on CDNA4), recomputes the shrinking halo ring
// redundantly (region (B+4(K-m-1))^2 at sub-step m) and still writes EVERY intermediate frame of
// its own tile to the trajectory -- the API returns all steps (train_2drd.py:187-188).
//   * stencil reads: each lane computes a strip of 4 x-points from 12 eight-byte LDS reads per species
//   * global traffic: 16-byte coalesced loads of the window, 16-byte coalesced stores of the tile
//   * one __syncthreads per sub-step (ping-pong state buffers)
//   * per-point arithmetic and its order are IDENTICAL to the direct kernels (bit-equal results)
// The adjoint flavour keeps the adjoint state in LDS and streams h_{t-1}, dL/dout_{t-1} pointwise
// from HBM (they need no neighbours); diffusion-coefficient gradients are accumulated in registers
// across the K sub-steps and reduced once per launch.
#pragma once
#include <type_traits>
#include "pi_device.h"

namespace pi {

#ifndef PI_TILE_ADJ_PIPE
#define PI_TILE_ADJ_PIPE 1
#endif
#ifndef PI_PIN_MOMENTS
#define PI_PIN_MOMENTS 1
#endif
#ifndef PI_FWD_IDLE_STORE
#define PI_FWD_IDLE_STORE 1
#endif
#ifndef PI_FWD_IDLE_MAX
#define PI_FWD_IDLE_MAX 4
#endif
#ifndef PI_PERSIST_GEO
#define PI_PERSIST_GEO 1
#endif
#ifndef PI_PERSIST_LAUNDER
#define PI_PERSIST_LAUNDER 0
#endif
#ifndef PI_PERSIST_OPAQUE_TID
#define PI_PERSIST_OPAQUE_TID 1
#endif


#ifdef PI_TILE_TIMING
// debug build only: per-workgroup s_memtime stamps {start, window loaded, after each sub-step (compute, store issued), end}
__device__ long long pi_tile_stamps[4096 * 16];
__device__ long long pi_tile_wave_stamps[256 * 16 * 16];      // [block < 256][wave < 16][slot]: per-wave view of the same stamps
#define PI_STAMP(i) do { if (threadIdx.x == 0) pi_tile_stamps[blockIdx.x * 16 + (i)] = wall_clock64();                      \
                         if (threadIdx.x % 64 == 0 && blockIdx.x < 256)                                                      \
                             pi_tile_wave_stamps[(blockIdx.x * 16 + threadIdx.x / 64) * 16 + (i)] = wall_clock64(); } while (0)
// slot 14 keeps the END stamp of the previous launch of this block id: start - max(previous ends) = the launch boundary
#define PI_STAMP_PREV() do { if (threadIdx.x == 0) pi_tile_stamps[blockIdx.x * 16 + 14] = pi_tile_stamps[blockIdx.x * 16 + 15]; } while (0)
#else
#define PI_STAMP(i) do { } while (0)
#define PI_STAMP_PREV() do { } while (0)
#endif

#ifdef PI_PERSIST_STAMPS
// debug build only (tools/persist_dev.hip): per-wave 100 MHz stamps of ONE group of the persistent sweeps
__device__ long long pi_persist_stamps[256 * 8 * 16];
#define PI_PSTAMP(i) do { if (grp == PI_PERSIST_STAMPS && threadIdx.x % 64 == 0 && blockIdx.x < 256)                      \
                              pi_persist_stamps[(blockIdx.x * 8 + threadIdx.x / 64) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define PI_PSTAMP(i) do { } while (0)
#endif

template <int K, int BX, int BY>
struct Tile {
    static constexpr int LX = BX + 4 * K, LY = BY + 4 * K;
    static constexpr int PLANE = LX * LY;
    static_assert(LX % 4 == 0 && PLANE % 4 == 0, "quad-aligned rows");
    static constexpr int region_w(int m) { return LX - 4 * (m + 1); }
    static constexpr int region_h(int m) { return LY - 4 * (m + 1); }
    static constexpr int region_n(int m) { return region_w(m) * region_h(m); }
};

// bytes of the two ping-pong state buffers (both species) incl. their alignment pads, rounded up to 16
template <typename T, int K, int BX, int BY>
__host__ __device__ constexpr size_t tile_state_bytes()
{
    return ((size_t)4 * Tile<K, BX, BY>::PLANE * sizeof(T) + 32 + 15) / 16 * 16;
}

// LDS add without a return value (`ds_add_f64`): the caller is the only lane that touches the address
__device__ __forceinline__ void lds_add_f64(double* p, double v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct TileGeom {
    int H, W;          // grid
    long ss;           // species stride = H*W
    int tiles_x;       // ceil(W / BX)
    // XCD-aware block -> tile map: block b runs on XCD b % 8 (private L2), so each XCD is handed ONE contiguous
    // rw x rh rectangle of tiles and the halo rings of neighbouring tiles are fetched from the fabric once per XCD
    // instead of once per tile (the sweep moved 1.7x its algorithmic bytes with the identity map).  rx == 0: identity.
    int rx, rw, rh;    // rectangles per row of rectangles, rectangle width / height in tiles
    LossInj loss;      // adjoint kernel: what the frames behind `gframes` mean (pi_device.h; mode 1: gframes = the trajectory)
};

__device__ __forceinline__ int tile_of_block(int b, const TileGeom& g)
{
    if (g.rx == 0) return b;
    const int xcd = b % NXCD, j = b / NXCD;
    const int ty = (xcd / g.rx) * g.rh + j / g.rw, tx = (xcd % g.rx) * g.rw + j % g.rw;
    return ty * g.tiles_x + tx;
}

// window coordinates lie in [-2K, n + BX + 2K): one conditional add / subtract suffices for n >= BX + 2K (checked by the host)
__device__ __forceinline__ int wrap1(int g, int n) { return g < 0 ? g + n : (g >= n ? g - n : g); }

// LDS read width (fp32): a strip starts at window column 4*rc + 2*(M+1), and lanes 16 B apart that read 8 B each touch
// only half of the banks per pass.  Even sub-steps read buffer 0 at columns = 2 (mod 4), odd sub-steps read buffer 1 at
// columns = 0 (mod 4), so buffer 0 is shifted by two floats: every strip origin is then 16-byte aligned in the buffer
// it is READ from and the stencil rows come in as 16-byte reads (5 x b128 + 2 x b64 instead of 12 x b64 per species and
// strip).  Measured at 512^2: forward 8.54 -> 8.22, sweep 12.51 -> 12.20 us per K=4 launch, SQ_LDS_IDX_ACTIVE -6.5 %.
// (The LDS is not what bounds these kernels: a what-if build that dropped the four neighbour-row reads altogether --
// two thirds of the stencil's LDS bytes -- only gained another 0.6 / 0.8 us.)
template <typename T> struct lds_pad0 { static constexpr int value = sizeof(T) == 4 ? 2 : 0; };
template <typename T> struct lds_pad1 { static constexpr int value = sizeof(T) == 4 ? 4 : 0; };   // keeps buffer 1 16-B aligned

// stage the (LY x LX) periodic window of both species of `src` into buf[2][LY][LX].
// issue() puts all global loads in flight, commit() writes them to LDS: a load -> wait -> write loop costs one full
// memory round trip per trip (3 trips = +0.8 us per launch at 512^2, measured from the device timeline), and the
// adjoint kernel issues its first operand loads between the two.
template <typename T, int K, int BX, int BY, int NT>
struct WindowLoader {
    using TL = Tile<K, BX, BY>;
    static constexpr int VEC = vec_width<T>::value;
    static constexpr int LXV = TL::LX / VEC;
    static constexpr int N = 2 * TL::LY * LXV;
    static constexpr int TRIPS = (N + NT - 1) / NT;
    Pack<T, VEC> p[TRIPS];
    int dst[TRIPS];
    __device__ __forceinline__ void issue(const T* __restrict__ src, const TileGeom& g, int ty0, int tx0)
    {
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) {
            const int i = threadIdx.x + q * NT;
            dst[q] = -1;
            if (i < N) {
                const int s = i / (TL::LY * LXV);
                const int r = i - s * (TL::LY * LXV);
                const int ly = r / LXV, c = r - ly * LXV;
                const int gy = wrap1(ty0 - 2 * K + ly, g.H);
                const int gx = wrap1(tx0 - 2 * K + c * VEC, g.W);
                p[q] = ld<T, VEC>(src + s * g.ss + (long)gy * g.W + gx);
                dst[q] = s * TL::PLANE + ly * TL::LX + c * VEC;
            }
        }
    }
    __device__ __forceinline__ void commit(T* buf) const    // buf = buffer 0 (8-byte aligned only when padded)
    {
#pragma unroll
        for (int q = 0; q < TRIPS; ++q)
            if (dst[q] >= 0) {
                if constexpr (lds_pad0<T>::value != 0) {
                    st<T, 2>(buf + dst[q], Pack<T, 2>{{p[q].v[0], p[q].v[1]}});
                    st<T, 2>(buf + dst[q] + 2, Pack<T, 2>{{p[q].v[2], p[q].v[3]}});
                } else {
                    st<T, VEC>(buf + dst[q], p[q]);
                }
            }
    }
};

template <typename T, int K, int BX, int BY, int NT>
__device__ __forceinline__ void tile_load(const T* __restrict__ src, const TileGeom& g, int ty0, int tx0, T* buf)
{
    WindowLoader<T, K, BX, BY, NT> w;
    w.issue(src, g, ty0, tx0);
    w.commit(buf);
}

// write the BX x BY centre of buf to frame `dst`
// WT = false: plain (write-back) stores -- the resident forwards: nobody reads a frame from memory before the launch ends
template <typename T, int K, int BX, int BY, int NT, bool PADDED = false, bool WT = true>
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
        if (ty0 + y >= g.H || tx0 + c * VEC >= g.W) continue;          // partial edge tile of a ragged grid
        const T* src = buf + s * TL::PLANE + (2 * K + y) * TL::LX + 2 * K + c * VEC;
        Pack<T, VEC> p;
        if constexpr (PADDED && lds_pad0<T>::value != 0) {
            const Pack<T, 2> a = ld<T, 2>(src), b = ld<T, 2>(src + 2);
            p.v[0] = a.v[0]; p.v[1] = a.v[1]; p.v[2] = b.v[0]; p.v[3] = b.v[1];
        } else {
            p = ld<T, VEC>(src);
        }
        if constexpr (WT) st_frame_wt<T, VEC>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC, p);
        else *reinterpret_cast<Pack<T, VEC>*>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC) = p;
    }
}

// radius-2 star for a strip of 4 consecutive x points starting at (ly, lx), lx even: 12 eight-byte
// (fp64: sixteen-byte) LDS reads per species instead of 36 four-byte ones -- the poly-mode tile kernels
// are LDS-instruction-bound otherwise.  Tap order as pi::star (axis 0 = y first, then x).
template <typename T, int LX, int FLIP>
__device__ __forceinline__ void lds_star4(const T* pl, int ly, int lx, const T* __restrict__ P, T (&ctr)[4], T (&lap)[4])
{
    const T* c = pl + ly * LX + lx;                       // fp32: 16-byte aligned (lds_pad0)
    T win[8];                                             // x = lx-2 .. lx+5 of the centre row
    if constexpr (sizeof(T) == 4) {
        const Pack<T, 2> l = ld<T, 2>(c - 2), r = ld<T, 2>(c + 4);
        const Pack<T, 4> m = ld<T, 4>(c);
        win[0] = l.v[0]; win[1] = l.v[1]; win[6] = r.v[0]; win[7] = r.v[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) win[2 + i] = m.v[i];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Pack<T, 2> p = ld<T, 2>(c - 2 + 2 * j);
            win[2 * j] = p.v[0];
            win[2 * j + 1] = p.v[1];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { ctr[i] = win[2 + i]; lap[i] = P[P_C0] * win[2 + i]; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        T n[4];
        if constexpr (sizeof(T) == 4) {
            const Pack<T, 4> a = ld<T, 4>(c + k * LX);
#pragma unroll
            for (int i = 0; i < 4; ++i) n[i] = a.v[i];
        } else {
            const Pack<T, 2> a = ld<T, 2>(c + k * LX), b = ld<T, 2>(c + k * LX + 2);
            n[0] = a.v[0]; n[1] = a.v[1]; n[2] = b.v[0]; n[3] = b.v[1];
        }
        const T w = P[P_TAPS + t];
#pragma unroll
        for (int i = 0; i < 4; ++i) lap[i] = fma_(w, n[i], lap[i]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        const T w = P[P_TAPS + 4 + t];
#pragma unroll
        for (int i = 0; i < 4; ++i) lap[i] = fma_(w, win[2 + i + k], lap[i]);
    }
}

template <typename T>
__device__ __forceinline__ void lds_store4(T* p, const T (&v)[4])
{
    st<T, 2>(p, Pack<T, 2>{{v[0], v[1]}});
    st<T, 2>(p + 2, Pack<T, 2>{{v[2], v[3]}});
}

// ---- explicit 2-wide arithmetic ---------------------------------------------------------------------------------
// The adjoint sub-step is VALU-bound and hipcc's SLP vectoriser left it entirely scalar (0 v_pk_* against 900 scalar
// fp32 ops per launch, while it packs the forward body); written on 2-vectors the same IEEE operations in the same
// order issue as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (fp64: scalarised again by the compiler, same results).
template <typename T> using V2 = T __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ V2<T> vs(T x) { return V2<T>{x, x}; }
template <typename T> __device__ __forceinline__ V2<T> vfma(V2<T> a, V2<T> b, V2<T> c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T> __device__ __forceinline__ V2<T> ldv2(const T* p)
{
    const Pack<T, 2> q = ld<T, 2>(p);
    return V2<T>{q.v[0], q.v[1]};
}
template <typename T> __device__ __forceinline__ void stv2(T* p, V2<T> v) { st<T, 2>(p, Pack<T, 2>{{v.x, v.y}}); }
// elements idx, idx+1 of the 8-point row window W[4]
template <typename T> __device__ __forceinline__ V2<T> win_pair(const V2<T> (&W)[4], int idx)
{
    return (idx & 1) ? V2<T>{W[idx / 2].y, W[idx / 2 + 1].x} : W[idx / 2];
}

// lds_star4 on 2-vectors: ctr/lap[0] = points 0,1 of the strip, [1] = points 2,3; identical operation order
template <typename T, int LX, int FLIP>
__device__ __forceinline__ void lds_star4v_at(const T* c, const T* __restrict__ P, V2<T> (&ctr)[2], V2<T> (&lap)[2]);
template <typename T, int LX, int FLIP>
__device__ __forceinline__ void lds_star4v(const T* pl, int ly, int lx, const T* __restrict__ P, V2<T> (&ctr)[2], V2<T> (&lap)[2])
{
    lds_star4v_at<T, LX, FLIP>(pl + ly * LX + lx, P, ctr, lap);
}
// c = address of the strip's first point (fp32: 16-byte aligned, lds_pad0); cf(i) = the pair {P[i], P[i]}
template <typename T, int LX, int FLIP, typename CF>
__device__ __forceinline__ void lds_star4v_cf(const T* c, CF cf, V2<T> (&ctr)[2], V2<T> (&lap)[2])
{
    V2<T> W[4];
    if constexpr (sizeof(T) == 4) {
        W[0] = ldv2(c - 2);
        const Pack<T, 4> m = ld<T, 4>(c);
        W[1] = V2<T>{m.v[0], m.v[1]}; W[2] = V2<T>{m.v[2], m.v[3]};
        W[3] = ldv2(c + 4);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) W[j] = ldv2(c - 2 + 2 * j);
    }
    ctr[0] = W[1]; ctr[1] = W[2];
    const V2<T> c0 = cf(P_C0);
    lap[0] = c0 * ctr[0]; lap[1] = c0 * ctr[1];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        V2<T> a, b;
        if constexpr (sizeof(T) == 4) {
            const Pack<T, 4> n = ld<T, 4>(c + k * LX);
            a = V2<T>{n.v[0], n.v[1]}; b = V2<T>{n.v[2], n.v[3]};
        } else {
            a = ldv2(c + k * LX); b = ldv2(c + k * LX + 2);
        }
        const V2<T> w = cf(P_TAPS + t);
        lap[0] = vfma(w, a, lap[0]);
        lap[1] = vfma(w, b, lap[1]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        const V2<T> w = cf(P_TAPS + 4 + t);
        lap[0] = vfma(w, win_pair(W, 2 + k), lap[0]);
        lap[1] = vfma(w, win_pair(W, 4 + k), lap[1]);
    }
}
template <typename T, int LX, int FLIP>
__device__ __forceinline__ void lds_star4v_at(const T* c, const T* __restrict__ P, V2<T> (&ctr)[2], V2<T> (&lap)[2])
{
    lds_star4v_cf<T, LX, FLIP>(c, [P](int i) { return vs(P[i]); }, ctr, lap);
}

// the same for the two points at c (a HALF-strip: c is 8-byte aligned); operations and their order per point as lds_star4v_cf
template <typename T> __device__ __forceinline__ V2<T> win_pair3(const V2<T> (&W)[3], int idx)
{
    return (idx & 1) ? V2<T>{W[idx / 2].y, W[idx / 2 + 1].x} : W[idx / 2];
}
template <typename T, int LX, int FLIP, typename CF>
__device__ __forceinline__ void lds_star2v_cf(const T* c, CF cf, V2<T>& ctr, V2<T>& lap)
{
    const V2<T> W[3] = {ldv2(c - 2), ldv2(c), ldv2(c + 2)};
    ctr = W[1];
    lap = cf(P_C0) * ctr;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        lap = vfma(cf(P_TAPS + t), ldv2(c + k * LX), lap);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        lap = vfma(cf(P_TAPS + 4 + t), win_pair3(W, 2 + k), lap);
    }
}

// pi::poly_r on 2-vectors (same operations in the same order)
template <typename T>
__device__ __forceinline__ V2<T> poly_r_v(const T* __restrict__ c, V2<T> u, V2<T> v)
{
    const V2<T> A0 = vfma(v, vfma(v, vfma(v, vs(c[9]), vs(c[5])), vs(c[2])), vs(c[0]));
    const V2<T> A1 = vfma(v, vfma(v, vs(c[8]), vs(c[4])), vs(c[1]));
    const V2<T> A2 = vfma(v, vs(c[7]), vs(c[3]));
    return vfma(u, vfma(u, vfma(u, vs(c[6]), A2), A1), A0);
}

template <typename T>
__device__ __forceinline__ void poly_dr_v(const T* __restrict__ c, V2<T> u, V2<T> v, V2<T>& ru, V2<T>& rv)
{
    const V2<T> A1 = vfma(v, vfma(v, vs(c[8]), vs(c[4])), vs(c[1]));
    const V2<T> A2x2 = vfma(v, vs(T(2) * c[7]), vs(T(2) * c[3]));
    ru = vfma(u, vfma(u, vs(T(3) * c[6]), A2x2), A1);
    const V2<T> B0 = vfma(v, vfma(v, vs(T(3) * c[9]), vs(T(2) * c[5])), vs(c[2]));
    const V2<T> B1 = vfma(v, vs(T(2) * c[8]), vs(c[4]));
    rv = vfma(u, vfma(u, vs(c[7]), B1), B0);
}

// The Jacobian's coefficient multiples as ready pairs held in vector registers (persistent split sweep).  gfx950 has no scalar
// float multiply: 2 c / 3 c are VALU products of a uniform value, and every use of one in a packed FMA needs the pair {x, x}
// built by a v_mov -- per pass, in a loop bound by instruction issue.  Entry j of species s: 0 2c7, 1 2c3, 2 3c6, 3 3c9, 4 2c5,
// 5 2c8, 6 c4; PI_JAC_MASK says which are held (the others are formed as before; all seven since the two parts of a mixed
// pass share one body and the registers are there -- masks measured: profiles/r04_persist_issue_trim.txt).  Same single
// multiplication: bit-identical.
#ifndef PI_JAC_MASK
#define PI_JAC_MASK 0x7F
#endif
#ifndef PI_STEN_MASK
#define PI_STEN_MASK 0xFFF
#endif
// ... and `st`: the pairs of dt (0), the diffusion coefficients (1, 2), c0 (3) and the eight taps (4..11) -- P[0..11] in P's own
// order; PI_STEN_MASK says which are held
template <typename T> struct JacPairs { V2<T> m[2][7]; V2<T> st[12]; };
// (float32 only: a float64 pair is four registers -- 26 held pairs would be 104 of the 256; float64 forms them where they are used)
template <typename T> struct held_masks { static constexpr int jac = sizeof(T) == 4 ? (PI_JAC_MASK) : 0, sten = sizeof(T) == 4 ? (PI_STEN_MASK) : 0; };
template <typename T>
__device__ __forceinline__ void jac_pairs_load(JacPairs<T>& jp, const T* __restrict__ P)
{
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const T* c = P + P_W + 10 * s;
        const T x[7] = {T(2) * c[7], T(2) * c[3], T(3) * c[6], T(3) * c[9], T(2) * c[5], T(2) * c[8], c[4]};
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            jp.m[s][j] = vs(x[j]);
            if constexpr (sizeof(T) == 4) {
                if ((held_masks<T>::jac >> j) & 1) asm volatile("" : "+v"(jp.m[s][j]));        // a register pair from here on, not a recipe
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        jp.st[i] = vs(P[i]);
        if constexpr (sizeof(T) == 4) {
            if ((held_masks<T>::sten >> i) & 1) asm volatile("" : "+v"(jp.st[i]));
        }
    }
}
template <typename T>
__device__ __forceinline__ void poly_dr_v_j(const T* __restrict__ c, const V2<T> (&j)[7], V2<T> u, V2<T> v, V2<T>& ru, V2<T>& rv)
{
    constexpr int MSK = held_masks<T>::jac;
    const V2<T> k0 = (MSK & 1) ? j[0] : vs(T(2) * c[7]), k1 = (MSK & 2) ? j[1] : vs(T(2) * c[3]), k2 = (MSK & 4) ? j[2] : vs(T(3) * c[6]);
    const V2<T> k3 = (MSK & 8) ? j[3] : vs(T(3) * c[9]), k4 = (MSK & 16) ? j[4] : vs(T(2) * c[5]), k5 = (MSK & 32) ? j[5] : vs(T(2) * c[8]);
    const V2<T> k6 = (MSK & 64) ? j[6] : vs(c[4]);
    const V2<T> A1 = vfma(v, vfma(v, vs(c[8]), k6), vs(c[1]));
    const V2<T> A2x2 = vfma(v, k0, k1);
    ru = vfma(u, vfma(u, k2, A2x2), A1);
    const V2<T> B0 = vfma(v, vfma(v, k3, k4), vs(c[2]));
    const V2<T> B1 = vfma(v, k5, k6);
    rv = vfma(u, vfma(u, vs(c[7]), B1), B0);
}

// ------------------------------------------------------------------------------------------------
// forward: frames t+1 .. t+K from frame t
// ------------------------------------------------------------------------------------------------
template <typename T, int HC, int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void fwd_substep(T* cur, T* nxt, const T* __restrict__ P)
{
    using TL = Tile<K, BX, BY>;
    constexpr int RW4 = TL::region_w(M) / 4, RN4 = TL::region_n(M) / 4, O = 2 * (M + 1);
    constexpr int PT = (RN4 + NT - 1) / NT;              // strips of 4 points per lane
    const T dt = P[P_DT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        int idx = threadIdx.x + q * NT;
        // Waves with no strip left in this (shrinking) region skip it: they share a SIMD with a working wave and the
        // sub-steps are VALU-issue-bound (per-wave timeline: the second wave of a SIMD finishes 0.45 us after the first),
        // so a wave that merely repeats the last strip doubles that SIMD's work.  Sub-step K-1 of a 32 x 32 tile with 512
        // threads is exactly four waves of strips -> one working wave per SIMD.
        if (((int)threadIdx.x & ~(WAVE - 1)) + q * NT >= RN4) continue;
        if (idx >= RN4) idx = RN4 - 1;                     // tail lanes of a partial wave redo the last strip (identical values)
        const int ry = idx / RW4, rc = idx - ry * RW4;
        const int off = (ry + O) * TL::LX + 4 * rc + O;
        T u[4], v[4], lap[2][4];
        lds_star4<T, TL::LX, +1>(cur, ry + O, 4 * rc + O, P, u, lap[0]);
        lds_star4<T, TL::LX, +1>(cur + TL::PLANE, ry + O, 4 * rc + O, P, v, lap[1]);
        auto emit = [&](int s, const T (&rr)[4]) {
            const T coef = P[P_COEF + s];
            T o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const T lp = s == 0 ? lap[0][i] : lap[1][i];
                const T res = coef * lp + rr[i];
                const T inc = res * dt;
                o[i] = (s == 0 ? u[i] : v[i]) + inc;
            }
            lds_store4(nxt + s * TL::PLANE + off, o);
        };
        if constexpr (HC == POLY) {
            // 9 FMAs per species: unrolled, so the 20 coefficients are loop-invariant scalar loads the compiler hoists
            // (rolled, every iteration re-issued them and waited on lgkmcnt(0))
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                T rr[4];
                const T* c = P + P_W + 10 * s;
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[i] = poly_r(c, u[i], v[i]);
                emit(s, rr);
            }
        } else {
            // species / hidden-channel loops stay rolled (small I$-resident body, scalars prefetched)
#pragma clang loop unroll(disable)
            for (int s = 0; s < 2; ++s) {
                T rr[4];
                const T* W = P + P_W + s * species_block(HC);
#pragma unroll
                for (int i = 0; i < 4; ++i) rr[i] = W[10 * HC];
                W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
                for (int j = 0; j < HC; ++j) {
                    const W10<T> c = nx;
                    if (j + 1 < HC) nx = load_w10(W + 10 * (j + 1));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const T a1 = fma_(c.w[0], u[i], fma_(c.w[1], v[i], c.w[2]));
                        const T a2 = fma_(c.w[3], u[i], fma_(c.w[4], v[i], c.w[5]));
                        const T a3 = fma_(c.w[6], u[i], fma_(c.w[7], v[i], c.w[8]));
                        rr[i] = fma_(c.w[9], (a1 * a2) * a3, rr[i]);
                    }
                }
                emit(s, rr);
            }
        }
    }
}

// first lane of the waves that own no strip in sub-step M (NT: every wave works)
template <int K, int BX, int BY, int NT, int M, int CHUNKS, int PTS = 4>      // PTS: points per lane (2: half-strips)
constexpr int fwd_first_idle_lane()
{
    constexpr int rn4 = Tile<K, BX, BY>::region_n(M) / PTS;
    constexpr int busy = (rn4 + WAVE - 1) / WAVE * WAVE;
    // only where the idle waves get away with <= PI_FWD_IDLE_MAX chunks per lane: ONE idle wave storing a whole frame (8 chunks
    // per lane, 16 in float64) takes longer than the sub-step it hides behind and becomes the critical path (measured: 512^2
    // forward 1.70 -> 1.86 us per step, lambda-omega 2.78 -> 4.04)
    return (busy < NT && CHUNKS <= PI_FWD_IDLE_MAX * (NT - busy)) ? busy : NT;
}

// tile_store by the lanes from FIRST on only (whole waves): frame M of the launch, written while the other waves compute
template <typename T, int K, int BX, int BY, int NT, int FIRST, bool PADDED, bool WT = true>
__device__ __forceinline__ void tile_store_by_idle(const T* buf, T* __restrict__ dst, const TileGeom& g, int ty0, int tx0)
{
    using TL = Tile<K, BX, BY>;
    constexpr int VEC = vec_width<T>::value;
    constexpr int BXV = BX / VEC;
    constexpr int N = 2 * BY * BXV;
    constexpr int LANES = NT - FIRST;
    for (int i = (int)threadIdx.x - FIRST; i < N; i += LANES) {
        const int s = i / (BY * BXV);
        const int r = i - s * (BY * BXV);
        const int y = r / BXV, c = r - y * BXV;
        if (ty0 + y >= g.H || tx0 + c * VEC >= g.W) continue;          // partial edge tile of a ragged grid
        const T* src = buf + s * TL::PLANE + (2 * K + y) * TL::LX + 2 * K + c * VEC;
        Pack<T, VEC> p;
        if constexpr (PADDED && lds_pad0<T>::value != 0) {
            const Pack<T, 2> a = ld<T, 2>(src), b = ld<T, 2>(src + 2);
            p.v[0] = a.v[0]; p.v[1] = a.v[1]; p.v[2] = b.v[0]; p.v[3] = b.v[1];
        } else {
            p = ld<T, VEC>(src);
        }
        if constexpr (WT) st_frame_wt<T, VEC>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC, p);
        else *reinterpret_cast<Pack<T, VEC>*>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC) = p;
    }
}

// Who stores frame M (= the level sub-step M reads, M >= 1)?  The regions shrink -- (B + 4 (K - M - 1))^2 points at sub-step M --,
// so from sub-step 1 on whole waves have no strip (32 x 32 tile, 512 lanes: one wave in sub-step 1, two in 2, four in 3), and
// the sub-steps are VALU-issue-bound on the waves that do.  Round 4: those idle waves store frame M from the buffer the busy
// waves are READING (LDS reads and global stores issue next to the other waves' VALU work) instead of all waves storing it
// between the barrier and the next sub-step, where the LDS round trip and the store issue sat on every wave's critical path
// (device timeline, round 1: 0.3-0.4 us of each 1.1 us sub-step).  Frame K is stored by everybody at the end, as before.
// `STORE_AHEAD`: sub-step M has idle waves (compile time; otherwise the all-waves store after the barrier stays).
template <typename T, int K, int BX, int BY>
__device__ __forceinline__ void fwd_strip_geo(const T* cur, T* nxt, const T* __restrict__ P, unsigned w);
template <typename T, int K, int BX, int BY>
__device__ __forceinline__ void fwd_half_strip_geo(const T* cur, T* nxt, const T* __restrict__ P, unsigned w);
// HALFS (round 6, resident small-tile forward; float32): the lanes work on half-strips, `geo`[K] = the lane's own geometry words
// (fwd_half_word<PART_FULL>, held in registers by the caller).
// LAST_STORE = false: frame K stays in LDS (the resident small-tile forward stores it after its hand-over has been started);
// GEO: the lane's strip of each sub-step comes as a geometry word from an LDS table ([K][NT], persist_geo_word; one strip per
// lane) instead of being derived from the lane id in every sub-step of every group; WT = false: plain frame stores
template <typename T, int HC, int K, int BX, int BY, int NT, int M, bool LAST_STORE = true, bool GEO = false, bool WT = true,
          bool HALFS = false>
__device__ __forceinline__ void fwd_substeps(T* b0, T* b1, T* __restrict__ frames, long frame_stride, const TileGeom& g,
                                             int ty0, int tx0, const T* __restrict__ P, const unsigned* geo = nullptr)
{
    T* cur = (M & 1) ? b1 : b0;
    T* nxt = (M & 1) ? b0 : b1;
    constexpr int CHUNKS = 2 * BY * BX / vec_width<T>::value;
    constexpr int PTS = HALFS ? 2 : 4;
    constexpr int IDLE = fwd_first_idle_lane<K, BX, BY, NT, M, CHUNKS, PTS>();
    constexpr bool STORE_HERE = PI_FWD_IDLE_STORE && M >= 1 && IDLE < NT;       // frame M: by this sub-step's idle waves
    if constexpr (STORE_HERE) {
        if ((int)threadIdx.x >= IDLE)
            tile_store_by_idle<T, K, BX, BY, NT, IDLE, ((M - 1) & 1) != 0, WT>(cur, frames + (long)M * frame_stride, g, ty0, tx0);
    }
    if constexpr (HALFS) {
        static_assert(HC == POLY && Tile<K, BX, BY>::region_n(0) / 2 <= NT, "one half-strip per lane, pre-contracted block");
        fwd_half_strip_geo<T, K, BX, BY>(cur, nxt, P, geo[M]);
    } else if constexpr (GEO) {
        static_assert(HC == POLY && Tile<K, BX, BY>::region_n(0) / 4 <= NT, "one strip per lane, pre-contracted block");
        fwd_strip_geo<T, K, BX, BY>(cur, nxt, P, geo[M * NT + (int)threadIdx.x]);
    } else {
        fwd_substep<T, HC, K, BX, BY, NT, M>(cur, nxt, P);
    }
    PI_STAMP(2 + 2 * M);
    lds_barrier();                                         // do not drain the previous frame's global stores
    // frame M + 1: left to the idle waves of the next sub-step if it has any, else stored now by everybody
    constexpr bool NEXT_STORES = PI_FWD_IDLE_STORE && M + 1 < K && fwd_first_idle_lane<K, BX, BY, NT, (M + 1 < K ? M + 1 : M), CHUNKS, PTS>() < NT;
    if constexpr (!NEXT_STORES && (M + 1 < K || LAST_STORE))
        tile_store<T, K, BX, BY, NT, (M & 1) != 0, WT>(nxt, frames + (long)(M + 1) * frame_stride, g, ty0, tx0);
    PI_STAMP(3 + 2 * M);
    if constexpr (M + 1 < K)
        fwd_substeps<T, HC, K, BX, BY, NT, M + 1, LAST_STORE, GEO, WT, HALFS>(b0, b1, frames, frame_stride, g, ty0, tx0, P, geo);
}

template <typename T, int HC, int K, int BX, int BY, int NT>
__global__ void __launch_bounds__(NT)
pi_fwd2d_tile_kernel(T* __restrict__ frames /* frame t; t+1..t+K are written */, long frame_stride,
                     const T* __restrict__ P, TileGeom g)
{
    using TL = Tile<K, BX, BY>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int ty0 = (tile / g.tiles_x) * BY, tx0 = (tile % g.tiles_x) * BX;
    PI_STAMP_PREV();
    PI_STAMP(0);
    tile_load<T, K, BX, BY, NT>(frames, g, ty0, tx0, b0);
    // (holding the 36-entry block in registers across the sub-steps, which pays in the resident kernels, LOSES here: 6.68 -> 7.7 us
    // per launch of four steps -- a short launch waits for the 36 scalar loads before its first sub-step instead of under it; round 6)
    __syncthreads();
    PI_STAMP(1);
    fwd_substeps<T, HC, K, BX, BY, NT, 0>(b0, b1, frames, frame_stride, g, ty0, tx0, P);
    PI_STAMP(15);
}

// ------------------------------------------------------------------------------------------------
// physics-residual LOSS of a 2D trajectory on the tile machinery (generic flavour: pi_residual_sq_kernel<GRAD = false>;
// reference: loss_gen / get_phy_Loss, train_2drd.py:270-353 -- evaluated every training iteration as a monitor, :405):
//   R_s(f, x) = coef_s * Lap(h_f)_s + r_s(h_f) - (h_{f+1,s} - h_{f,s}) / dt ,   partial = sum w * R^2   (Q: the TRUE equation)
// A workgroup stages the BX x BY tile of frame f plus its 2-wide periodic ring in LDS once (K = 1 window) and every lane forms
// the residual of one 4-point strip from it with lds_star4 -- the taps in pi::star's order, i.e. the generic kernel's R bit for
// bit -- and the strip of frame f + 1 it requested with the window.  Workgroup (x, y) walks frames y, y + gridDim.y, ...
// The generic kernel spends, per 512^2 frame, 0.6 us issuing 316 unpacked VALU instructions per wave, 0.5 us on 16 vector-L1
// requests per chunk and 0.26 us on HBM, one after the other (1.38 us); here: 5 requests per lane and the packed strip body.
// ------------------------------------------------------------------------------------------------
template <typename T, int BX, int BY, int NT>
__global__ void __launch_bounds__(NT)
pi_res2d_tile_kernel(const T* __restrict__ traj, double* __restrict__ partials, const T* __restrict__ Q, TileGeom g,
                     long frame_stride, int nframes, int weighted)
{
    // window: BY + 4 rows (2 above, 2 below) x BX + 8 columns -- FOUR halo columns per side although the star needs two, so
    // that every 16-byte piece of the window starts at a multiple of 4 columns and never straddles the periodic wrap (the
    // K-step kernels' windows start at tx0 - 2K with K even; a window starting at tx0 - 2 would)
    constexpr int VEC = vec_width<T>::value;
    constexpr int LX = BX + 8, LY = BY + 4, PLANE = LX * LY, LXV = LX / VEC;
    constexpr int NLD = 2 * LY * LXV, TRIPS = (NLD + NT - 1) / NT;
    constexpr int RW4 = BX / 4, RN4 = BX * BY / 4, PT = RN4 / NT;
    static_assert(RN4 % NT == 0 && LX % 4 == 0, "whole strips per lane, 16-byte rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[NT / WAVE];
    T* b0 = reinterpret_cast<T*>(smem_raw);
    const int tile = tile_of_block(blockIdx.x, g);
    const int ty0 = (tile / g.tiles_x) * BY, tx0 = (tile % g.tiles_x) * BX;
    const T dt = Q[P_DT];
    // what a lane loads of the window (the same pieces of every frame) ...
    long woff[TRIPS];
    int wdst[TRIPS];
#pragma unroll
    for (int q = 0; q < TRIPS; ++q) {
        const int i = (int)threadIdx.x + q * NT;
        wdst[q] = -1;
        woff[q] = 0;
        if (i < NLD) {
            const int s = i / (LY * LXV), r = i - s * (LY * LXV);
            const int ly = r / LXV, c = r - ly * LXV;
            const int gy = wrap1(ty0 - 2 + ly, g.H), gx = wrap1(tx0 - 4 + c * VEC, g.W);
            woff[q] = s * g.ss + (long)gy * g.W + gx;
            wdst[q] = s * PLANE + ly * LX + c * VEC;
        }
    }
    // ... and the strips it owns: window position, offset inside a species plane, ownership (edge tiles of a ragged grid)
    int ry[PT], rc[PT];
    long e[PT];
    bool own[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int idx = (int)threadIdx.x + q * NT;
        ry[q] = idx / RW4; rc[q] = idx - ry[q] * RW4;
        own[q] = ty0 + ry[q] < g.H && tx0 + 4 * rc[q] < g.W;
        e[q] = own[q] ? (long)(ty0 + ry[q]) * g.W + tx0 + 4 * rc[q] : 0;
    }
    double acc = 0.0;
    for (int f = (int)blockIdx.y; f < nframes; f += (int)gridDim.y) {
        const T* h = traj + (long)f * frame_stride;
        Pack<T, VEC> wreg[TRIPS];
#pragma unroll
        for (int q = 0; q < TRIPS; ++q) wreg[q] = ld<T, VEC>(h + woff[q]);
        Pack<T, 2> nx[PT][2][2];                           // [strip][species][half]: the strip in frame f + 1
#pragma unroll
        for (int q = 0; q < PT; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                nx[q][s][0] = ld<T, 2>(h + frame_stride + s * g.ss + e[q]);
                nx[q][s][1] = ld<T, 2>(h + frame_stride + s * g.ss + e[q] + 2);
            }
#pragma unroll
        for (int q = 0; q < TRIPS; ++q)
            if (wdst[q] >= 0) st<T, VEC>(b0 + wdst[q], wreg[q]);
        lds_barrier();
        // the strip body on explicit 2-vectors (v_pk_fma_f32 / v_pk_mul_f32): left to itself the compiler issued 292 scalar
        // VALU instructions per strip here, none packed -- the pass is issue-bound (one wave per SIMD and workgroup)
        T part = T(0);
#pragma unroll
        for (int q = 0; q < PT; ++q) {
            V2<T> c2[2][2], lap[2][2];                     // [species][half of the strip]
            lds_star4v<T, LX, +1>(b0, ry[q] + 2, 4 * rc[q] + 4, Q, c2[0], lap[0]);
            lds_star4v<T, LX, +1>(b0 + PLANE, ry[q] + 2, 4 * rc[q] + 4, Q, c2[1], lap[1]);
            const T wrow = (weighted && ty0 + ry[q] == 0) ? T(2) : T(1);
            const V2<T> w0 = V2<T>{(weighted && tx0 + 4 * rc[q] == 0) ? wrow * T(2) : wrow, wrow}, w1 = vs(wrow);
            V2<T> sp = vs(T(0));
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const T* c = Q + P_W + 10 * s;
                const V2<T> coef = vs(Q[P_COEF + s]);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const V2<T> rhs = coef * lap[s][hh] + poly_r_v(c, c2[0][hh], c2[1][hh]);
                    const V2<T> nxv = V2<T>{nx[q][s][hh].v[0], nx[q][s][hh].v[1]};
                    const V2<T> d = nxv - c2[s][hh];
                    const V2<T> r = rhs - V2<T>{d.x / dt, d.y / dt};
                    sp = vfma((hh == 0 ? w0 : w1) * r, r, sp);
                }
            }
            part += own[q] ? sp.x + sp.y : T(0);
        }
        acc += (double)part;
        lds_barrier();                                     // the next frame's window overwrites this one
    }
    acc = wave_sum_to_last(acc);
    if (threadIdx.x % WAVE == REDUCE_LANE) red[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < NT / WAVE; ++w) t += red[w];
        partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// adjoint sweep: adj frames t-1 .. t-K from adj frame t (adjoint state lives in LDS)
//   hframes : trajectory; hframes + (t-1-m)*frame_stride is the state sub-step m linearises about
//   gframes : dL/dtraj;   gframes + (t-1-m)*frame_stride is injected at sub-step m if inj_mask bit m
//   aframes : adjoint trajectory; frame t is read, frames t-1..t-K are written
// ------------------------------------------------------------------------------------------------
// pointwise operands (state h_{t-1-M}, injected dL/dout_{t-1-M}) of one 4-point strip
template <typename T> struct StripOps { T u[4], v[4], ju[4], jv[4]; };

// Which strips of sub-step M's region R_M (side B + 4 (K - M - 1), origin O = 2 (M + 1) in window coordinates) a pass covers:
//   PART_FULL  all of R_M, row-major (the launch-per-group kernels);
//   PART_PYR   the centred square I_M of side B - 4 (M + 1): the points whose value after sub-step M depends on the workgroup's
//              OWN tile only -- computable before the neighbours' halo has arrived (persistent sweep, split flavour);
//   PART_ANN   the rest, R_M \ I_M: a ring 2K points thick (top / bottom bands of whole rows, then 2K-wide side pieces).
// I_M sits on R_M's own grid of 4-point strips (its origin is 2K = 0 mod 4 further in), so every strip keeps the 16-byte
// LDS alignment of lds_pad0 and the arithmetic of a point does not depend on which pass computes it.
enum : int { PART_FULL = 0, PART_PYR = 1, PART_ANN = 2 };

template <int K, int BX, int BY, int M, int PART>
struct StripMap {
    using TL = Tile<K, BX, BY>;
    static constexpr int RW4 = TL::region_w(M) / 4, RH = TL::region_h(M);
    static constexpr int SW4 = (BX - 4 * (M + 1)) / 4, SH = BY - 4 * (M + 1);
    static constexpr int HWD = 2 * K, HW4 = HWD / 4;
    static_assert(PART == PART_FULL || (HWD % 4 == 0 && SW4 > 0 && SH > 0), "split passes: K even, tile larger than 4K");
    static constexpr int BANDS = 2 * HWD * RW4;                            // strips of the top + bottom bands of the ring
    static constexpr int N = PART == PART_FULL ? RW4 * RH : (PART == PART_PYR ? SW4 * SH : BANDS + 2 * HW4 * SH);
    static __device__ __forceinline__ void locate(int idx, int& ry, int& rc)
    {
        if constexpr (PART == PART_FULL) {
            ry = idx / RW4; rc = idx - ry * RW4;
        } else if constexpr (PART == PART_PYR) {
            const int r = idx / SW4;
            ry = HWD + r; rc = HW4 + idx - r * SW4;
        } else {
            if (idx < BANDS) {
                const int j = idx / RW4;
                rc = idx - j * RW4;
                ry = j < HWD ? j : j + SH;
            } else {
                const int k = idx - BANDS, r = k / (2 * HW4), c = k - r * (2 * HW4);
                ry = HWD + r;
                rc = c < HW4 ? c : c + SW4;
            }
        }
    }
};

// element offsets (inside one species plane) of the two halves of the strip a lane owns in sub-step M
struct StripAddr { long e0, e1; };

template <int K, int BX, int BY, int NT, int M, int PART = PART_FULL, int TID0 = 0>
__device__ __forceinline__ StripAddr strip_addr(int q, const TileGeom& g, int ty0, int tx0)
{
    using SM = StripMap<K, BX, BY, M, PART>;
    constexpr int RN4 = SM::N, O = 2 * (M + 1);
    int idx = (int)threadIdx.x - TID0 + q * NT;            // TID0: first lane of the waves that work on this part (mixed passes)
    if (idx >= RN4) idx = RN4 - 1;
    if (idx < 0) idx = 0;
    int ry, rc;
    SM::locate(idx, ry, rc);
    const int ly = ry + O, lx = 4 * rc + O;
    // two 8/16-byte pieces per row: the strip may straddle the periodic wrap
    const int gy = wrap1(ty0 - 2 * K + ly, g.H);
    const int gx0 = wrap1(tx0 - 2 * K + lx, g.W), gx1 = wrap1(tx0 - 2 * K + lx + 2, g.W);
    return StripAddr{(long)gy * g.W + gx0, (long)gy * g.W + gx1};
}

template <int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void strip_addr_table(StripAddr (&sa)[K], const TileGeom& g, int ty0, int tx0)
{
    sa[M] = strip_addr<K, BX, BY, NT, M>(0, g, ty0, tx0);
    asm volatile("" : "+v"(sa[M].e0), "+v"(sa[M].e1));      // computed HERE (kernel prologue), not where it is consumed
    if constexpr (M + 1 < K) strip_addr_table<K, BX, BY, NT, M + 1>(sa, g, ty0, tx0);
}

// sa: the strip's offsets if they were computed ahead (kernel prologue), else nullptr
template <typename T, int K, int BX, int BY, int NT, int M>
__device__ __forceinline__ void adj_load_ops(StripOps<T>& o, int q, const T* __restrict__ hfr, const T* __restrict__ gfr,
                                             const TileGeom& g, int ty0, int tx0, const StripAddr* sa = nullptr)
{
    const StripAddr A = sa ? *sa : strip_addr<K, BX, BY, NT, M>(q, g, ty0, tx0);
    // The sub-steps are VALU-issue-bound (two waves per SIMD: the per-wave timeline shows waves 4-7 finishing 0.45 us after
    // waves 0-3 in every sub-step), so the operand addressing was A/B-measured on one box (us per step, 512^2 backward):
    // scalar frame bases + 32-bit byte offsets (`global_load v, v_off, s[base]`) vs per-lane 64-bit addresses --
    // float64 5.80 -> 5.56, float32 2.96 -> 3.06 (fused) / 3.51 -> 3.68 (split): kept for float64 only.
    Pack<T, 2> a, b, c, d, a2, b2, c2, d2;
    // (the loss-gradient loads are unconditional -- a frame without gradient re-reads the state frame, the values are ignored:
    // with the loads in a branch the compiler cannot count the outstanding requests and makes the window commit wait for
    // these HBM-cold operands too, s_waitcnt vmcnt(4) instead of vmcnt(8): +1.5 us per launch on the device timeline)
    const T* gsrc = gfr ? gfr : hfr;
    if constexpr (sizeof(T) == 8) {
        const unsigned e0 = (unsigned)A.e0 * (unsigned)sizeof(T), e1 = (unsigned)A.e1 * (unsigned)sizeof(T);
        const char* hu = reinterpret_cast<const char*>(hfr);
        const char* hv = reinterpret_cast<const char*>(hfr + g.ss);
        const char* ju = reinterpret_cast<const char*>(gsrc);
        const char* jv = reinterpret_cast<const char*>(gsrc + g.ss);
        a = *reinterpret_cast<const Pack<T, 2>*>(hu + e0); b = *reinterpret_cast<const Pack<T, 2>*>(hu + e1);
        c = *reinterpret_cast<const Pack<T, 2>*>(hv + e0); d = *reinterpret_cast<const Pack<T, 2>*>(hv + e1);
        a2 = *reinterpret_cast<const Pack<T, 2>*>(ju + e0); b2 = *reinterpret_cast<const Pack<T, 2>*>(ju + e1);
        c2 = *reinterpret_cast<const Pack<T, 2>*>(jv + e0); d2 = *reinterpret_cast<const Pack<T, 2>*>(jv + e1);
    } else {
        const long e0 = A.e0, e1 = A.e1;
        a = ld<T, 2>(hfr + e0); b = ld<T, 2>(hfr + e1);
        c = ld<T, 2>(hfr + g.ss + e0); d = ld<T, 2>(hfr + g.ss + e1);
        a2 = ld<T, 2>(gsrc + e0); b2 = ld<T, 2>(gsrc + e1);
        c2 = ld<T, 2>(gsrc + g.ss + e0); d2 = ld<T, 2>(gsrc + g.ss + e1);
    }
    o.u[0] = a.v[0]; o.u[1] = a.v[1]; o.u[2] = b.v[0]; o.u[3] = b.v[1];
    o.v[0] = c.v[0]; o.v[1] = c.v[1]; o.v[2] = d.v[0]; o.v[3] = d.v[1];
    o.ju[0] = a2.v[0]; o.ju[1] = a2.v[1]; o.ju[2] = b2.v[0]; o.ju[3] = b2.v[1];
    o.jv[0] = c2.v[0]; o.jv[1] = c2.v[1]; o.jv[2] = d2.v[0]; o.jv[3] = d2.v[1];
}

// PRE = true: the strip operands of this sub-step were requested at kernel start (`pre`), so the cold
// HBM latency of the trajectory / loss-gradient frames overlaps the window load and earlier sub-steps.
// MOM (float32 poly mode): the 20 coefficient moments  sum_x dt*adj_t[s]*phi_m(h_{t-1})  of the OWNED points are carried in
// registers (2-vectors, summed at the end of the launch) -- no adjoint trajectory, no separate moments pass.
// float32: 2-vector accumulators (v_pk_fma_f32); float64: scalar ones -- 40 instead of 80 VGPRs, what lets the 512-thread
// lambda-omega sweep carry them inside its 256-register budget (together with PRE = false, see the kernel)
template <typename T> struct MomAcc { using type = V2<T>; };
template <> struct MomAcc<double> { using type = double; };
template <typename T, bool MOM> struct TileMoments { typename MomAcc<T>::type a[MOM ? 2 : 1][MOM ? 10 : 1]; };
__device__ __forceinline__ void mom_add(V2<float>& a, V2<float> g, V2<float> phi) { a = vfma(g, phi, a); }
__device__ __forceinline__ void mom_add(double& a, V2<double> g, V2<double> phi) { a = fma_(g.x, phi.x, fma_(g.y, phi.y, a)); }
__device__ __forceinline__ void mom_add1(V2<float>& a, V2<float> g) { a += g; }
__device__ __forceinline__ void mom_add1(double& a, V2<double> g) { a += g.x + g.y; }
__device__ __forceinline__ float mom_total(V2<float> a) { return a.x + a.y; }
__device__ __forceinline__ double mom_total(double a) { return a; }

// the geometry word of this lane for one part of a pass (adj_substep<GEO>): bits 0-15 LDS offset of its strip inside a species
// plane, 16 live, 17-20 ownership of its four points.  Same strip map, clamps and compares as adj_substep's own derivation.
template <int K, int BX, int BY, int NT, int M, int PART, int TID0>
__device__ __forceinline__ unsigned persist_geo_word(const TileGeom& g, int ty0, int tx0)
{
    using TL = Tile<K, BX, BY>;
    using SM = StripMap<K, BX, BY, M, PART>;
    constexpr int RN4 = SM::N, O = 2 * (M + 1);
    int idx = (int)threadIdx.x - TID0;
    const bool live = idx >= 0 && idx < RN4;
    if (idx >= RN4) idx = RN4 - 1;
    if (idx < 0) idx = 0;
    int ry, rc;
    SM::locate(idx, ry, rc);
    const int ly = ry + O, lx = 4 * rc + O;
    const unsigned own_ny = (unsigned)min(BY, g.H - ty0), own_nx = (unsigned)min(BX, g.W - tx0);
    const bool rowin = live && (unsigned)(ly - 2 * K) < own_ny;
    unsigned w = (unsigned)(ly * TL::LX + lx) | (live ? 1u << 16 : 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) w |= (rowin && (unsigned)(lx + i - 2 * K) < own_nx) ? (1u << (17 + i)) : 0u;
    return w;
}
// ... of a HALF-strip: lane index hh = tid - TID0 -> strip hh / 2 of the part, points 2 (hh % 2) .. + 1 (bits 17, 18: their ownership)
template <int K, int BX, int BY, int NT, int M, int PART, int TID0>
__device__ __forceinline__ unsigned persist_half_geo_word(const TileGeom& g, int ty0, int tx0)
{
    using TL = Tile<K, BX, BY>;
    using SM = StripMap<K, BX, BY, M, PART>;
    constexpr int RN4 = SM::N, O = 2 * (M + 1);
    int hh = (int)threadIdx.x - TID0;
    const bool live = hh >= 0 && hh < 2 * RN4;
    // (idle lanes of a partly live wave shadow the part's first / last half-strip -- the one persist_half_off gives them the operands
    // of: they compute and store that half-strip's own values once more, like the idle lanes of the whole-strip passes)
    if (hh < 0) hh = 0;
    if (hh >= 2 * RN4) hh = 2 * RN4 - 1;
    int ry, rc;
    SM::locate(hh >> 1, ry, rc);
    const int ly = ry + O, lx = 4 * rc + O + 2 * (hh & 1);
    const unsigned own_ny = (unsigned)min(BY, g.H - ty0), own_nx = (unsigned)min(BX, g.W - tx0);
    const bool rowin = live && (unsigned)(ly - 2 * K) < own_ny;
    unsigned w = (unsigned)(ly * TL::LX + lx) | (live ? 1u << 16 : 0u);
#pragma unroll
    for (int i = 0; i < 2; ++i) w |= (rowin && (unsigned)(lx + i - 2 * K) < own_nx) ? (1u << (17 + i)) : 0u;
    return w;
}
// GEO: the lane's strip geometry of this pass -- LDS offset, liveness, ownership of its four points -- comes packed in one word
// of an LDS table built once per launch (persistent split sweep: persist_geo_word) instead of being derived from the lane id in
// every pass of every group: the derivation (strip map with its divisions, clamps, four ownership compares and selects) was
// ~60 of the ~300 VALU instructions of an issue-bound pass, and hoisting it into registers for all passes at once spills.
// LACC: lanes per row of the float64 moment accumulators in LDS (`lacc`[20][LACC]; lanes tid and tid + LACC share a slot -- the
// adds are LDS atomics)
// HALFS (round 6, geometry words only): the lane works on a HALF-strip -- the two points at the word's offset; `pre` holds their
// operands in elements 0, 1 and bits 17, 18 of the word their ownership.  Same operations per point.
template <typename T, int HC, int K, int BX, int BY, int NT, int M, bool PRE, bool MOM, int PART = PART_FULL, int TID0 = 0,
          bool GEO = false, int LACC = NT, bool HALFS = false>
__device__ __forceinline__ void adj_substep(T* cur, T* nxt, const T* __restrict__ hfr, const T* __restrict__ gfr,
                                            const TileGeom& g, int ty0, int tx0, const T* __restrict__ P,
                                            double (&acc_c)[2], const StripOps<T>& pre, TileMoments<T, MOM>& mom,
                                            double* lacc = nullptr, const unsigned* geo = nullptr,
                                            const JacPairs<T>* jp = nullptr)
{
    static_assert(P_DT == 0 && P_COEF == 1 && P_C0 == 3 && P_TAPS == 4, "JacPairs::st follows P's order");
    auto cf = [P, jp](int i) -> V2<T> {
        if constexpr (GEO && held_masks<T>::sten != 0) { if ((held_masks<T>::sten >> i) & 1) return jp->st[i]; }
        return vs(P[i]);
    };
    using TL = Tile<K, BX, BY>;
    using SM = StripMap<K, BX, BY, M, PART>;
    constexpr int RN4 = SM::N, O = 2 * (M + 1);
    constexpr int PT = (RN4 + NT - 1) / NT;
    static_assert(!PRE || PT == 1, "prefetched operands cover one strip per lane");
    static_assert(!HALFS || (GEO && PRE && HC == POLY), "half-strips: resident sweep passes with geometry words and prefetched operands");
    constexpr int NH = HALFS ? 1 : 2;                       // point pairs per lane
    int tid = (int)threadIdx.x;
#if PI_PERSIST_OPAQUE_TID
    // split persistent sweep: eight passes inlined into one loop.  Everything derived from the lane's strip position (LDS
    // offsets, ownership masks, ...) is loop-invariant there and would be hoisted out of the group loop for ALL eight passes
    // at once -- ~80 VGPRs held for the whole kernel, which then spills.  An opaque lane id keeps each pass's geometry local.
    if constexpr (PART != PART_FULL) asm volatile("" : "+v"(tid));
#endif
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        int off;
        bool live;
        unsigned ownbits = 0;                              // bit i: point i of the strip is an owned, in-grid point
        if constexpr (GEO) {
            static_assert(PT == 1, "geometry words: one strip per lane");
            const unsigned w = geo[(int)threadIdx.x];
            off = (int)(w & 0xFFFFu);
            live = (w >> 16) & 1u;
            ownbits = (w >> 17) & 0xFu;
        } else {
            int idx = tid - TID0 + q * NT;                 // (TID0 != 0: the caller has sent the waves below TID0 elsewhere)
            live = idx < RN4;
            if (!live) idx = RN4 - 1;
            int ry, rc;
            SM::locate(idx, ry, rc);
            const int ly = ry + O, lx = 4 * rc + O;
            off = ly * TL::LX + lx;
            // owned, in-grid points: one unsigned compare per coordinate against the tile's owned extent (edge tiles of a
            // ragged grid own less) -- the sub-step is issue-bound, the three-compare form cost 16 VALU per strip
            const unsigned own_ny = (unsigned)min(BY, g.H - ty0), own_nx = (unsigned)min(BX, g.W - tx0);
            const bool rowin = live && (unsigned)(ly - 2 * K) < own_ny;
#pragma unroll
            for (int i = 0; i < 4; ++i) ownbits |= (rowin && (unsigned)(lx + i - 2 * K) < own_nx) ? (1u << i) : 0u;
        }
        StripOps<T> lo;
        if constexpr (!PRE) adj_load_ops<T, K, BX, BY, NT, M>(lo, q, hfr, gfr, g, ty0, tx0);
        // whole waves beyond the region skip the strip (see fwd_substep); their operand loads above stay unconditional --
        // loads inside a branch would cost the compiler its count of outstanding requests
        if constexpr (GEO) {
            // (from the geometry word too: the two parts of a mixed pass share this one body, see persist_pass)
            if (__builtin_amdgcn_ballot_w64(live) == 0ull) continue;
        } else {
            if (((int)threadIdx.x & ~(WAVE - 1)) - TID0 + q * NT >= RN4) continue;
        }
        const StripOps<T>& op = PRE ? pre : lo;
        const T (&u)[4] = op.u;
        const T (&v)[4] = op.v;
        const T (&ju)[4] = op.ju;
        const T (&jv)[4] = op.jv;
        const V2<T> U[2] = {V2<T>{u[0], u[1]}, V2<T>{u[2], u[3]}}, V[2] = {V2<T>{v[0], v[1]}, V2<T>{v[2], v[3]}};
        V2<T> gc[2][2], dl[2][2];                          // [species][half of the strip]
        if constexpr (HALFS) {
            lds_star2v_cf<T, TL::LX, -1>(cur + off, cf, gc[0][0], dl[0][0]);
            lds_star2v_cf<T, TL::LX, -1>(cur + TL::PLANE + off, cf, gc[1][0], dl[1][0]);
        } else {
            lds_star4v_cf<T, TL::LX, -1>(cur + off, cf, gc[0], dl[0]);
            lds_star4v_cf<T, TL::LX, -1>(cur + TL::PLANE + off, cf, gc[1], dl[1]);
        }
        const V2<T> dtv = cf(P_DT);
        V2<T> own[2];                                      // 1 for owned, in-grid points, else 0 (MOM only)
        V2<T> cs[2] = {vs(T(0)), vs(T(0))};                 // this strip's owned part of sum_x dt*LapT(a)*h, per species
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            dl[0][h] *= dtv;
            dl[1][h] *= dtv;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                own[h][e] = ((ownbits >> (2 * h + e)) & 1u) ? T(1) : T(0);        // owned, in-grid points only
            }
            // the four products of a strip are added in the compute type, the strip sum goes to the fp64 accumulator (was:
            // every product converted and added in fp64 -- 16 half-rate instructions per strip in an issue-bound loop; the
            // products themselves are already rounded to the compute type, so the sum loses nothing that matters)
            cs[0] = vfma(dl[0][h] * U[h], own[h], cs[0]);
            cs[1] = vfma(dl[1][h] * V[h], own[h], cs[1]);
        }
        acc_c[0] += (double)(cs[0].x + cs[0].y);
        acc_c[1] += (double)(cs[1].x + cs[1].y);
        V2<T> du[2] = {vs(T(0)), vs(T(0))}, dv[2] = {vs(T(0)), vs(T(0))};
        if constexpr (HC == POLY) {
            // unrolled over the species: the 20 coefficients become loop-invariant scalar loads
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const T* c = P + P_W + 10 * s;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const V2<T> gr = gc[s][h] * dtv;
                    V2<T> ru, rv;
                    if constexpr (GEO && held_masks<T>::jac != 0) poly_dr_v_j(c, jp->m[s], U[h], V[h], ru, rv);
                    else poly_dr_v(c, U[h], V[h], ru, rv);
                    du[h] = vfma(gr, ru, du[h]);
                    dv[h] = vfma(gr, rv, dv[h]);
                    if constexpr (MOM && sizeof(T) == 4) {
                        auto& a = mom.a[s];
                        const V2<T> gm = gr * own[h];
                        const V2<T> uu = U[h], vv = V[h];
                        const V2<T> u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
                        mom_add1(a[0], gm);
                        mom_add(a[1], gm, uu); mom_add(a[2], gm, vv);
                        mom_add(a[3], gm, u2); mom_add(a[4], gm, uv); mom_add(a[5], gm, v2);
                        mom_add(a[6], gm, u2 * uu); mom_add(a[7], gm, u2 * vv);
                        mom_add(a[8], gm, uu * v2); mom_add(a[9], gm, v2 * vv);
                    }
                }
            }
        } else {
#pragma clang loop unroll(disable)
            for (int s = 0; s < 2; ++s) {
                const V2<T> gr[2] = {(s == 0 ? gc[0][0] : gc[1][0]) * dtv, (s == 0 ? gc[0][1] : gc[1][1]) * dtv};
                const T* W = P + P_W + s * species_block(HC);
                W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
                for (int j = 0; j < HC; ++j) {
                    const W10<T> c = nx;
                    if (j + 1 < HC) nx = load_w10(W + 10 * (j + 1));
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const V2<T> a1 = vfma(vs(c.w[0]), U[h], vfma(vs(c.w[1]), V[h], vs(c.w[2])));
                        const V2<T> a2 = vfma(vs(c.w[3]), U[h], vfma(vs(c.w[4]), V[h], vs(c.w[5])));
                        const V2<T> a3 = vfma(vs(c.w[6]), U[h], vfma(vs(c.w[7]), V[h], vs(c.w[8])));
                        const V2<T> p12 = a1 * a2;
                        const V2<T> gw = gr[h] * vs(c.w[9]);
                        const V2<T> q1 = gw * (a2 * a3), q2 = gw * (a1 * a3), q3 = gw * p12;
                        du[h] = vfma(q1, vs(c.w[0]), vfma(q2, vs(c.w[3]), vfma(q3, vs(c.w[6]), du[h])));
                        dv[h] = vfma(q1, vs(c.w[1]), vfma(q2, vs(c.w[4]), vfma(q3, vs(c.w[7]), dv[h])));
                    }
                }
            }
        }
        const V2<T> cu = cf(P_COEF + 0), cv = cf(P_COEF + 1);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const V2<T> tu = cu * dl[0][h] + du[h];
            const V2<T> tv = cv * dl[1][h] + dv[h];
            V2<T> ou = gc[0][h] + tu, ov = gc[1][h] + tv;
            if (gfr) {
                V2<T> iu = V2<T>{ju[2 * h], ju[2 * h + 1]}, iv = V2<T>{jv[2 * h], jv[2 * h + 1]};
                if (g.loss.mode) {                         // wave-uniform: the loss gradient is formed here (pi_device.h)
                    const V2<T> la = vs(loss_factor<T>(g.loss));
                    iu = la * (g.loss.mode == 2 ? U[h] - iu : U[h]);
                    iv = la * (g.loss.mode == 2 ? V[h] - iv : V[h]);
                }
                ou += iu;
                ov += iv;
            }
            stv2(nxt + off + 2 * h, ou);
            stv2(nxt + TL::PLANE + off + 2 * h, ov);
        }
        if constexpr (MOM && sizeof(T) == 8) {
            // float64: 20 more accumulators (40 registers) do not fit next to the operand pipeline in the 256 registers of a
            // 512-thread workgroup (the register flavour spilled 57 doubles: 5.98 -> 9.13 us per step).  Each lane keeps its
            // 20 sums in LDS instead -- slot [moment][thread], touched by that lane only -- and adds to them with
            // `ds_add_f64` (no return value, nothing to wait for): the LDS pipe does the accumulation, the VALU only the
            // products, and the layout is already the transposed one the tail reduction reads.  Moments LAST, half a strip
            // at a time: the stencil / Jacobian temporaries are dead by now.
            double* slot = lacc + (LACC == NT ? (int)threadIdx.x : (int)threadIdx.x % LACC);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const V2<T> uu = U[h], vv = V[h];
                const V2<T> u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const V2<T> gm = (gc[s][h] * dtv) * own[h];
                    double* a = slot + 10 * s * LACC;
                    auto add = [&](int m, V2<T> phi) { lds_add_f64(a + m * LACC, fma_(gm.x, phi.x, gm.y * phi.y)); };
                    lds_add_f64(a, gm.x + gm.y);
                    add(1, uu); add(2, vv);
                    add(3, u2); add(4, uv); add(5, v2);
                    add(6, u2 * uu); add(7, u2 * vv);
                    add(8, uu * v2); add(9, v2 * vv);
                }
            }
        }
    }
}

// HALF-strips (round 6): byte offset of the lane's two points inside a species plane, and their pointwise operands
template <typename T, int K, int BX, int BY, int NT, int M, int PART, int TID0>
__device__ __forceinline__ unsigned persist_half_off(const TileGeom& g, int ty0, int tx0)
{
    using SM = StripMap<K, BX, BY, M, PART>;
    constexpr int RN4 = SM::N, O = 2 * (M + 1);
    int hh = (int)threadIdx.x - TID0;
    if (hh < 0) hh = 0;
    if (hh >= 2 * RN4) hh = 2 * RN4 - 1;
    int ry, rc;
    SM::locate(hh >> 1, ry, rc);
    const int ly = ry + O, lx = 4 * rc + O + 2 * (hh & 1);
    const int gy = wrap1(ty0 - 2 * K + ly, g.H), gx = wrap1(tx0 - 2 * K + lx, g.W);      // (two points never straddle the wrap: W is even)
    return (unsigned)((long)gy * g.W + gx) * (unsigned)sizeof(T);
}
template <typename T>
__device__ __forceinline__ void persist_load_ops_half(StripOps<T>& o, const T* __restrict__ hfr, const T* __restrict__ gfr,
                                                      const TileGeom& g, unsigned off)
{
    const T* gsrc = gfr ? gfr : hfr;
    const Pack<T, 2> a = *reinterpret_cast<const Pack<T, 2>*>(reinterpret_cast<const char*>(hfr) + off);
    const Pack<T, 2> c = *reinterpret_cast<const Pack<T, 2>*>(reinterpret_cast<const char*>(hfr + g.ss) + off);
    const Pack<T, 2> a2 = *reinterpret_cast<const Pack<T, 2>*>(reinterpret_cast<const char*>(gsrc) + off);
    const Pack<T, 2> c2 = *reinterpret_cast<const Pack<T, 2>*>(reinterpret_cast<const char*>(gsrc + g.ss) + off);
    o.u[0] = a.v[0]; o.u[1] = a.v[1]; o.v[0] = c.v[0]; o.v[1] = c.v[1];
    o.ju[0] = a2.v[0]; o.ju[1] = a2.v[1]; o.jv[0] = c2.v[0]; o.jv[1] = c2.v[1];
}

// PRE: the pointwise operands of sub-step M+1 are requested before sub-step M is computed and stay in flight across
// its LDS barrier (one strip per lane only); `ops` holds the operands of sub-step M, requested one sub-step earlier.
// HALFS (round 6, resident small-tile sweep): the lanes work on half-strips -- `geo` rows from persist_half_geo_word, operand byte
// offsets per sub-step in `hoff`[K] (persist_half_off) instead of `sa`.
template <typename T, int HC, int K, int BX, int BY, int NT, int M, bool PRE, bool MOM, bool GEO = false, bool HALFS = false>
__device__ __forceinline__ void adj_substeps(T* b0, T* b1, const T* __restrict__ hbase, const T* __restrict__ gbase,
                                             T* __restrict__ abase, long frame_stride, unsigned inj_mask,
                                             T* __restrict__ g_h0, int steps_to_zero, const TileGeom& g, int ty0,
                                             int tx0, const T* __restrict__ P, double (&acc_c)[2],
                                             const StripOps<T>& ops, TileMoments<T, MOM>& mom, const StripAddr (&sa)[K],
                                             double* lacc = nullptr, bool store_handover = true, const unsigned* geo = nullptr,
                                             const JacPairs<T>* jp = nullptr, const unsigned* hoff = nullptr)
{
    T* cur = (M & 1) ? b1 : b0;
    T* nxt = (M & 1) ? b0 : b1;
    const long fo = -(long)(M + 1) * frame_stride;         // frame t-1-M relative to frame t
    StripOps<T> ahead;
    if constexpr (PRE && M + 1 < K) {
        const long fn = -(long)(M + 2) * frame_stride;
        if constexpr (HALFS)
            persist_load_ops_half<T>(ahead, hbase + fn, (inj_mask >> (M + 1)) & 1u ? gbase + fn : nullptr, g, hoff[M + 1]);
        else
            adj_load_ops<T, K, BX, BY, NT, M + 1>(ahead, 0, hbase + fn, (inj_mask >> (M + 1)) & 1u ? gbase + fn : nullptr, g,
                                                  ty0, tx0, &sa[M + 1]);
    }
    // (GEO: row M of the caller's [K][NT] table of geometry words + its held coefficient pairs -- resident sweeps only)
    adj_substep<T, HC, K, BX, BY, NT, M, PRE, MOM, PART_FULL, 0, GEO, NT, HALFS>(cur, nxt, hbase + fo, (inj_mask >> M) & 1u ? gbase + fo : nullptr, g,
                                                                                 ty0, tx0, P, acc_c, ops, mom, lacc, GEO ? geo + M * NT : nullptr, jp);
#if PI_PIN_MOMENTS
    // Pin this sub-step's moment accumulation HERE.  Left alone, the scheduler sinks the moment FMAs of all four sub-steps
    // (they depend on no LDS traffic) behind the last barrier -- 350 VALU instructions in the tail of the launch, where all
    // 256 workgroups execute them at the same time with nothing to overlap; inside the sub-step they fill LDS-latency and
    // barrier bubbles of the two waves that share a SIMD.
    if constexpr (MOM && sizeof(T) == 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) asm volatile("" : "+v"(mom.a[s][m]));
    }
#endif
    PI_STAMP(2 + 3 * M);
    lds_barrier();
    PI_STAMP(3 + 3 * M);
    // the adjoint of frame 0 is the caller's dL/dh0 output.  MOM: nobody reads the intermediate adjoint frames (the
    // moments are reduced right here), only the hand-over frame t-K goes to memory
    if constexpr (!MOM || M + 1 == K) {
        // (persistent sweeps keep the state in LDS between groups: store_handover = false skips the store of frame t - K; the
        // split schedule's intermediate frames -- !MOM, M + 1 < K -- are always written)
        if (store_handover || (!MOM && M + 1 < K)) {
            T* dst = (M + 1 == steps_to_zero && g_h0) ? g_h0 : abase + fo;
            tile_store<T, K, BX, BY, NT, (M & 1) != 0>(nxt, dst, g, ty0, tx0);
        }
    }
    PI_STAMP(4 + 3 * M);
    if constexpr (M + 1 < K)
        adj_substeps<T, HC, K, BX, BY, NT, M + 1, PRE, MOM, GEO, HALFS>(b0, b1, hbase, gbase, abase, frame_stride, inj_mask, g_h0,
                                                                        steps_to_zero, g, ty0, tx0, P, acc_c, ahead, mom, sa, lacc,
                                                                        store_handover, geo, jp, hoff);
}

template <typename T, int HC, int K, int BX, int BY, int NT, bool MOM = false>
__global__ void __launch_bounds__(NT)
pi_adj2d_tile_kernel(const T* __restrict__ hframe_t, const T* __restrict__ gframe_t, T* __restrict__ aframe_t,
                     long frame_stride, unsigned inj_mask, T* __restrict__ g_h0, int steps_to_zero,
                     double* __restrict__ partials, int np, const T* __restrict__ P, TileGeom g)
{
    using TL = Tile<K, BX, BY>;
    // Operand pipeline: one sub-step ahead (2 x 16 VGPRs).  Requesting ALL sub-steps' operands at kernel start was
    // measured slower (17.6 vs 15.5 us per K=4 launch: 64 extra VGPRs, requests queued ahead of the window load).
    // (the float64 fused-moments flavour gives the operand pipeline's 32 registers to its 20 accumulators: with both it
    // needs ~270 of the 256 registers a 512-thread workgroup can have and spills -- 20 -> 47 us per launch, measured)
    constexpr bool PRE = PI_TILE_ADJ_PIPE && TL::region_n(0) / 4 <= NT;
    constexpr bool MOM_LACC = MOM && sizeof(T) == 8;       // float64: per-lane moment accumulators in LDS (adj_substep)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int ty0 = (tile / g.tiles_x) * BY, tx0 = (tile % g.tiles_x) * BX;
    // [20][NT] doubles behind the state buffers (not aliased: they live through all sub-steps)
    double* lacc = MOM_LACC ? reinterpret_cast<double*>(smem_raw + tile_state_bytes<T, K, BX, BY>()) : nullptr;
    PI_STAMP_PREV();
    PI_STAMP(0);
    WindowLoader<T, K, BX, BY, NT> wl;
    wl.issue(aframe_t, g, ty0, tx0);                       // adjoint window first, then the operands of sub-step 0
    if constexpr (MOM_LACC) {
#pragma unroll
        for (int m = 0; m < 20; ++m) lacc[m * NT + (int)threadIdx.x] = 0.0;
    }
    // running diffusion-coefficient partial of this tile: requested now, needed at the very end (was a dependent
    // load -> add -> store at the end of every launch: 1 us)
    static_assert(!MOM || HC == POLY, "fused moments: pre-contracted blocks");
    // slots of the partial row this thread updates at the end: threads 0,1 the two coefficient sums, MOM: threads
    // 2..21 the 20 moments (row layout of the direct kernels: P_W + 10*s + m)
    const int slot = threadIdx.x < 2 ? P_COEF + (int)threadIdx.x : P_W + (int)threadIdx.x - 2;
    const bool has_slot = threadIdx.x < (MOM ? 22 : 2);
    double* pslot = partials + (long)blockIdx.x * np + (has_slot ? slot : P_COEF);
    const double pold = has_slot ? *pslot : 0.0;
    StripOps<T> ops0;
    // The strip offsets of ALL sub-steps are computed here, in the shadow of the window load (the VALU is idle for ~1 us),
    // and pinned: inside the issue-bound sub-steps the wraps, 64-bit multiplies and shifts of the next strip's operand
    // addresses were ~45 of ~290 VALU instructions per wave and sub-step.
    StripAddr sa[K];
    if constexpr (PRE) {
        strip_addr_table<K, BX, BY, NT, 0>(sa, g, ty0, tx0);
        adj_load_ops<T, K, BX, BY, NT, 0>(ops0, 0, hframe_t - frame_stride, inj_mask & 1u ? gframe_t - frame_stride : nullptr,
                                          g, ty0, tx0, &sa[0]);
    }
    wl.commit(b0);
    lds_barrier();                                         // LDS only: the operand loads stay in flight
    PI_STAMP(1);
    double acc_c[2] = {0.0, 0.0};                          // heavily cancelling sums (stencil row-sum ~ 0): fp64
    TileMoments<T, MOM> mom;
    if constexpr (MOM) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) mom.a[s][m] = typename MomAcc<T>::type{};
    }
    adj_substeps<T, HC, K, BX, BY, NT, 0, PRE, MOM>(b0, b1, hframe_t, gframe_t, aframe_t, frame_stride, inj_mask, g_h0,
                                                    steps_to_zero, g, ty0, tx0, P, acc_c, ops0, mom, sa, lacc);
    // diffusion-coefficient gradients of this tile over the K sub-steps: one reduction per launch
    // (LDS-only barriers: the last frame's global stores need not drain first)
    lds_barrier();
    double* red = reinterpret_cast<double*>(smem_raw);     // state buffers are dead now
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    constexpr int NW = NT / WAVE;
    // red[0 .. 2*NW): per-wave coefficient sums; MOM (float32): red[2*NW .. 2*NW + 20): block totals of the moments;
    // float scratch [20][NT + 16] behind them
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double r = wave_sum_to_last(acc_c[s]);
        if (lane == REDUCE_LANE) red[wave * 2 + s] = r;
    }
    constexpr bool MOM_LDS = MOM && sizeof(T) == 4;        // moments through an LDS transpose (see below)
    if constexpr (MOM_LDS) {
        // Block-wide sums of the 20 per-thread moments through an LDS transpose: every thread writes its 20 values, then
        // 16 lanes per moment add NT/16 values each and fold with four DPP steps.  All 8 waves of all 256 workgroups reach
        // this tail at the same time, so its instruction count is exposed in full: the earlier 20 six-step DPP wave
        // reductions per wave cost ~1 us of a 12.6 us launch (same finding as in pi_bwd_kernel, where removing the
        // reduction in a timing experiment gained 1.6 us at 128^3).
        constexpr int RS = NT + 16;                        // + 16 floats per row: 4 rows cover all 64 banks
        T* scr = reinterpret_cast<T*>(red + 2 * NW + 20);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) scr[(10 * s + m) * RS + (int)threadIdx.x] = mom_total(mom.a[s][m]);
        lds_barrier();
        if (threadIdx.x < 320) {                           // five whole waves: 20 moments x 16 lanes
            const int mm = (int)threadIdx.x >> 4, part = (int)threadIdx.x & 15;
            const T* row = scr + mm * RS + part;
            T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
#pragma unroll
            for (int k = 0; k < NT; k += 128) {
                const T v0 = row[k], v1 = row[k + 16], v2 = row[k + 32], v3 = row[k + 48];
                const T v4 = row[k + 64], v5 = row[k + 80], v6 = row[k + 96], v7 = row[k + 112];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
                a0 += v4; a1 += v5; a2 += v6; a3 += v7;
            }
            T a = (a0 + a1) + (a2 + a3);
            a += dpp_mov<0x111, 0xF>(a);                   // row_shr:1, :2, :4, :8 -> lane 15 of each row of 16
            a += dpp_mov<0x112, 0xF>(a);
            a += dpp_mov<0x114, 0xF>(a);
            a += dpp_mov<0x118, 0xF>(a);
            if (part == 15) red[2 * NW + mm] = (double)a;
        }
    } else if constexpr (MOM_LACC) {
        // the per-lane sums already sit transposed in LDS ([moment][thread], complete: the barrier above waited for the
        // LDS adds): 16 lanes per moment add NT/16 of them each and fold with four DPP steps
        if (threadIdx.x < 320) {
            const int mm = (int)threadIdx.x >> 4, part = (int)threadIdx.x & 15;
            const double* row = lacc + mm * NT + part;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int k = 0; k < NT; k += 64) {
                const double v0 = row[k], v1 = row[k + 16], v2 = row[k + 32], v3 = row[k + 48];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            double a = (a0 + a1) + (a2 + a3);
            a += dpp_mov<0x111, 0xF>(a);
            a += dpp_mov<0x112, 0xF>(a);
            a += dpp_mov<0x114, 0xF>(a);
            a += dpp_mov<0x118, 0xF>(a);
            if (part == 15) red[2 * NW + mm] = a;
        }
    }
    lds_barrier();
    if (has_slot) {
        double sum = 0.0;
        if (threadIdx.x < 2) {
            for (int w = 0; w < NW; ++w) sum += red[w * 2 + threadIdx.x];
        } else if constexpr (MOM) {
            sum = red[2 * NW + threadIdx.x - 2];
        }
        *pslot = pold + sum;
    }
    PI_STAMP(15);
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT fused sweep (round 3): the whole reverse sweep of a rollout in ONE cooperative launch, one resident workgroup
// per tile.  What a K = 4 sweep launch spends outside its four sub-steps -- kernel boundary 2.3 us, cold window load 2.4 us,
// moment reduction + partial-row update 1.2 us, of 11.6 us -- is replaced by a halo hand-over between resident workgroups
// with DATA-TAGGED GRANULES (8-byte {group number, value} words, agent-scope relaxed stores / loads: the data is the flag;
// tools/handover_granule_microbench.hip measured 3.4 us per hand-over for exactly this geometry).  Per group of K steps a
// workgroup: gathers its 2K-wide halo ring from the eight neighbours' bands (polling until every tag matches) around the tile
// it kept in LDS, runs the K sub-steps of pi_adj2d_tile_kernel (same device functions, same arithmetic: dL/dh0 bit-identical),
// publishes the 2K-wide border band of its tile.  The 20 coefficient moments are folded into double registers once per group
// and reduced ONCE per rollout.  Whole 32 x 32 tiles, float32 pre-contracted blocks; frame masks up to 4096 frames (kernel argument).
// Double-buffered by group parity; why that suffices: my publish of group e + 2 comes after my gather of e + 1, which saw the
// neighbour's band e + 1, which it published after ITS gather of e -- the last read of the slot I am about to overwrite.
// ------------------------------------------------------------------------------------------------
struct PersistArgs {
    unsigned long long* outbox;    // [2][tiles][BAND] granules, zeroed before the launch
    unsigned* sync;                // device words, zeroed before the launch: [0] workgroups that have started, [1] abort flag,
                                   // [2] number of time-outs
    int* host;                     // host-mapped status slot of THIS launch (may be null): [0] 1 once every workgroup has started
                                   // (roll call), [3] 1 once the launch has ABORTED (no output of it is valid), [1] / [2] group /
                                   // tile of the first time-out.  Separate words: a late workgroup that completes the roll call of
                                   // an aborted launch must not hide the abort.  Plain system-scope stores (no PCIe atomics).
    int ngroups;                   // groups of K steps run here: frames t_top .. t_top - K * ngroups
    unsigned long long timeout_ticks;        // bound of a hand-over wait (100 MHz ticks)
    unsigned long long first_timeout_ticks;  // ... of the FIRST hand-over: that is where a workgroup that never became resident
                                   // (CUs held by another process / kernel, CU mask) shows -- kept short so the host can fall back
    int t_top;                     // frame number of the top frame (group g covers frames t_top - K g - 1 ... t_top - K g - K)
    int pause;                     // small-tile resident forward: units of 64 clocks between the publish and the FIRST request of the
                                   // ring (granules asked for before the neighbours' stores are visible come back stale and cost a
                                   // second round trip)
    int masked;                    // 1: only the frames whose bit is set in `frames` carry a gradient (RCNN.observe's strided loss)
    unsigned frames[128];          // bit t of word t / 32: frame t carries a gradient (t < 4096)
};

// injection mask of the K steps below frame t: bit q = frame t - 1 - q carries a gradient
template <int K>
__device__ __forceinline__ unsigned persist_mask(const PersistArgs& pa, int t)
{
    if (!pa.masked) return (1u << K) - 1u;
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int f = t - 1 - q;
        m |= ((pa.frames[f >> 5] >> (f & 31)) & 1u) << q;
    }
    return m;
}

template <int B, int HW>
__device__ __forceinline__ int band_index(int y, int x)      // position inside the tile -> index inside its border band
{
    if (y < HW) return y * B + x;
    if (y >= B - HW) return HW * B + (y - (B - HW)) * B + x;
    const int r = y - HW;                                     // middle rows: 2 * HW values per row
    return 2 * HW * B + r * 2 * HW + (x < HW ? x : HW + (x - (B - HW)));
}

template <typename T, int K, int BX, int BY, int NT>
__global__ void __launch_bounds__(NT)
pi_adj2d_persist_kernel(const T* __restrict__ hframe_t, const T* __restrict__ gframe_t, T* __restrict__ aframe_t,
                        long frame_stride, T* __restrict__ g_h0, double* __restrict__ partials, int np,
                        const T* __restrict__ P, TileGeom g, PersistArgs pa)
{
    static_assert(sizeof(T) == 4 && BX == BY, "float32, square tiles");
    using TL = Tile<K, BX, BY>;
    constexpr int HW = 2 * K, LXW = TL::LX;
    constexpr int BANDH = BX * BX - (BX - 2 * HW) * (BX - 2 * HW);      // border values per species
    constexpr int RINGH = LXW * LXW - BX * BX;                          // halo values per species
    constexpr int NPUB = (2 * BANDH + NT - 1) / NT, NGAT = (2 * RINGH + NT - 1) / NT;
    constexpr bool PRE = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int tyi = tile / g.tiles_x, txi = tile % g.tiles_x, tiles_y = g.H / BY;
    const int ty0 = tyi * BY, tx0 = txi * BX;
    const int ntiles = g.tiles_x * tiles_y;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* outbox = (gu64*)pa.outbox;

    // ---- what this lane publishes / gathers in every hand-over (fixed geometry): tables in LDS, not registers -- the sub-steps
    // hold 164 registers, and geometry + double accumulators on top of that spilled
    // LDS: state buffers | [20][NT] doubles: the lane's moments of all groups so far | int tables
    double* lacc = reinterpret_cast<double*>(smem_raw + tile_state_bytes<T, K, BX, BY>());
    int* tab_pub = reinterpret_cast<int*>(lacc + 20 * NT);                  // [NPUB][NT]: LDS position of a border value
    int* tab_gl = tab_pub + NPUB * NT;                                      // [NGAT][NT]: LDS position of a halo value
    int* tab_gs = tab_gl + NGAT * NT;                                       // [NGAT][NT]: granule index inside a parity half
    int* wg_abort = tab_gs + NGAT * NT;                                     // [1]: some wave of this workgroup gave up
    // residency roll call: the workgroup that completes it tells the host (which may be waiting for exactly that before it
    // returns from the entry point -- see launch_adj_persist); a workgroup that never starts shows as a time-out of the first
    // hand-over of its neighbours
    if (threadIdx.x == 0) {
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(pa.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)ntiles && pa.host) __hip_atomic_store(pa.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int m = 0; m < 20; ++m) lacc[m * NT + (int)threadIdx.x] = 0.0;
#pragma unroll
    for (int q = 0; q < NPUB; ++q) {
        const int i = (int)threadIdx.x + q * NT;
        int pl = -1;
        if (i < 2 * BANDH) {
            const int sp = i / BANDH, e = i - sp * BANDH;
            int y, x;                                      // inverse of band_index
            if (e < HW * BX) { y = e / BX; x = e - y * BX; }
            else if (e < 2 * HW * BX) { const int m = e - HW * BX; y = BX - HW + m / BX; x = m % BX; }
            else { const int m = e - 2 * HW * BX; y = HW + m / (2 * HW); const int c = m % (2 * HW); x = c < HW ? c : BX - 2 * HW + c; }
            pl = sp * TL::PLANE + (HW + y) * LXW + HW + x;
        }
        tab_pub[q * NT + (int)threadIdx.x] = pl;
    }
#pragma unroll
    for (int q = 0; q < NGAT; ++q) {
        const int r = (int)threadIdx.x + q * NT;
        int gl = -1, gs = 0;
        if (r < 2 * RINGH) {
            const int sp = r / RINGH, e = r - sp * RINGH;
            int wy, wx;                                    // ring positions row-major over the window, skipping the centre
            if (e < HW * LXW) { wy = e / LXW; wx = e - wy * LXW; }
            else if (e < HW * LXW + BX * 2 * HW) { const int m = e - HW * LXW; wy = HW + m / (2 * HW); const int c = m % (2 * HW); wx = c < HW ? c : BX + c; }
            else { const int m = e - HW * LXW - BX * 2 * HW; wy = HW + BX + m / LXW; wx = m % LXW; }
            const int gy = ty0 + wy - HW, gx = tx0 + wx - HW;                       // global point (may wrap)
            const int nty = ((gy + g.H) / BY) % tiles_y, ntx = ((gx + g.W) / BX) % g.tiles_x;
            const int ly = (gy + g.H) % BY, lx = (gx + g.W) % BX;
            gl = sp * TL::PLANE + wy * LXW + wx;
            gs = (nty * g.tiles_x + ntx) * (2 * BANDH) + sp * BANDH + band_index<BX, HW>(ly, lx);
        }
        tab_gl[q * NT + (int)threadIdx.x] = gl;
        tab_gs[q * NT + (int)threadIdx.x] = gs;
    }

    // ---- group 0 starts from the adjoint frame in memory (the top frame), like a launch of pi_adj2d_tile_kernel ----------
    WindowLoader<T, K, BX, BY, NT> wl;
    wl.issue(aframe_t, g, ty0, tx0);
    StripOps<T> ops0;
    StripAddr sa[K];
    strip_addr_table<K, BX, BY, NT, 0>(sa, g, ty0, tx0);
    unsigned gmask = persist_mask<K>(pa, pa.t_top);
    adj_load_ops<T, K, BX, BY, NT, 0>(ops0, 0, hframe_t - frame_stride, gmask & 1u ? gframe_t - frame_stride : nullptr, g, ty0, tx0,
                                      &sa[0]);
    wl.commit(b0);
    lds_barrier();
    double acc_c[2] = {0.0, 0.0};
    bool failed = false;
    TileMoments<T, true> mom;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 10; ++m) mom.a[s][m] = V2<T>{T(0), T(0)};
    for (int grp = 0; grp < pa.ngroups; ++grp) {
        const long go = -(long)grp * K * frame_stride;     // this group's frame t relative to the top frame
        const bool last = grp + 1 == pa.ngroups;
        // the last group hands its result to memory (frame t - K of the adjoint trajectory, or dL/dh0 itself)
        adj_substeps<T, POLY, K, BX, BY, NT, 0, PRE, true>(b0, b1, hframe_t + go, gframe_t + go, aframe_t + go, frame_stride,
                                                           gmask, g_h0, g_h0 && last ? K : 0, g, ty0, tx0, P, acc_c, ops0,
                                                           mom, sa, nullptr, last);
        // the float32 2-vector moment sums (what a launch of the tile sweep carries over its K steps) are folded into the lane's
        // double sums in LDS every fourth group -- plain read-add-write, the slot is the lane's own (ds_add_f64 processes about one
        // lane per clock: 20 of them per lane and group were 5 us of a 10.7 us group)
        if ((grp & 3) == 3 || last) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int m = 0; m < 10; ++m) {
                    double* slot = lacc + (10 * s + m) * NT + (int)threadIdx.x;
                    *slot += (double)mom_total(mom.a[s][m]);
                    mom.a[s][m] = V2<T>{T(0), T(0)};
                }
        }
        if (last) break;
        // operands of the next group's first sub-step: requested now, they travel during the hand-over
        const long gn = go - (long)K * frame_stride;
        gmask = persist_mask<K>(pa, pa.t_top - K * (grp + 1));
        adj_load_ops<T, K, BX, BY, NT, 0>(ops0, 0, hframe_t + gn - frame_stride, gmask & 1u ? gframe_t + gn - frame_stride : nullptr, g,
                                          ty0, tx0, &sa[0]);
        lds_barrier();                                     // sub-step K - 1 wrote buffer 0 (K even): everybody's strips are in
        // ---- hand-over: publish my band, gather my ring ----
        const unsigned epoch = (unsigned)grp + 1u;
        gu64* half = outbox + (size_t)(epoch & 1u) * (size_t)ntiles * (2 * BANDH);
        gu64* mine = half + (size_t)tile * (2 * BANDH);
#pragma unroll
        for (int q = 0; q < NPUB; ++q) {
            const int pl = tab_pub[q * NT + (int)threadIdx.x];
            if (pl >= 0) {
                const unsigned v = __builtin_bit_cast(unsigned, b0[pl]);
                __hip_atomic_store(mine + (int)threadIdx.x + q * NT, ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        int gl[NGAT], gs[NGAT];
#pragma unroll
        for (int q = 0; q < NGAT; ++q) { gl[q] = tab_gl[q * NT + (int)threadIdx.x]; gs[q] = tab_gs[q * NT + (int)threadIdx.x]; }
        unsigned gv[NGAT];
        const unsigned long long t0 = wall_clock64();
        const unsigned long long bound = grp == 0 ? pa.first_timeout_ticks : pa.timeout_ticks;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < NGAT; ++q)
                if (gl[q] >= 0) {
                    const unsigned long long x = __hip_atomic_load(half + gs[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gv[q] = (unsigned)x;
                    ok &= (unsigned)(x >> 32) == epoch;
                }
            if (__all(ok)) break;
            // give up when the wait is over its bound (the first to do so raises the abort flag below) or when another workgroup
            // already has: a launch whose workgroups are not all resident ends within the bound instead of hanging the GPU
            if (wall_clock64() - t0 > bound ||
                __hip_atomic_load(pa.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (failed) {
            if (threadIdx.x % WAVE == 0) *wg_abort = 1;
        } else {
#pragma unroll
            for (int q = 0; q < NGAT; ++q)
                if (gl[q] >= 0) b0[gl[q]] = __builtin_bit_cast(T, gv[q]);
        }
        lds_barrier();
        if (*wg_abort) {
            // ABORT: nothing this launch was asked for is written (no adjoint frame, no partial row) -- the host learns it from its
            // status slot and runs the launch-per-group sweep instead (or, without the handshake, reports the error at the next
            // entry point): never a poisoned result
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(pa.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(pa.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && pa.host) {
                    __hip_atomic_store(pa.host + 1, grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 2, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }

    // ---- once per rollout: diffusion-coefficient sums and the 20 moments of this tile -> its partial row -------------------
    lds_barrier();
    double* red = reinterpret_cast<double*>(smem_raw);     // state buffers are dead now: [2][NT / WAVE] doubles
    constexpr int NW = NT / WAVE;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double r = wave_sum_to_last(acc_c[s]);
        if (lane == REDUCE_LANE) red[s * NW + wave] = r;
    }
    lds_barrier();
    if (threadIdx.x < 2) {
        double sum = 0.0;
        for (int w = 0; w < NW; ++w) sum += red[(int)threadIdx.x * NW + w];
        partials[(long)blockIdx.x * np + P_COEF + (int)threadIdx.x] += sum;
    }
    // the moments sit transposed in LDS ([moment][lane], complete: the barrier above waited for the LDS adds): 16 lanes per
    // moment add NT / 16 of them each and fold with four DPP steps (as the float64 tile sweep does)
    if (threadIdx.x < 320) {
        const int mm = (int)threadIdx.x >> 4, part = (int)threadIdx.x & 15;
        const double* row = lacc + mm * NT + part;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int k = 0; k < NT; k += 64) {
            const double v0 = row[k], v1 = row[k + 16], v2 = row[k + 32], v3 = row[k + 48];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        double a = (a0 + a1) + (a2 + a3);
        a += dpp_mov<0x111, 0xF>(a);
        a += dpp_mov<0x112, 0xF>(a);
        a += dpp_mov<0x114, 0xF>(a);
        a += dpp_mov<0x118, 0xF>(a);
        if (part == 15) partials[(long)blockIdx.x * np + P_W + mm] += a;
    }
}


// ------------------------------------------------------------------------------------------------
// PERSISTENT sweep for the SMALL-TILE regime (round 4): grids below ~300^2 -- the reference's own 100^2 x 200
// (train_2drd.py:597-636) among them -- run on 32 x 8 tiles of 256 lanes with the SPLIT schedule (every adjoint frame is stored,
// the 20 moments come from one time-parallel pass afterwards), ragged edge tiles included.  A K = 4 sweep launch there is 7.3 us
// of which the four sub-steps are under two: the rest is the kernel boundary and the cold window load.  Same idea as
// pi_adj2d_persist_kernel -- resident workgroups, adjoint tile kept in LDS, data-tagged granules -- with two simplifications the
// small tiles allow: a tile publishes ALL its BX x BY values (its 2K-wide border band would be the whole tile anyway), and the
// gather tables are built from global coordinates (owner tile = (gy / BY, gx / BX) after the periodic wrap), which makes ragged
// grids, halos that reach across two neighbours and grids of a single tile column the same code.  The out-of-grid part of a ragged
// edge tile holds the periodic duplicates of the first columns / rows exactly as the launch-per-group kernel's window does; they
// are recomputed there like a halo and never published.  Sub-steps = the device functions of pi_adj2d_tile_kernel<MOM = false>:
// adjoint frames and dL/dh0 bit-identical.
// ------------------------------------------------------------------------------------------------
// Data-tagged granules by value type: ONE 16-byte write-through (sc1) store publishes a granule, ONE 16-byte sc1 load reads it
// (requests are what a hand-over costs, not bytes).  float64 (round 5, configs[2]): {lo32, tag, hi32, tag} = one value as two
// self-validating 8-byte words.  float32 (round 6): {value 0, tag, value 1, tag} = TWO x-adjacent values -- the 8-byte {tag, value}
// words of rounds 3-5 cost a request per handed-over float (3 + 5 per lane and group; now 2 + 3), VERDICT r5 #2.  The reader
// accepts when BOTH tags match, so a torn pair is just "not yet".  PI_GRANULE_PAIR=0 keeps the 8-byte float32 words (A/B builds of
// the harnesses only; the library is built with pairs).
#ifndef PI_GRANULE_PAIR
#define PI_GRANULE_PAIR 1
#endif
typedef unsigned pi_v4u __attribute__((ext_vector_type(4)));
template <typename T> struct GranuleIO;
#if PI_GRANULE_PAIR
template <> struct GranuleIO<float> {
    using Raw = pi_v4u;
    static constexpr int BYTES = 16, VALS = 2;
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ GranuleIO(void* outbox, size_t bytes) : rs(__builtin_amdgcn_make_buffer_rsrc(outbox, 0, (int)bytes, 0x00020000)) {}
    // src: the pair in LDS (8-byte aligned: even window column in either state buffer)
    __device__ __forceinline__ void put(size_t idx, unsigned epoch, const float* src) const
    {
        const Pack<float, 2> v = ld<float, 2>(src);
        const pi_v4u w = {__builtin_bit_cast(unsigned, v.v[0]), epoch, __builtin_bit_cast(unsigned, v.v[1]), epoch};
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)(idx * 16), 0, /*aux: sc1*/ 16);
    }
    __device__ __forceinline__ Raw get(size_t idx) const { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * 16), 0, 16); }
    static __device__ __forceinline__ bool ok(Raw x, unsigned epoch) { return x.y == epoch && x.w == epoch; }
    static __device__ __forceinline__ void land(Raw x, float* dst)
    {
        // (scalars first: __builtin_bit_cast applied to an element of an ext_vector read element 0 for every element -- hipcc 7.2)
        const unsigned a = x.x, b = x.z;
        st<float, 2>(dst, Pack<float, 2>{{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)}});
    }
};
#else
template <> struct GranuleIO<float> {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    using Raw = unsigned long long;
    static constexpr int BYTES = 8, VALS = 1;
    gu64* base;
    __device__ __forceinline__ GranuleIO(void* outbox, size_t) : base((gu64*)outbox) {}
    __device__ __forceinline__ void put(size_t idx, unsigned epoch, const float* src) const
    {
        __hip_atomic_store(base + idx, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, *src), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ Raw get(size_t idx) const { return __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ bool ok(Raw x, unsigned epoch) { return (unsigned)(x >> 32) == epoch; }
    static __device__ __forceinline__ void land(Raw x, float* dst) { *dst = __builtin_bit_cast(float, (unsigned)x); }
};
#endif
template <> struct GranuleIO<double> {
    using Raw = pi_v4u;
    static constexpr int BYTES = 16, VALS = 1;
    __amdgpu_buffer_rsrc_t rs;
    // (the descriptor is built from kernel arguments only: wave-uniform by construction)
    __device__ __forceinline__ GranuleIO(void* outbox, size_t bytes) : rs(__builtin_amdgcn_make_buffer_rsrc(outbox, 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ void put(size_t idx, unsigned epoch, const double* src) const
    {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, *src);
        const pi_v4u w = {(unsigned)b, epoch, (unsigned)(b >> 32), epoch};
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)(idx * 16), 0, /*aux: sc1*/ 16);
    }
    __device__ __forceinline__ Raw get(size_t idx) const { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx * 16), 0, 16); }
    static __device__ __forceinline__ bool ok(Raw x, unsigned epoch) { return x.y == epoch && x.w == epoch; }
    static __device__ __forceinline__ void land(Raw x, double* dst) { *dst = __builtin_bit_cast(double, ((unsigned long long)x.z << 32) | x.x); }
};

// the granule of the small-tile resident kernels: pairs (GranuleIO<float>) or the 8-byte {tag, value} words of rounds 4-5
template <typename T, bool PAIRS> struct SmallGranule {
    using Raw = unsigned long long;
    static __device__ __forceinline__ bool ok(Raw x, unsigned epoch) { return (unsigned)(x >> 32) == epoch; }
    static __device__ __forceinline__ void land(Raw x, T* dst) { *dst = __builtin_bit_cast(T, (unsigned)x); }
};
template <typename T> struct SmallGranule<T, true> {
    using Raw = typename GranuleIO<T>::Raw;
    static __device__ __forceinline__ bool ok(Raw x, unsigned epoch) { return GranuleIO<T>::ok(x, epoch); }
    static __device__ __forceinline__ void land(Raw x, T* dst) { GranuleIO<T>::land(x, dst); }
};
// HALFS (round 6): half-strips, NT = twice the lanes -- 7 | 5 | 4 | 2 waves busy in the four sub-steps of a 32 x 8 tile instead of
// 4 | 3 | 2 | 1 (one per SIMD, each alone with its LDS latency), and half the granule requests per lane in the hand-over.
template <typename T, int K, int BX, int BY, int NT, bool HALFS = false>
__global__ void __launch_bounds__(NT)
pi_adj2d_persist_small_kernel(const T* __restrict__ hframe_t, const T* __restrict__ gframe_t, T* __restrict__ aframe_t,
                              long frame_stride, T* __restrict__ g_h0, double* __restrict__ partials, int np,
                              const T* __restrict__ P_in, TileGeom g, PersistArgs pa)
{
    static_assert(sizeof(T) == 4 && K % 2 == 0, "float32; an even number of sub-steps leaves the state in buffer 0");
    using TL = Tile<K, BX, BY>;
    constexpr int HW = 2 * K, LXW = TL::LX, LYW = TL::LY;
    constexpr int OWN = BX * BY;                                         // values per species a tile publishes
    constexpr int RINGH = LXW * LYW - OWN;                               // halo values per species
    // PAIRS (round 6, with the half-strip kernels): the tile and its ring travel as 16-byte granules of two x-adjacent values
    // (GranuleIO<float>) -- tile origins, the halo width and the grid width are even, so a pair never straddles two owners or the wrap
    constexpr bool PAIRS = HALFS && GranuleIO<T>::VALS == 2;
    constexpr int GV = PAIRS ? 2 : 1;
    static_assert(OWN % (2 * GV) == 0 && RINGH % GV == 0 && BX % GV == 0 && HW % GV == 0, "granules of x-adjacent values");
    constexpr int NPUB = (2 * OWN / GV + NT - 1) / NT, NGAT = (2 * RINGH / GV + NT - 1) / NT;
    constexpr bool PRE = PI_TILE_ADJ_PIPE && TL::region_n(0) / (HALFS ? 2 : 4) <= NT;
    static_assert(!HALFS || (PRE && PI_PERSIST_GEO != 0), "half-strips: prefetched operands, geometry words");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int tiles_y = (g.H + BY - 1) / BY;
    const int tyi = tile / g.tiles_x, txi = tile % g.tiles_x;
    const int ty0 = tyi * BY, tx0 = txi * BX;
    const int ntiles = g.tiles_x * tiles_y;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* outbox = (gu64*)pa.outbox;
    const GranuleIO<T> gio(pa.outbox, (size_t)2 * (size_t)ntiles * (2 * OWN) * 8);             // (PAIRS; same bytes either way)
    // LDS: state buffers | int tables | abort word
    int* tab_gl = reinterpret_cast<int*>(smem_raw + tile_state_bytes<T, K, BX, BY>());      // [NGAT][NT]: LDS position of a halo value
    int* tab_gs = tab_gl + NGAT * NT;                                                       // [NGAT][NT]: granule index inside a parity half
    unsigned* tab_geo = reinterpret_cast<unsigned*>(tab_gs + NGAT * NT);                    // [K][NT]: the lane's strip in each sub-step
    int* wg_abort = reinterpret_cast<int*>(tab_geo + K * NT);
    static_assert(K == 4, "geometry rows below");
    unsigned hoff[K] = {0u, 0u, 0u, 0u};                                                    // HALFS: operand byte offsets per sub-step
    if constexpr (HALFS) {
        tab_geo[0 * NT + (int)threadIdx.x] = persist_half_geo_word<K, BX, BY, NT, 0, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[1 * NT + (int)threadIdx.x] = persist_half_geo_word<K, BX, BY, NT, 1, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[2 * NT + (int)threadIdx.x] = persist_half_geo_word<K, BX, BY, NT, 2, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[3 * NT + (int)threadIdx.x] = persist_half_geo_word<K, BX, BY, NT, 3, PART_FULL, 0>(g, ty0, tx0);
        hoff[0] = persist_half_off<T, K, BX, BY, NT, 0, PART_FULL, 0>(g, ty0, tx0);
        hoff[1] = persist_half_off<T, K, BX, BY, NT, 1, PART_FULL, 0>(g, ty0, tx0);
        hoff[2] = persist_half_off<T, K, BX, BY, NT, 2, PART_FULL, 0>(g, ty0, tx0);
        hoff[3] = persist_half_off<T, K, BX, BY, NT, 3, PART_FULL, 0>(g, ty0, tx0);
#pragma unroll
        for (int m = 0; m < K; ++m) asm volatile("" : "+v"(hoff[m]));
    } else {
        tab_geo[0 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 0, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[1 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 1, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[2 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 2, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[3 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 3, PART_FULL, 0>(g, ty0, tx0);
    }
    if (threadIdx.x == 0) {                                                                 // residency roll call (pi_adj2d_persist_kernel)
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(pa.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)ntiles && pa.host) __hip_atomic_store(pa.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int q = 0; q < NGAT; ++q) {
        const int r = (int)threadIdx.x + q * NT;
        int gl = -1, gs = 0;
        if (r < 2 * RINGH / GV) {
            const int sp = r / (RINGH / GV), e = (r - sp * (RINGH / GV)) * GV;       // (e: first value of the granule)
            int wy, wx;                                    // ring positions row-major over the window, skipping the centre
            if (e < HW * LXW) { wy = e / LXW; wx = e - wy * LXW; }
            else if (e < HW * LXW + BY * 2 * HW) { const int m = e - HW * LXW; wy = HW + m / (2 * HW); const int c = m % (2 * HW); wx = c < HW ? c : BX + c; }
            else { const int m = e - HW * LXW - BY * 2 * HW; wy = HW + BY + m / LXW; wx = m % LXW; }
            int gy = (ty0 + wy - HW) % g.H, gx = (tx0 + wx - HW) % g.W;                     // the global point, periodic
            gy += gy < 0 ? g.H : 0;
            gx += gx < 0 ? g.W : 0;
            const int nty = gy / BY, ntx = gx / BX;
            gl = sp * TL::PLANE + wy * LXW + wx;
            gs = ((nty * g.tiles_x + ntx) * (2 * OWN) + sp * OWN + (gy - nty * BY) * BX + (gx - ntx * BX)) / GV;
        }
        tab_gl[q * NT + (int)threadIdx.x] = gl;
        tab_gs[q * NT + (int)threadIdx.x] = gs;
    }
    // group 0 starts from the adjoint frame in memory, like a launch of pi_adj2d_tile_kernel
    WindowLoader<T, K, BX, BY, NT> wl;
    wl.issue(aframe_t, g, ty0, tx0);
    StripOps<T> ops0;
    StripAddr sa[K];
    unsigned gmask = persist_mask<K>(pa, pa.t_top);
    if constexpr (HALFS) {
        persist_load_ops_half<T>(ops0, hframe_t - frame_stride, gmask & 1u ? gframe_t - frame_stride : nullptr, g, hoff[0]);
    } else if constexpr (PRE) {
        strip_addr_table<K, BX, BY, NT, 0>(sa, g, ty0, tx0);
        adj_load_ops<T, K, BX, BY, NT, 0>(ops0, 0, hframe_t - frame_stride, gmask & 1u ? gframe_t - frame_stride : nullptr, g, ty0, tx0,
                                          &sa[0]);
    }
    wl.commit(b0);
    lds_barrier();
    double acc_c[2] = {0.0, 0.0};
    bool failed = false;
    TileMoments<T, false> mom;
    // (round 6, as in the resident forwards: the parameter block held in registers -- behind the barriers' memory clobbers every
    // sub-step re-read what JacPairs does not hold with scalar loads on its critical path; 191 -> ~230 of 256 registers)
    T Ph[NPOLY];
#pragma unroll
    for (int i = 0; i < NPOLY; ++i) {
        T x = P_in[i];
        asm volatile("" : "+v"(x));
        Ph[i] = x;
    }
    const T* P = Ph;
    JacPairs<T> jp;
    jac_pairs_load<T>(jp, P);
    for (int grp = 0; grp < pa.ngroups; ++grp) {
        const long go = -(long)grp * K * frame_stride;     // this group's frame t relative to the top frame
        const bool last = grp + 1 == pa.ngroups;
        // every adjoint frame goes to memory (the moments pass reads them); the last one of the sweep may be dL/dh0 itself.  The
        // frame a group ends on is stored AFTER the tile has been published: the neighbours wait for the granules, nobody for it
        adj_substeps<T, POLY, K, BX, BY, NT, 0, PRE, false, PI_PERSIST_GEO != 0, HALFS>(b0, b1, hframe_t + go, gframe_t + go, aframe_t + go,
                                                                                        frame_stride, gmask, g_h0, g_h0 && last ? K : 0, g, ty0,
                                                                                        tx0, P, acc_c, ops0, mom, sa, nullptr, last, tab_geo, &jp,
                                                                                        hoff);
        if (last) break;
        const long gn = go - (long)K * frame_stride;
        gmask = persist_mask<K>(pa, pa.t_top - K * (grp + 1));
        if constexpr (HALFS)
            persist_load_ops_half<T>(ops0, hframe_t + gn - frame_stride, gmask & 1u ? gframe_t + gn - frame_stride : nullptr, g, hoff[0]);
        else if constexpr (PRE)
            adj_load_ops<T, K, BX, BY, NT, 0>(ops0, 0, hframe_t + gn - frame_stride, gmask & 1u ? gframe_t + gn - frame_stride : nullptr, g,
                                              ty0, tx0, &sa[0]);
        // (sub-step K - 1 ended with a barrier: buffer 0 is complete)
        // ---- hand-over: publish my tile, gather my ring ----
        const unsigned epoch = (unsigned)grp + 1u;
        gu64* half = outbox + (size_t)(epoch & 1u) * (size_t)ntiles * (2 * OWN);
        gu64* mine = half + (size_t)tile * (2 * OWN);
        const size_t halfu = (size_t)(epoch & 1u) * (size_t)ntiles * (2 * OWN / GV);      // ... in granules (PAIRS)
#pragma unroll
        for (int q = 0; q < NPUB; ++q) {
            const int i = (int)threadIdx.x + q * NT;
            if (i < 2 * OWN / GV) {
                const int sp = i / (OWN / GV), e = (i - sp * (OWN / GV)) * GV, y = e / BX, x = e - y * BX;
                const T* src = b0 + sp * TL::PLANE + (HW + y) * LXW + HW + x;
                if constexpr (PAIRS) {
                    gio.put(halfu + (size_t)tile * (2 * OWN / GV) + (size_t)i, epoch, src);
                } else {
                    const unsigned v = __builtin_bit_cast(unsigned, *src);
                    __hip_atomic_store(mine + i, ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        tile_store<T, K, BX, BY, NT, true>(b0, aframe_t + gn, g, ty0, tx0);       // frame t - K of this group (buffer 0: K even)
        for (int w = 0; w < pa.pause; ++w) __builtin_amdgcn_s_sleep(1);            // (granules asked for too early come back stale)
        int gl[NGAT], gs[NGAT];
        typename SmallGranule<T, PAIRS>::Raw gx[NGAT];
        auto fetch = [&](int idx) {
            if constexpr (PAIRS) return gio.get(halfu + (size_t)idx);
            else return __hip_atomic_load(half + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
#pragma unroll
        for (int q = 0; q < NGAT; ++q) {
            gl[q] = tab_gl[q * NT + (int)threadIdx.x];
            gs[q] = tab_gs[q * NT + (int)threadIdx.x];
            gx[q] = fetch(gs[q]);                                                                     // (lanes without: granule 0)
        }
        const unsigned long long t0 = wall_clock64();
        const unsigned long long bound = grp == 0 ? pa.first_timeout_ticks : pa.timeout_ticks;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < NGAT; ++q)
                if (gl[q] >= 0) ok &= SmallGranule<T, PAIRS>::ok(gx[q], epoch);
            if (__all(ok)) break;
            if (wall_clock64() - t0 > bound ||
                __hip_atomic_load(pa.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int q = 0; q < NGAT; ++q)                      // only what has not arrived yet is asked for again
                if (gl[q] >= 0 && !SmallGranule<T, PAIRS>::ok(gx[q], epoch)) gx[q] = fetch(gs[q]);
        }
        if (failed) {
            if (threadIdx.x % WAVE == 0) *wg_abort = 1;
        } else {
#pragma unroll
            for (int q = 0; q < NGAT; ++q)
                if (gl[q] >= 0) SmallGranule<T, PAIRS>::land(gx[q], b0 + gl[q]);
        }
        lds_barrier();
        if (*wg_abort) {
            // ABORT (see pi_adj2d_persist_kernel): the frames of the groups done so far ARE in memory -- they are the same values the
            // launch-per-group sweep writes -- but this launch reports failure and the host re-runs the whole sweep
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(pa.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(pa.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && pa.host) {
                    __hip_atomic_store(pa.host + 1, grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 2, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
    // once per rollout: the two diffusion-coefficient sums of this tile -> its partial row
    lds_barrier();
    double* red = reinterpret_cast<double*>(smem_raw);     // state buffers are dead now
    constexpr int NW = NT / WAVE;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double r = wave_sum_to_last(acc_c[s]);
        if (lane == REDUCE_LANE) red[s * NW + wave] = r;
    }
    lds_barrier();
    if (threadIdx.x < 2) {
        double sum = 0.0;
        for (int w = 0; w < NW; ++w) sum += red[(int)threadIdx.x * NW + w];
        partials[(long)blockIdx.x * np + P_COEF + (int)threadIdx.x] += sum;
    }
}


// ------------------------------------------------------------------------------------------------
// PERSISTENT fused sweep, SPLIT flavour (round 4).  The hand-over of pi_adj2d_persist_kernel -- publish, ~1 us until the
// granules are visible on the memory side, ~1.8 us for 2560 uncached 8-byte loads per workgroup, the slowest neighbour -- was
// 3.8-4.8 of every 8.8-10.8 us group and nothing ran under it.  A third of a group's work does not need the halo at all:
// sub-step m restricted to the centred square I_m of side B - 4 (m + 1) depends on the workgroup's own tile only (the
// "pyramid": 196 + 144 + 100 + 64 of 1464 strips); the rest of each sub-step's region is a ring A_m 2K points thick.
// Device timelines (tools/persist_dev.hip) say what a barrier interval costs: ~1.15 us for ONE wave per SIMD whatever the
// strip count (a wave's ~400 instructions issue in ~1 us), ~1.6-1.8 us for two -- so doing pyramid and ring in eight
// intervals of their own (built first: 12.5 us per group against 10.8) loses what it hides.  What pays is to run the two
// chains CONCURRENTLY on different waves of the same interval, and only as much pyramid up front as the hand-over needs:
//     publish | P0: I_0 | request the ring granules | P1: I_1 | check tags, ring -> LDS |
//     P2: I_2 (waves 0-1) + A_0 (waves 2-6) | P3: A_1 | P4: A_2 | P5: I_3 (wave 0) + A_3 (waves 1-3) | publish ...
// (seven wave-slots on the busiest SIMD, as the four sub-steps of the unsplit kernel have)
// The ping-pong buffers need no extra storage: I_m overwrites only the centre of the buffer A_(m-1) no longer reads (A_(m-1)
// reads level m - 2 outside the square of side B - 4 m - 4, I_m writes level m inside it), and the two parts of a mixed pass
// write disjoint regions of the same buffer.  Every point is computed by the same device function with the same operands as
// in the launch-per-group kernel: adjoint state and dL/dh0 stay bit-identical; only the order in which a lane's moment sums
// are accumulated changes (gradients to summation round-off, as between any two of the sweep's flavours).
// ------------------------------------------------------------------------------------------------
// BYTE offsets (inside one species plane) of the two halves of a lane's strip: six passes per group keep six of these
// alive, and 32-bit offsets next to wave-uniform frame bases (`global_load v, v_off, s[base]`) cost neither the 64-bit
// per-lane address registers nor the address arithmetic of the StripAddr form
struct StripOff { unsigned o0, o1; };

template <typename T, int K, int BX, int BY, int NT, int M, int PART, int TID0>
__device__ __forceinline__ StripOff persist_strip_off(const TileGeom& g, int ty0, int tx0)
{
    const StripAddr a = strip_addr<K, BX, BY, NT, M, PART, TID0>(0, g, ty0, tx0);
    return StripOff{(unsigned)a.e0 * (unsigned)sizeof(T), (unsigned)a.e1 * (unsigned)sizeof(T)};
}

// pointwise operands of one strip through scalar frame bases (see adj_load_ops for why the loads are unconditional)
template <typename T>
__device__ __forceinline__ void persist_load_ops(StripOps<T>& o, const T* __restrict__ hfr, const T* __restrict__ gfr,
                                                 const TileGeom& g, const StripOff& so)
{
    const T* gsrc = gfr ? gfr : hfr;
    const char* hu = reinterpret_cast<const char*>(hfr);
    const char* hv = reinterpret_cast<const char*>(hfr + g.ss);
    const char* ju = reinterpret_cast<const char*>(gsrc);
    const char* jv = reinterpret_cast<const char*>(gsrc + g.ss);
    const unsigned o0 = so.o0, o1 = so.o1;
    const Pack<T, 2> a = *reinterpret_cast<const Pack<T, 2>*>(hu + o0), b = *reinterpret_cast<const Pack<T, 2>*>(hu + o1);
    const Pack<T, 2> c = *reinterpret_cast<const Pack<T, 2>*>(hv + o0), d = *reinterpret_cast<const Pack<T, 2>*>(hv + o1);
    const Pack<T, 2> a2 = *reinterpret_cast<const Pack<T, 2>*>(ju + o0), b2 = *reinterpret_cast<const Pack<T, 2>*>(ju + o1);
    const Pack<T, 2> c2 = *reinterpret_cast<const Pack<T, 2>*>(jv + o0), d2 = *reinterpret_cast<const Pack<T, 2>*>(jv + o1);
    o.u[0] = a.v[0]; o.u[1] = a.v[1]; o.u[2] = b.v[0]; o.u[3] = b.v[1];
    o.v[0] = c.v[0]; o.v[1] = c.v[1]; o.v[2] = d.v[0]; o.v[3] = d.v[1];
    o.ju[0] = a2.v[0]; o.ju[1] = a2.v[1]; o.ju[2] = b2.v[0]; o.ju[3] = b2.v[1];
    o.jv[0] = c2.v[0]; o.jv[1] = c2.v[1]; o.jv[2] = d2.v[0]; o.jv[3] = d2.v[1];
}

// The hand-over tables of the 32 x 32 resident kernels, in units of one granule (GV = values per granule, x-adjacent): a tile's
// border band of width HW is numbered by band_index (rows of B, then rows of 2 HW values: even x <-> even index), the halo ring
// row-major over the window (rows of LXW, then rows of 2 HW: even window column <-> even index); a tile origin is a multiple of
// B, so a pair never straddles two owners.
template <int K, int BX, int GV, int NT>
struct HandOver {
    using TL = Tile<K, BX, BX>;
    static constexpr int HW = 2 * K, LXW = TL::LX;
    static constexpr int BANDH = BX * BX - (BX - 2 * HW) * (BX - 2 * HW);      // border values per species
    static constexpr int RINGH = LXW * LXW - BX * BX;                          // halo values per species
    static_assert(BANDH % GV == 0 && RINGH % GV == 0 && HW % GV == 0 && BX % GV == 0, "granules of x-adjacent values");
    static constexpr int BANDU = BANDH / GV, RINGU = RINGH / GV;               // granules per species
    static constexpr int NPUB = (2 * BANDU + NT - 1) / NT, NGAT = (2 * RINGU + NT - 1) / NT;
    // LDS position (inside a state buffer) of the first value of band granule i (-1: none)
    static __device__ __forceinline__ int pub_pos(int i)
    {
        if (i >= 2 * BANDU) return -1;
        const int sp = i / BANDU, e = (i - sp * BANDU) * GV;
        int y, x;                                          // inverse of band_index
        if (e < HW * BX) { y = e / BX; x = e - y * BX; }
        else if (e < 2 * HW * BX) { const int m = e - HW * BX; y = BX - HW + m / BX; x = m % BX; }
        else { const int m = e - 2 * HW * BX; y = HW + m / (2 * HW); const int c = m % (2 * HW); x = c < HW ? c : BX - 2 * HW + c; }
        return sp * TL::PLANE + (HW + y) * LXW + HW + x;
    }
    // ring granule r: gl = LDS position of its first value (-1: none), gs = granule index inside a parity half of the outbox
    static __device__ __forceinline__ void gat_pos(int r, const TileGeom& g, int ty0, int tx0, int tiles_y, int& gl, int& gs)
    {
        gl = -1; gs = 0;
        if (r >= 2 * RINGU) return;
        const int sp = r / RINGU, e = (r - sp * RINGU) * GV;
        int wy, wx;                                        // ring positions row-major over the window, skipping the centre
        if (e < HW * LXW) { wy = e / LXW; wx = e - wy * LXW; }
        else if (e < HW * LXW + BX * 2 * HW) { const int m = e - HW * LXW; wy = HW + m / (2 * HW); const int c = m % (2 * HW); wx = c < HW ? c : BX + c; }
        else { const int m = e - HW * LXW - BX * 2 * HW; wy = HW + BX + m / LXW; wx = m % LXW; }
        const int gy = ty0 + wy - HW, gx = tx0 + wx - HW;                       // global point (may wrap)
        const int nty = ((gy + g.H) / BX) % tiles_y, ntx = ((gx + g.W) / BX) % g.tiles_x;
        const int ly = (gy + g.H) % BX, lx = (gx + g.W) % BX;
        gl = sp * TL::PLANE + wy * LXW + wx;
        gs = (nty * g.tiles_x + ntx) * (2 * BANDU) + sp * BANDU + band_index<BX, HW>(ly, lx) / GV;
    }
};

// int rows of NT the split sweep keeps in LDS behind the moments: 13 of hand-over tables + 6 of strip geometry (+ the abort word)
constexpr int PERSIST_SPLIT_TABLE_ROWS = 19;

// HALF-STRIP passes of the split sweep (float32, round 6).  P3, P4, P5 -- A_1 | A_2 | I_3 + A_3 -- have 4, 3.5 and 4 wave-strips for
// eight waves: one wave per SIMD issues a 450-instruction strip at the single-wave rate (1.04 us per pass) while the other wave of the
// SIMD idles; P2, where two strips share a SIMD, does them in 0.8 us each.  On half-strips these passes occupy all eight waves
// (512 / 448 / 128 + 384 lanes) with half the points per lane.  P0 .. P2 keep whole strips.
#ifndef PI_ADJ_PERSIST_PAUSE
#define PI_ADJ_PERSIST_PAUSE 0          // split sweep: s_sleep units before the ring request (round 6: 0 / 8 / 32 measured, see profiles)
#endif
#ifndef PI_SWEEP_HALF
#define PI_SWEEP_HALF 1
#endif
#ifndef PI_SWEEP_HALF_MASK
#define PI_SWEEP_HALF_MASK 0x39         // bit p: pass p runs on half-strips: P0, P3, P4, P5 (P2 mixes two sub-steps and keeps whole strips;
                                        // P1 on half-strips measured slower: 1.79 -> 1.81 us per step)
#endif
template <typename T, int PASS> struct sweep_half_pass {
#ifndef PI_SWEEP_HALF_F64
#define PI_SWEEP_HALF_F64 1
#endif
#ifndef PI_SWEEP_HALF_MASK_F64
#define PI_SWEEP_HALF_MASK_F64 0x39         // (P1 too, 0x3B: 3.29 -> 3.29 us per step, no change)
#endif
    static constexpr bool value = PI_SWEEP_HALF != 0 && (sizeof(T) == 4 || PI_SWEEP_HALF_F64 != 0) && PASS != 2 &&
                                  ((((sizeof(T) == 8 ? (PI_SWEEP_HALF_MASK_F64) : (PI_SWEEP_HALF_MASK))) >> PASS) & 1) != 0;
};
constexpr int SWEEP_HALF_P5_SPLIT = 128;                   // P5 on half-strips: I_3 = 64 strips on lanes 0 .. 127, A_3 = 192 from lane 128 on

// The six passes of a group: which sub-step / part the waves below SPLIT work on (M1, PART1; strips indexed from lane 0) and
// which the waves from SPLIT on (M2, PART2; strips indexed from lane SPLIT); SPLIT == NT: one part only.
template <int PASS> struct PersistPass;
template <> struct PersistPass<0> { static constexpr int M1 = 0, P1 = PART_PYR, SPLIT = 1 << 30, M2 = 0, P2 = PART_PYR; };
template <> struct PersistPass<1> { static constexpr int M1 = 1, P1 = PART_PYR, SPLIT = 1 << 30, M2 = 1, P2 = PART_PYR; };
template <> struct PersistPass<2> { static constexpr int M1 = 2, P1 = PART_PYR, SPLIT = 128, M2 = 0, P2 = PART_ANN; };
template <> struct PersistPass<3> { static constexpr int M1 = 1, P1 = PART_ANN, SPLIT = 1 << 30, M2 = 1, P2 = PART_ANN; };
template <> struct PersistPass<4> { static constexpr int M1 = 2, P1 = PART_ANN, SPLIT = 1 << 30, M2 = 2, P2 = PART_ANN; };
template <> struct PersistPass<5> { static constexpr int M1 = 3, P1 = PART_PYR, SPLIT = 64, M2 = 3, P2 = PART_ANN; };

// the sub-step whose frame a lane's strip of pass PASS belongs to (wave-uniform)
template <int PASS>
__device__ __forceinline__ int persist_level(bool upper)
{
    using PP = PersistPass<PASS>;
    return upper ? PP::M2 : PP::M1;
}

template <typename T, int K, int BX, int BY, int NT, int PASS>
__device__ __forceinline__ StripOff persist_pass_off(const TileGeom& g, int ty0, int tx0, bool upper)
{
    using PP = PersistPass<PASS>;
    StripOff so;
    if constexpr (PP::SPLIT >= NT) {
        so = persist_strip_off<T, K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
    } else {
        const StripOff lo = persist_strip_off<T, K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
        const StripOff hi = persist_strip_off<T, K, BX, BY, NT, PP::M2, PP::P2, PP::SPLIT>(g, ty0, tx0);
        so = upper ? hi : lo;
    }
    asm volatile("" : "+v"(so.o0), "+v"(so.o1));           // computed in the prologue, not where it is consumed
    return so;
}

// ... of pass PASS for this lane (lanes from SPLIT on: the second part)
template <int K, int BX, int BY, int NT, int PASS>
__device__ __forceinline__ unsigned persist_pass_half_geo(const TileGeom& g, int ty0, int tx0)
{
    using PP = PersistPass<PASS>;
    static_assert(PASS != 2, "half-strip passes");
    if constexpr (PASS == 5) {
        const unsigned lo = persist_half_geo_word<K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
        const unsigned hi = persist_half_geo_word<K, BX, BY, NT, PP::M2, PP::P2, SWEEP_HALF_P5_SPLIT>(g, ty0, tx0);
        return (int)threadIdx.x >= SWEEP_HALF_P5_SPLIT ? hi : lo;
    } else {
        return persist_half_geo_word<K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
    }
}
template <typename T, int K, int BX, int BY, int NT, int PASS>
__device__ __forceinline__ StripOff persist_pass_half_off(const TileGeom& g, int ty0, int tx0)
{
    using PP = PersistPass<PASS>;
    StripOff so;
    if constexpr (PASS == 5) {
        const unsigned lo = persist_half_off<T, K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
        const unsigned hi = persist_half_off<T, K, BX, BY, NT, PP::M2, PP::P2, SWEEP_HALF_P5_SPLIT>(g, ty0, tx0);
        so.o0 = (int)threadIdx.x >= SWEEP_HALF_P5_SPLIT ? hi : lo;
    } else {
        so.o0 = persist_half_off<T, K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
    }
    so.o1 = so.o0;
    asm volatile("" : "+v"(so.o0));
    return so;
}

template <int K, int BX, int BY, int NT, int PASS>
__device__ __forceinline__ unsigned persist_pass_geo(const TileGeom& g, int ty0, int tx0)
{
    using PP = PersistPass<PASS>;
    if constexpr (PP::SPLIT >= NT) {
        return persist_geo_word<K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
    } else {
        const unsigned lo = persist_geo_word<K, BX, BY, NT, PP::M1, PP::P1, 0>(g, ty0, tx0);
        const unsigned hi = persist_geo_word<K, BX, BY, NT, PP::M2, PP::P2, PP::SPLIT>(g, ty0, tx0);
        return (int)threadIdx.x >= PP::SPLIT ? hi : lo;
    }
}

// one pass: request the NEXT pass's pointwise operands (frames hn / gn, chosen by the caller for this wave) into `ahead`, compute
// with `ops`, barrier.  The caller alternates two operand sets (six passes per group: the roles repeat), nothing is copied.
template <typename T, int K, int BX, int BY, int NT, int PASS>
__device__ __forceinline__ void persist_pass(T* b0, T* b1, const T* const (&hf)[K], const T* const (&gf)[K],
                                             const T* __restrict__ hn, const T* __restrict__ gn, const StripOff& so_next,
                                             const TileGeom& g, int ty0, int tx0, const T* __restrict__ P, double (&acc_c)[2],
                                             const StripOps<T>& ops, StripOps<T>& ahead, TileMoments<T, true>& mom, bool upper,
                                             const unsigned* geo, const JacPairs<T>& jp, double* lacc = nullptr)
{
    using PP = PersistPass<PASS>;
    constexpr bool GEO = PI_PERSIST_GEO != 0;
    constexpr bool HALF_HERE = GEO && sweep_half_pass<T, PASS>::value, HALF_NEXT = GEO && sweep_half_pass<T, (PASS + 1) % 6>::value;
    if constexpr (HALF_NEXT) persist_load_ops_half<T>(ahead, hn, gn, g, so_next.o0);
    else persist_load_ops<T>(ahead, hn, gn, g, so_next);
    constexpr int LACC = sizeof(T) == 8 ? NT / 2 : NT;     // float64: moments straight into the shared LDS rows (adj_substep)
    if constexpr (HALF_HERE) {
        // every lane a half-strip of sub-step M (P5: both parts are sub-step 3 and share the injection frame)
        static_assert(PP::M1 == PP::M2 || PP::SPLIT >= NT, "half-strip passes: one sub-step");
        constexpr int M = PP::M1;
        adj_substep<T, POLY, K, BX, BY, NT, M, true, true, PP::P1, 0, GEO, LACC, true>((M & 1) ? b1 : b0, (M & 1) ? b0 : b1, hf[M], gf[M], g, ty0,
                                                                                       tx0, P, acc_c, ops, mom, lacc, geo, &jp);
    } else if constexpr (PP::SPLIT >= NT || GEO) {
        // With the geometry in a table the sub-step's body no longer depends on WHICH strips a wave works on: the two parts of a
        // mixed pass (same parity of M: same buffers) run the same instructions with their own table words and their own
        // injection frame -- one body instead of two behind a branch (the moments' 20 register pairs were copied at every merge).
        static_assert(PP::SPLIT >= NT || ((PP::M1 ^ PP::M2) & 1) == 0, "both parts read the same buffer");
        constexpr int M = PP::M1;
        const T* gfr = (PP::SPLIT < NT && upper) ? gf[PP::M2] : gf[M];
        adj_substep<T, POLY, K, BX, BY, NT, M, true, true, PP::P1, 0, GEO, LACC>((M & 1) ? b1 : b0, (M & 1) ? b0 : b1, hf[M], gfr, g, ty0,
                                                                                 tx0, P, acc_c, ops, mom, lacc, geo, &jp);
    } else if (!upper) {                                   // wave-uniform
        constexpr int M = PP::M1;
        adj_substep<T, POLY, K, BX, BY, NT, M, true, true, PP::P1, 0, GEO, LACC>((M & 1) ? b1 : b0, (M & 1) ? b0 : b1, hf[M], gf[M], g, ty0,
                                                                                 tx0, P, acc_c, ops, mom, lacc, geo, &jp);
    } else {
        constexpr int M = PP::M2;
        adj_substep<T, POLY, K, BX, BY, NT, M, true, true, PP::P2, PP::SPLIT, GEO, LACC>((M & 1) ? b1 : b0, (M & 1) ? b0 : b1, hf[M], gf[M],
                                                                                         g, ty0, tx0, P, acc_c, ops, mom, lacc, geo, &jp);
    }
#if PI_PIN_MOMENTS
    if constexpr (sizeof(T) == 4) {                        // (float64 keeps no moment registers: LDS rows)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) asm volatile("" : "+v"(mom.a[s][m]));
    }
#endif
    lds_barrier();
}

template <typename T, int K, int BX, int BY, int NT>
__global__ void __launch_bounds__(NT)
pi_adj2d_persist_split_kernel(const T* __restrict__ hframe_t, const T* __restrict__ gframe_t, T* __restrict__ aframe_t,
                              long frame_stride, T* __restrict__ g_h0, double* __restrict__ partials, int np,
                              const T* __restrict__ P, TileGeom g, PersistArgs pa)
{
    static_assert(BX == BY && K == 4 && BX == 32 && NT == 512, "32 x 32 tiles, four sub-steps, 8 waves");
    using TL = Tile<K, BX, BY>;
    using HO = HandOver<K, BX, GranuleIO<T>::VALS, NT>;                 // band / ring numbering in granules
    constexpr int BANDU = HO::BANDU, NPUB = HO::NPUB, NGAT = HO::NGAT;
    static_assert(NPUB + 2 * NGAT <= 13, "tables fit the LDS the host reserves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int tyi = tile / g.tiles_x, txi = tile % g.tiles_x, tiles_y = g.H / BY;
    const int ty0 = tyi * BY, tx0 = txi * BX;
    const int ntiles = g.tiles_x * tiles_y;
    const GranuleIO<T> gio(pa.outbox, (size_t)2 * (size_t)ntiles * (2 * BANDU) * GranuleIO<T>::BYTES);
    constexpr int LACC = sizeof(T) == 8 ? NT / 2 : NT;     // float64: two lanes share a moment slot (LDS atomics), see persist_pass
    // which half of a mixed pass this wave works on (wave-uniform, kept in a scalar register)
    const int wave_id = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool up2 = wave_id * WAVE >= PersistPass<2>::SPLIT, up5 = wave_id * WAVE >= PersistPass<5>::SPLIT;

    // LDS: state buffers | [20][LACC] doubles: the lanes' moments of all groups so far | int tables | abort word
    double* lacc = reinterpret_cast<double*>(smem_raw + tile_state_bytes<T, K, BX, BY>());
    int* tab_pub = reinterpret_cast<int*>(lacc + 20 * LACC);                // [NPUB][NT]: LDS position of a border value
    int* tab_gl = tab_pub + NPUB * NT;                                      // [NGAT][NT]: LDS position of a halo value
    int* tab_gs = tab_gl + NGAT * NT;                                       // [NGAT][NT]: granule index inside a parity half
    unsigned* tab_geo = reinterpret_cast<unsigned*>(tab_pub + 13 * NT);     // [6][NT]: the lane's strip in each pass (persist_geo_word)
    int* wg_abort = tab_pub + PERSIST_SPLIT_TABLE_ROWS * NT;
    if (threadIdx.x == 0) {                                                 // residency roll call (see pi_adj2d_persist_kernel)
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(pa.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)ntiles && pa.host) __hip_atomic_store(pa.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if constexpr (sweep_half_pass<T, 0>::value) tab_geo[0 * NT + (int)threadIdx.x] = persist_pass_half_geo<K, BX, BY, NT, 0>(g, ty0, tx0);
    else tab_geo[0 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 0>(g, ty0, tx0);
    if constexpr (sweep_half_pass<T, 1>::value) tab_geo[1 * NT + (int)threadIdx.x] = persist_pass_half_geo<K, BX, BY, NT, 1>(g, ty0, tx0);
    else tab_geo[1 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 1>(g, ty0, tx0);
    tab_geo[2 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 2>(g, ty0, tx0);
    if constexpr (sweep_half_pass<T, 3>::value) tab_geo[3 * NT + (int)threadIdx.x] = persist_pass_half_geo<K, BX, BY, NT, 3>(g, ty0, tx0);
    else tab_geo[3 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 3>(g, ty0, tx0);
    if constexpr (sweep_half_pass<T, 4>::value) tab_geo[4 * NT + (int)threadIdx.x] = persist_pass_half_geo<K, BX, BY, NT, 4>(g, ty0, tx0);
    else tab_geo[4 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 4>(g, ty0, tx0);
    if constexpr (sweep_half_pass<T, 5>::value) tab_geo[5 * NT + (int)threadIdx.x] = persist_pass_half_geo<K, BX, BY, NT, 5>(g, ty0, tx0);
    else tab_geo[5 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 5>(g, ty0, tx0);
    if ((int)threadIdx.x < LACC) {
#pragma unroll
        for (int m = 0; m < 20; ++m) lacc[m * LACC + (int)threadIdx.x] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < NPUB; ++q) tab_pub[q * NT + (int)threadIdx.x] = HO::pub_pos((int)threadIdx.x + q * NT);
#pragma unroll
    for (int q = 0; q < NGAT; ++q) {
        int gl, gs;
        HO::gat_pos((int)threadIdx.x + q * NT, g, ty0, tx0, tiles_y, gl, gs);
        tab_gl[q * NT + (int)threadIdx.x] = gl;
        tab_gs[q * NT + (int)threadIdx.x] = gs;
    }

    // group 0 starts from the adjoint frame in memory (window = tile + ring), like a launch of pi_adj2d_tile_kernel
    WindowLoader<T, K, BX, BY, NT> wl;
    wl.issue(aframe_t, g, ty0, tx0);
    // the lane's strip in each of the six passes (byte offsets of its pointwise operands)

    const StripOff so2 = persist_pass_off<T, K, BX, BY, NT, 2>(g, ty0, tx0, up2);
    auto pass_off = [&](auto pass_c, bool upper) {
        constexpr int PASS = decltype(pass_c)::value;
        if constexpr (sweep_half_pass<T, PASS>::value) return persist_pass_half_off<T, K, BX, BY, NT, PASS>(g, ty0, tx0);
        else return persist_pass_off<T, K, BX, BY, NT, PASS>(g, ty0, tx0, upper);
    };
    const StripOff so0 = pass_off(std::integral_constant<int, 0>{}, false);
    const StripOff so1 = pass_off(std::integral_constant<int, 1>{}, false);
    const StripOff so3 = pass_off(std::integral_constant<int, 3>{}, false);
    const StripOff so4 = pass_off(std::integral_constant<int, 4>{}, false);
    const StripOff so5 = pass_off(std::integral_constant<int, 5>{}, up5);
    unsigned gmask = persist_mask<K>(pa, pa.t_top);
    StripOps<T> ops, ops2;                                  // operands of the pass at hand / of the next one, alternating
    if constexpr (sweep_half_pass<T, 0>::value) persist_load_ops_half<T>(ops, hframe_t - frame_stride, gmask & 1u ? gframe_t - frame_stride : nullptr, g, so0.o0);
    else persist_load_ops<T>(ops, hframe_t - frame_stride, gmask & 1u ? gframe_t - frame_stride : nullptr, g, so0);
    wl.commit(b0);
    lds_barrier();
    double acc_c[2] = {0.0, 0.0};
    TileMoments<T, true> mom;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 10; ++m) mom.a[s][m] = typename MomAcc<T>::type{};
    JacPairs<T> jp;
    jac_pairs_load<T>(jp, P);
    for (int grp = 0; grp < pa.ngroups; ++grp) {
        const bool last = grp + 1 == pa.ngroups;
        const T* hb = hframe_t - (long)grp * K * frame_stride;            // this group's frame t
        const T* gb = gframe_t - (long)grp * K * frame_stride;
        // frame t - 1 - m: the state sub-step m linearises about / the gradient it injects (null: none)
        const T* hf[K];
        const T* gf[K];
#pragma unroll
        for (int m = 0; m < K; ++m) {
            hf[m] = hb - (long)(m + 1) * frame_stride;
            gf[m] = (gmask >> m) & 1u ? gb - (long)(m + 1) * frame_stride : (const T*)nullptr;
        }
        PI_PSTAMP(0);
        // ---- P0: the top of the pyramid -- needs my own tile only; the neighbours' granules are on their way ----
        persist_pass<T, K, BX, BY, NT, 0>(b0, b1, hf, gf, hf[1], gf[1], so1, g, ty0, tx0, P, acc_c, ops, ops2, mom, false, tab_geo + 0 * NT, jp, lacc);
        PI_PSTAMP(1);
        // ---- request the halo ring my neighbours published at the end of their previous group: the loads travel under P1.
        // (Requested before P0 they come back stale and a second round trip is exposed; requested by the four waves that idle in
        // P0 / P1 alone, with the publish moved under P0 as well, the hand-over takes those waves 2 us and P0 waits for them --
        // both measured, tools/persist_dev.hip, profiles/r04_persistent_split_timelines.txt.) ----
        const unsigned epoch = (unsigned)grp;
        const size_t half = (size_t)(epoch & 1u) * (size_t)ntiles * (2 * BANDU);          // granule index of this parity's half
        int gs[NGAT];
        typename GranuleIO<T>::Raw gx[NGAT];
        if (grp > 0) {
#if PI_ADJ_PERSIST_PAUSE
            __builtin_amdgcn_s_sleep(PI_ADJ_PERSIST_PAUSE);
#endif
#pragma unroll
            for (int q = 0; q < NGAT; ++q) {
                gs[q] = tab_gs[q * NT + (int)threadIdx.x];
                gx[q] = gio.get(half + (size_t)gs[q]);                                     // (lanes without: granule 0)
            }
        }
        // ---- P1 ----
        persist_pass<T, K, BX, BY, NT, 1>(b0, b1, hf, gf, up2 ? hf[0] : hf[2], up2 ? gf[0] : gf[2], so2, g, ty0, tx0, P, acc_c, ops2, ops, mom,
                                          false, tab_geo + 1 * NT, jp, lacc);
        PI_PSTAMP(2);
        if (grp > 0) {
            int gl[NGAT];
#pragma unroll
            for (int q = 0; q < NGAT; ++q) gl[q] = tab_gl[q * NT + (int)threadIdx.x];
            const unsigned long long t0 = wall_clock64();
            const unsigned long long bound = grp == 1 ? pa.first_timeout_ticks : pa.timeout_ticks;
            bool failed = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NGAT; ++q)
                    if (gl[q] >= 0) ok &= GranuleIO<T>::ok(gx[q], epoch);
                if (__all(ok)) break;
                // give up when the wait is over its bound (the first to do so raises the abort flag below) or when another
                // workgroup already has: a launch whose workgroups are not all resident ends within the bound
                if (wall_clock64() - t0 > bound ||
                    __hip_atomic_load(pa.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int q = 0; q < NGAT; ++q)
                    if (gl[q] >= 0 && !GranuleIO<T>::ok(gx[q], epoch)) gx[q] = gio.get(half + (size_t)gs[q]);
            }
            if (failed) {
                if (threadIdx.x % WAVE == 0) *wg_abort = 1;
            } else {
#pragma unroll
                for (int q = 0; q < NGAT; ++q)
                    if (gl[q] >= 0) GranuleIO<T>::land(gx[q], b0 + gl[q]);
            }
            lds_barrier();
        }
        if (grp > 0 && *wg_abort) {                        // ABORT: nothing this launch was asked for is written
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(pa.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(pa.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && pa.host) {
                    __hip_atomic_store(pa.host + 1, grp - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 2, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
        PI_PSTAMP(3);
        // ---- P2 .. P5: the rest of the pyramid next to the ring passes ----
        persist_pass<T, K, BX, BY, NT, 2>(b0, b1, hf, gf, hf[1], gf[1], so3, g, ty0, tx0, P, acc_c, ops, ops2, mom, up2, tab_geo + 2 * NT, jp, lacc);
        PI_PSTAMP(4);
        persist_pass<T, K, BX, BY, NT, 3>(b0, b1, hf, gf, hf[2], gf[2], so4, g, ty0, tx0, P, acc_c, ops2, ops, mom, false, tab_geo + 3 * NT, jp, lacc);
        PI_PSTAMP(5);
        persist_pass<T, K, BX, BY, NT, 4>(b0, b1, hf, gf, hf[3], gf[3], so5, g, ty0, tx0, P, acc_c, ops, ops2, mom, false, tab_geo + 4 * NT, jp, lacc);
        PI_PSTAMP(6);
        // (the operands the last pass requests belong to the next group's P0: frame t - K - 1)
        const unsigned gmask_next = last ? gmask : persist_mask<K>(pa, pa.t_top - K * (grp + 1));
        const T* hn = last ? hf[3] : hb - (long)(K + 1) * frame_stride;
        const T* gn = last ? gf[3] : (gmask_next & 1u ? gb - (long)(K + 1) * frame_stride : (const T*)nullptr);
        persist_pass<T, K, BX, BY, NT, 5>(b0, b1, hf, gf, hn, gn, last ? so5 : so0, g, ty0, tx0, P, acc_c, ops2, ops, mom, up5, tab_geo + 5 * NT, jp, lacc);
        PI_PSTAMP(7);
        // the float32 2-vector moment sums are folded into the lane's double sums in LDS every fourth group (see the unsplit kernel)
        // (float64: the sub-steps add straight into the LDS rows)
        if constexpr (sizeof(T) == 4) {
            if ((grp & 3) == 3 || last) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int m = 0; m < 10; ++m) {
                        double* slot = lacc + (10 * s + m) * NT + (int)threadIdx.x;
                        *slot += (double)mom_total(mom.a[s][m]);
                        mom.a[s][m] = V2<T>{T(0), T(0)};
                    }
            }
        }
        if (last) {
            // the result of the last group goes to memory: frame t - K of the adjoint trajectory, or dL/dh0 itself
            T* dst = g_h0 ? g_h0 : aframe_t - (long)(grp + 1) * K * frame_stride;
            tile_store<T, K, BX, BY, NT, true>(b0, dst, g, ty0, tx0);
            break;
        }
        gmask = gmask_next;
        // ---- publish my band (the border 2K points of the tile, complete since the barrier that ended P5) ----
        const unsigned ep1 = (unsigned)grp + 1u;
        const size_t mine = (size_t)(ep1 & 1u) * (size_t)ntiles * (2 * BANDU) + (size_t)tile * (2 * BANDU);
#pragma unroll
        for (int q = 0; q < NPUB; ++q) {
            const int pl = tab_pub[q * NT + (int)threadIdx.x];
            if (pl >= 0) gio.put(mine + (size_t)((int)threadIdx.x + q * NT), ep1, b0 + pl);
        }
        PI_PSTAMP(8);
        // (no barrier: the next pass, P0, reads b0 -- complete -- and writes b1's centre, which nobody reads any more)
    }

    // ---- once per rollout: diffusion-coefficient sums and the 20 moments of this tile -> its partial row -------------------
    lds_barrier();
    double* red = reinterpret_cast<double*>(smem_raw);     // state buffers are dead now: [2][NT / WAVE] doubles
    constexpr int NW = NT / WAVE;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double r = wave_sum_to_last(acc_c[s]);
        if (lane == REDUCE_LANE) red[s * NW + wave] = r;
    }
    lds_barrier();
    if (threadIdx.x < 2) {
        double sum = 0.0;
        for (int w = 0; w < NW; ++w) sum += red[(int)threadIdx.x * NW + w];
        partials[(long)blockIdx.x * np + P_COEF + (int)threadIdx.x] += sum;
    }
    if (threadIdx.x < 320) {
        const int mm = (int)threadIdx.x >> 4, part = (int)threadIdx.x & 15;
        const double* row = lacc + mm * LACC + part;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int k = 0; k < LACC; k += 64) {
            const double v0 = row[k], v1 = row[k + 16], v2 = row[k + 32], v3 = row[k + 48];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        double a = (a0 + a1) + (a2 + a3);
        a += dpp_mov<0x111, 0xF>(a);
        a += dpp_mov<0x112, 0xF>(a);
        a += dpp_mov<0x114, 0xF>(a);
        a += dpp_mov<0x118, 0xF>(a);
        if (part == 15) partials[(long)blockIdx.x * np + P_W + mm] += a;
    }
}


// ------------------------------------------------------------------------------------------------
// PERSISTENT FORWARD (round 4): the T-step forward rollout of a grid of <= #CUs 32 x 32 tiles as ONE launch of resident
// workgroups, on the machinery of pi_adj2d_persist_split_kernel -- state tile in LDS across groups of K steps, the 2K-wide border
// band published / the halo ring gathered as data-tagged granules once per group, the halo-independent pyramid I_m of each
// sub-step computed while the granules travel, strip geometry from the LDS table, residency roll call / bounded waits / abort.
// What it removes from every K = 4 steps of the launch-per-group kernel: the dependent-kernel boundary (1.9 us) and the cold
// window load (0.8 us) of a 6.65 us launch.  Every frame still goes to memory (the backward reads the trajectory): the waves
// that have no strip in a pass store a finished level from LDS while the others compute --
//     P0 I_0 (+ store level 4 of the previous group) | P1 I_1 (+ the centre of level 1, which I_2 overwrites in P2) |
//     ring -> LDS | P2 I_2 + A_0 | P3 A_1 (+ the rest of level 1) | P4 A_2 (+ level 2) | P5 I_3 + A_3 (+ level 3) | publish
// Levels alternate between the two LDS buffers exactly as in pi_fwd2d_tile_kernel; a strip is computed by the same lds_star4 /
// poly_r / update sequence: the trajectory is that kernel's bit for bit.
// ------------------------------------------------------------------------------------------------
// one strip of the forward sub-step, placed by a geometry word (persist_geo_word); pre-contracted block
template <typename T, int K, int BX, int BY>
__device__ __forceinline__ void fwd_strip_geo(const T* cur, T* nxt, const T* __restrict__ P, unsigned w)
{
    using TL = Tile<K, BX, BY>;
    if (__builtin_amdgcn_ballot_w64(((w >> 16) & 1u) != 0u) == 0ull) return;       // whole waves without a strip in this pass
    const int off = (int)(w & 0xFFFFu);
    const T dt = P[P_DT];
    T u[4], v[4], lap[2][4];
    lds_star4<T, TL::LX, +1>(cur + off, 0, 0, P, u, lap[0]);
    lds_star4<T, TL::LX, +1>(cur + TL::PLANE + off, 0, 0, P, v, lap[1]);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        T rr[4];
        const T* c = P + P_W + 10 * s;
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[i] = poly_r(c, u[i], v[i]);
        const T coef = P[P_COEF + s];
        T o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T lp = s == 0 ? lap[0][i] : lap[1][i];
            const T res = coef * lp + rr[i];
            const T inc = res * dt;
            o[i] = (s == 0 ? u[i] : v[i]) + inc;
        }
        lds_store4(nxt + s * TL::PLANE + off, o);
    }
}

// The same strip with ALL its LDS reads issued first (float32; round 6).  hipcc schedules the second species' stencil rows behind
// the first species' arithmetic and store -- three load -> wait -> compute phases in a strip that is alone on its SIMD; with the
// 14 reads of both species in flight at once the strip waits once.  Same operations in the same order per value as lds_star4 /
// fwd_strip_geo: bit-identical.
template <typename T, int LX>
struct StarRows {
    Pack<T, 2> l, r;         // x = -2, -1 | +4, +5 of the centre row
    Pack<T, 4> m, n[4];      // the centre row's own four points; rows -2, -1, +1, +2
    __device__ __forceinline__ void load(const T* c)
    {
        l = ld<T, 2>(c - 2); m = ld<T, 4>(c); r = ld<T, 2>(c + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) n[t] = ld<T, 4>(c + (t < 2 ? t - 2 : t - 1) * LX);
    }
    __device__ __forceinline__ void star(const T* __restrict__ P, T (&ctr)[4], T (&lap)[4]) const
    {
        const T win[8] = {l.v[0], l.v[1], m.v[0], m.v[1], m.v[2], m.v[3], r.v[0], r.v[1]};
#pragma unroll
        for (int i = 0; i < 4; ++i) { ctr[i] = win[2 + i]; lap[i] = P[P_C0] * win[2 + i]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const T w = P[P_TAPS + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) lap[i] = fma_(w, n[t].v[i], lap[i]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = t < 2 ? t - 2 : t - 1;
            const T w = P[P_TAPS + 4 + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) lap[i] = fma_(w, win[2 + i + k], lap[i]);
        }
    }
};

template <typename T, int K, int BX, int BY>
__device__ __forceinline__ void fwd_strip_geo_loads_first(const T* cur, T* nxt, const T* __restrict__ P, unsigned w)
{
    static_assert(sizeof(T) == 4, "float32 (lds_star4's 16-byte reads)");
    using TL = Tile<K, BX, BY>;
    if (__builtin_amdgcn_ballot_w64(((w >> 16) & 1u) != 0u) == 0ull) return;
    const int off = (int)(w & 0xFFFFu);
    StarRows<T, TL::LX> ru, rv;
    ru.load(cur + off);
    rv.load(cur + TL::PLANE + off);
    __builtin_amdgcn_sched_barrier(0);                     // nothing is scheduled across: every read is issued before the arithmetic
    const T dt = P[P_DT];
    T u[4], v[4], lap[2][4];
    ru.star(P, u, lap[0]);
    rv.star(P, v, lap[1]);
    T o[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        T rr[4];
        const T* c = P + P_W + 10 * s;
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[i] = poly_r(c, u[i], v[i]);
        const T coef = P[P_COEF + s];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T lp = s == 0 ? lap[0][i] : lap[1][i];
            const T res = coef * lp + rr[i];
            const T inc = res * dt;
            o[s][i] = (s == 0 ? u[i] : v[i]) + inc;
        }
    }
    lds_store4(nxt + off, o[0]);
    lds_store4(nxt + TL::PLANE + off, o[1]);
}

// the owned BX x BY region of a level buffer -> its frame, by the lanes [FIRST, NT).  WHICH 0: all of it; 1: the block of rows /
// 16-byte chunks that covers what I_2 overwrites (level 1 only); 2: everything but that block
template <typename T, int K, int BX, int BY, int NT, int FIRST, bool PADDED, int WHICH>
__device__ __forceinline__ void persist_fwd_store(const T* buf, T* __restrict__ dst, const TileGeom& g, int ty0, int tx0)
{
    using TL = Tile<K, BX, BY>;
    constexpr int VEC = vec_width<T>::value;                               // 16-byte chunks: 4 float32 / 2 float64
    constexpr int BXV = BX / VEC, N = 2 * BY * BXV, LANES = NT - FIRST;
    constexpr int SIDE = BX - 4 * (K - 1), O2 = (BX - SIDE) / 2;           // I_2's output square: side 20 at offset 6
    constexpr int R0 = O2, R1 = O2 + SIDE, C0 = O2 / VEC, C1 = (O2 + SIDE + VEC - 1) / VEC;
    static_assert(C0 * VEC >= 2 && C1 * VEC <= BX - 2 && R0 >= 2 && R1 <= BY - 2, "the early block lies inside I_0's output");
    if ((int)threadIdx.x < FIRST) return;
    for (int i = (int)threadIdx.x - FIRST; i < N; i += LANES) {
        const int s = i / (BY * BXV);
        const int r = i - s * (BY * BXV);
        const int y = r / BXV, c = r - y * BXV;
        const bool early = y >= R0 && y < R1 && c >= C0 && c < C1;
        if (WHICH == 1 && !early) continue;
        if (WHICH == 2 && early) continue;
        const T* src = buf + s * TL::PLANE + (2 * K + y) * TL::LX + 2 * K + c * VEC;
        Pack<T, VEC> p;
        if constexpr (PADDED && lds_pad0<T>::value != 0) {
            const Pack<T, 2> a = ld<T, 2>(src), b = ld<T, 2>(src + 2);
            p.v[0] = a.v[0]; p.v[1] = a.v[1]; p.v[2] = b.v[0]; p.v[3] = b.v[1];
        } else {
            p = ld<T, VEC>(src);
        }
        #if PI_FWD_PERSIST_WT
        st_frame_wt<T, VEC>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC, p);
#else
        *reinterpret_cast<Pack<T, VEC>*>(dst + s * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC) = p;
#endif
    }
}

#ifndef PI_FWD_PERSIST_REQ_AFTER
#define PI_FWD_PERSIST_REQ_AFTER 0      // the ring is requested after pass P<this> (0 or 1).  Measured (tools/fwd_dev.hip): after P0 --
#endif                                  // 0.76 us into the group -- 5.74 us per group; after P1 (1.6 us) 6.55; one launch per group 6.64
#ifndef PI_FWD_PERSIST_WT
#define PI_FWD_PERSIST_WT 0             // frame stores of the resident forward: 0 = plain (write-back) stores.  Nobody reads a frame from
#endif                                  // memory before the launch ends (the state lives in LDS, halos travel as granules), and write-through
                                        // stores compete with the latency-critical granule loads: 5.8 -> 5.45 us per group (tools/fwd_dev.hip)
// HALF STRIPS (round 6; float32 and float64).  The annulus passes A_0 .. A_3 of a group are the loop ring -> A_0 .. A_3 -> publish -> flight that
// sets the resident forward's pace, and each of them is ONE four-point strip per wave on four to six of the eight waves: 128
// instructions (0.19 us of VALU issue) that take 0.56 us because the wave waits for its own LDS round trips with nobody to
// overlap them (tools/ubench/strip_ubench.hip, profiles/r06_granule_pairs.txt).  Cut in two, a pass's strips occupy all eight
// waves -- both waves of a SIMD -- with half the dependent work each.  Same operations in the same order per point: bit-identical.
// float64, whose strips are twice the instructions, gains 12 % of its forward from it (with I_3 of pass P5 on half-strips too),
// float32 2.4 %.  PART_FULL / PART_PYR words: the small-tile resident forward's sub-steps and I_3.
// Geometry word of half-strip h (strip h / 2 of the pass's annulus, points 2 (h % 2) .. + 1): bits 0-15 LDS offset, 16 live.
template <int K, int BX, int BY, int M, int PART = PART_ANN>
__device__ __forceinline__ unsigned fwd_half_word(int h)
{
    using TL = Tile<K, BX, BY>;
    using SM = StripMap<K, BX, BY, M, PART>;
    constexpr int O = 2 * (M + 1);
    const bool live = h >= 0 && (h >> 1) < SM::N;
    int idx = live ? (h >> 1) : 0;
    int ry, rc;
    SM::locate(idx, ry, rc);
    const int ly = ry + O, lx = 4 * rc + O + 2 * (h & 1);
    return (unsigned)(ly * TL::LX + lx) | (live ? 1u << 16 : 0u);
}

template <typename T, int K, int BX, int BY>
__device__ __forceinline__ void fwd_half_strip_geo(const T* cur, T* nxt, const T* __restrict__ P, unsigned w)
{
    using TL = Tile<K, BX, BY>;
    constexpr int LX = TL::LX;
    if (__builtin_amdgcn_ballot_w64(((w >> 16) & 1u) != 0u) == 0ull) return;       // whole waves without a half-strip in this pass
    const int off = (int)(w & 0xFFFFu);
    const T dt = P[P_DT];
    T ctr[2][2], lap[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const T* c = cur + s * TL::PLANE + off;
        const Pack<T, 2> a = ld<T, 2>(c - 2), m = ld<T, 2>(c), b = ld<T, 2>(c + 2);
        const T win[6] = {a.v[0], a.v[1], m.v[0], m.v[1], b.v[0], b.v[1]};         // x = -2 .. +3 of the centre row
#pragma unroll
        for (int i = 0; i < 2; ++i) { ctr[s][i] = win[2 + i]; lap[s][i] = P[P_C0] * win[2 + i]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = t < 2 ? t - 2 : t - 1;
            const Pack<T, 2> n = ld<T, 2>(c + k * LX);
            const T wt = P[P_TAPS + t];
#pragma unroll
            for (int i = 0; i < 2; ++i) lap[s][i] = fma_(wt, n.v[i], lap[s][i]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = t < 2 ? t - 2 : t - 1;
            const T wt = P[P_TAPS + 4 + t];
#pragma unroll
            for (int i = 0; i < 2; ++i) lap[s][i] = fma_(wt, win[2 + i + k], lap[s][i]);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const T* cf = P + P_W + 10 * s;
        const T coef = P[P_COEF + s];
        Pack<T, 2> o;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const T rr = poly_r(cf, ctr[0][i], ctr[1][i]);
            const T res = coef * lap[s][i] + rr;
            const T inc = res * dt;
            o.v[i] = ctr[s][i] + inc;
        }
        st<T, 2>(nxt + s * TL::PLANE + off, o);
    }
}

// The same with the lane's chunks located ONCE per rollout (round 6): which chunks a lane stores never changes, but the loop
// above re-derived species / row / column, the early-block test and both addresses for every chunk of every level of every group
// -- ~40 VALU instructions per chunk on the SIMD that a computing wave shares.  Here: LDS element offset + frame element offset per
// chunk in registers, a store is a read, a 32-bit offset next to the frame's wave-uniform base, and the write.
template <typename T, int K, int BX, int BY, int NT, int FIRST>
struct FwdStoreMap {
    using TL = Tile<K, BX, BY>;
    static constexpr int VEC = vec_width<T>::value, BXV = BX / VEC, N = 2 * BY * BXV, LANES = NT - FIRST, CH = (N + LANES - 1) / LANES;
    unsigned lds[CH], glb[CH];
    unsigned flags;                                        // bit j: chunk j exists; bit 8 + j: it lies in the early block
    __device__ __forceinline__ void init(const TileGeom& g, int ty0, int tx0)
    {
        constexpr int SIDE = BX - 4 * (K - 1), O2 = (BX - SIDE) / 2;
        constexpr int R0 = O2, R1 = O2 + SIDE, C0 = O2 / VEC, C1 = (O2 + SIDE + VEC - 1) / VEC;
        flags = 0u;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int i = (int)threadIdx.x - FIRST + j * LANES;
            const bool have = (int)threadIdx.x >= FIRST && i < N;
            const int ii = have ? i : 0;
            const int sp = ii / (BY * BXV), r = ii - sp * (BY * BXV), y = r / BXV, c = r - y * BXV;
            const bool early = y >= R0 && y < R1 && c >= C0 && c < C1;
            lds[j] = (unsigned)(sp * TL::PLANE + (2 * K + y) * TL::LX + 2 * K + c * VEC);
            glb[j] = (unsigned)((long)sp * g.ss + (long)(ty0 + y) * g.W + tx0 + c * VEC);
            flags |= (have ? 1u : 0u) << j | (early ? 1u : 0u) << (8 + j);
            asm volatile("" : "+v"(lds[j]), "+v"(glb[j]));
        }
        asm volatile("" : "+v"(flags));
    }
    template <bool PADDED, int WHICH>
    __device__ __forceinline__ void store(const T* buf, T* __restrict__ dst) const
    {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool early = (flags >> (8 + j)) & 1u;
            if (!((flags >> j) & 1u) || (WHICH == 1 && !early) || (WHICH == 2 && early)) continue;
            const T* src = buf + lds[j];
            Pack<T, VEC> p;
            if constexpr (PADDED && lds_pad0<T>::value != 0) {
                const Pack<T, 2> a = ld<T, 2>(src), b = ld<T, 2>(src + 2);
                p.v[0] = a.v[0]; p.v[1] = a.v[1]; p.v[2] = b.v[0]; p.v[3] = b.v[1];
            } else {
                p = ld<T, VEC>(src);
            }
#if PI_FWD_PERSIST_WT
            st_frame_wt<T, VEC>(dst + glb[j], p);
#else
            *reinterpret_cast<Pack<T, VEC>*>(dst + glb[j]) = p;
#endif
        }
    }
};

#ifndef PI_FWD_I2_EARLY
#define PI_FWD_I2_EARLY 0               // resident forward: the pyramid's third level under the ring's flight instead of next to A_0
                                        // (round 6: bit-identical, 1.26 -> 1.27-1.32 us per step: the extra barrier costs what P2 gains).  Off.
#endif
#ifndef PI_FWD_HALF_STRIPS_F64
#define PI_FWD_HALF_STRIPS_F64 1        // ... of the float64 forward (a strip is twice the instructions there)
#endif
#ifndef PI_FWD_HALF_I3
#define PI_FWD_HALF_I3 1                // half-strip builds: I_3 (P5) on half-strips too, on waves 0 and 7, instead of whole strips on wave 0
#endif
#ifndef PI_FWD_HALF_PYR_F64
#define PI_FWD_HALF_PYR_F64 0           // ... and its pyramid passes P0 / P1 (I_0 on seven waves instead of four, I_1 on five instead of three):
                                        // bit-identical, lambda-omega forward 1.807 -> 1.838 us per step (they run under the ring's flight).  Off.
#endif
#ifndef PI_FWD_HALF_PYR
#define PI_FWD_HALF_PYR 0                   // (float32: 1.239 -> 1.236 us per step, nothing)
#endif
#ifndef PI_FWD_HALF_STRIPS
#define PI_FWD_HALF_STRIPS 1            // float32 resident forward: the annulus passes on two-point half-strips (all eight waves).
                                        // Measured (round 6): bit-identical, 1.26 -> 1.24 us per step -- a half-strip takes as long as a strip
                                        // (0.52-0.56 us: the pass is LDS round trips, not arithmetic); again after the small-tile work:
                                        // 1.27 -> 1.258; with I_3 on half-strips as well (PI_FWD_HALF_I3) 1.27 -> 1.239, headline 325.9 ->
                                        // 328.4 k: ON since then (float64, twice the arithmetic per strip, gains 12 % from the same passes).
#endif
#ifndef PI_FWD_HOLD_P
#define PI_FWD_HOLD_P 1                 // float32 resident forward: parameter block (1: vector, 2: scalar registers), strip geometry and
                                        // frame-store map held in registers for the whole rollout; 0: round 5's body
#endif
#ifndef PI_FWD_LOADS_FIRST
#define PI_FWD_LOADS_FIRST 0            // 1: fwd_strip_geo_loads_first (float32 only)
#endif
#ifndef PI_FWD_PERSIST_PAUSE
#define PI_FWD_PERSIST_PAUSE 0          // s_sleep units before the request
#endif

template <typename T, int K, int BX, int BY, int NT, int HOLDP = PI_FWD_HOLD_P>
__global__ void __launch_bounds__(NT)
pi_fwd2d_persist_kernel(T* __restrict__ frames /* frame t0; t0+1 .. t0 + K * ngroups are written */, long frame_stride,
                        const T* __restrict__ P_in, TileGeom g, PersistArgs pa)
{
    static_assert(BX == BY && K == 4 && BX == 32 && NT == 512, "32 x 32 tiles, four sub-steps, 8 waves");
    using TL = Tile<K, BX, BY>;
    using HO = HandOver<K, BX, GranuleIO<T>::VALS, NT>;                 // band / ring numbering in granules
    constexpr int BANDU = HO::BANDU, NPUB = HO::NPUB, NGAT = HO::NGAT;
    static_assert(NPUB + 2 * NGAT <= 13, "tables fit the LDS the host reserves");
    constexpr int IDLE = 4 * WAVE;                                      // waves 4..7 own no strip in P0, P1, P3, P4, P5
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int tyi = tile / g.tiles_x, txi = tile % g.tiles_x, tiles_y = g.H / BY;
    const int ty0 = tyi * BY, tx0 = txi * BX;
    const int ntiles = g.tiles_x * tiles_y;
    const GranuleIO<T> gio(pa.outbox, (size_t)2 * (size_t)ntiles * (2 * BANDU) * GranuleIO<T>::BYTES);

    // LDS: state buffers | int tables (publish, gather, geometry) | abort word
    int* tab_pub = reinterpret_cast<int*>(smem_raw + tile_state_bytes<T, K, BX, BY>());     // [NPUB][NT]: LDS position of a border value
    int* tab_gl = tab_pub + NPUB * NT;                                      // [NGAT][NT]: LDS position of a halo value
    int* tab_gs = tab_gl + NGAT * NT;                                       // [NGAT][NT]: granule index inside a parity half
    unsigned* tab_geo = reinterpret_cast<unsigned*>(tab_pub + 13 * NT);     // [6][NT]: the lane's strip in each pass
    int* wg_abort = tab_pub + PERSIST_SPLIT_TABLE_ROWS * NT;
    if (threadIdx.x == 0) {                                                 // residency roll call (see pi_adj2d_persist_kernel)
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(pa.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)ntiles && pa.host) __hip_atomic_store(pa.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    tab_geo[0 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 0>(g, ty0, tx0);
    tab_geo[1 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 1>(g, ty0, tx0);
    tab_geo[2 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 2>(g, ty0, tx0);
    tab_geo[3 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 3>(g, ty0, tx0);
    tab_geo[4 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 4>(g, ty0, tx0);
    tab_geo[5 * NT + (int)threadIdx.x] = persist_pass_geo<K, BX, BY, NT, 5>(g, ty0, tx0);
#pragma unroll
    for (int q = 0; q < NPUB; ++q) tab_pub[q * NT + (int)threadIdx.x] = HO::pub_pos((int)threadIdx.x + q * NT);
#pragma unroll
    for (int q = 0; q < NGAT; ++q) {
        int gl, gs;
        HO::gat_pos((int)threadIdx.x + q * NT, g, ty0, tx0, tiles_y, gl, gs);
        tab_gl[q * NT + (int)threadIdx.x] = gl;
        tab_gs[q * NT + (int)threadIdx.x] = gs;
    }
    // group 0 starts from frame t0 in memory (window = tile + ring), like a launch of pi_fwd2d_tile_kernel
    tile_load<T, K, BX, BY, NT>(frames, g, ty0, tx0, b0);
    __syncthreads();
    const int tid = (int)threadIdx.x;
    // Round 6: what never changes during a rollout is held in registers instead of being re-derived / re-read in every pass.
    //  * the 36 entries of the parameter block: behind the barriers' memory clobbers the compiler re-read them with scalar loads in
    //    EVERY pass -- after the strip's table word had arrived, and once more in the middle of the strip: two scalar-memory
    //    round trips on the critical path of a 154-instruction strip;
    //  * the lane's six strip-geometry words (the sweep, at 251 registers, keeps its LDS table; the forward uses 78 of 256);
    //  * where the lane's frame chunks lie (FwdStoreMap).
    // 512^2 x 1000, tools/fwd_dev.hip, same box: 1.335 -> 1.283 (block) -> 1.260 (+ geometry) -> 1.248 us per step (+ store map);
    // trajectory bit-identical.  Reading all of a strip's LDS rows before its arithmetic (fwd_strip_geo_loads_first) measured no
    // gain on top (1.26 either way) and is not used.
    // HOLDP 1: the block in vector registers (142 registers: one workgroup per CU); 2: in scalar registers (83 vector registers: the
    // two-workgroups-per-CU mode of option fwd_persist_per_cu still fits; 1.29 instead of 1.26 us per step); 0: round 5's body
    constexpr bool HOLD = HOLDP != 0;                       // (float64: 101 -> ~190 of 256 registers)
    T Ph[HOLD ? NPOLY : 1];
    constexpr bool HALF = HOLD && ((PI_FWD_HALF_STRIPS != 0 && sizeof(T) == 4) || (PI_FWD_HALF_STRIPS_F64 != 0 && sizeof(T) == 8));
    constexpr bool HALF_PYR = HALF && ((PI_FWD_HALF_PYR != 0 && sizeof(T) == 4) || (PI_FWD_HALF_PYR_F64 != 0 && sizeof(T) == 8));
    unsigned gw[HOLD ? 7 : 1];
    FwdStoreMap<T, K, BX, BY, NT, IDLE> smap;
    FwdStoreMap<T, K, BX, BY, NT, 0> smap_all;              // (half-strip passes: every wave computes, every lane stores one chunk)
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (HOLD) {
#pragma unroll
        for (int i = 0; i < NPOLY; ++i) {
            T x = P_in[i];
            if constexpr (HOLDP == 2) asm volatile("" : "+s"(x));
            else asm volatile("" : "+v"(x));
            Ph[i] = x;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) gw[i] = tab_geo[i * NT + tid];
        gw[6] = 0u;
        if constexpr (HALF) {
            // P2: I_2 on waves 0-1 (four-point strips), A_0 = 576 half-strips on the 384 lanes of waves 2-7 in two rounds (gw[2], gw[6]);
            // P3: A_1 = 512 half-strips; P4: A_2 = 448; P5: I_3 on wave 0, A_3 = 384 half-strips on waves 1-6
            if (tid >= 128) { gw[2] = fwd_half_word<K, BX, BY, 0>(tid - 128); gw[6] = fwd_half_word<K, BX, BY, 0>(tid - 128 + 384); }
            gw[3] = fwd_half_word<K, BX, BY, 1>(tid);
            gw[4] = fwd_half_word<K, BX, BY, 2>(tid);
            if (tid >= 64) gw[5] = fwd_half_word<K, BX, BY, 3>(tid - 64);
            if constexpr (PI_FWD_HALF_I3 != 0) {            // I_3 = 128 half-strips on wave 0 and on wave 7 (idle in P5 otherwise)
                if (tid < 64) gw[5] = fwd_half_word<K, BX, BY, 3, PART_PYR>(tid);
                if (tid >= 448) gw[5] = fwd_half_word<K, BX, BY, 3, PART_PYR>(tid - 448 + 64);
            }
            if constexpr (HALF_PYR) {
                gw[0] = fwd_half_word<K, BX, BY, 0, PART_PYR>(tid);
                gw[1] = fwd_half_word<K, BX, BY, 1, PART_PYR>(tid);
            }
            smap_all.init(g, ty0, tx0);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(gw[i]));
        smap.init(g, ty0, tx0);
    }
    const T* P = HOLD ? Ph : P_in;
    auto geo = [&](int i) -> unsigned { if constexpr (HOLD) return gw[i]; else return tab_geo[i * NT + tid]; };
#define PI_FWD_GEO(i) geo(i)
#define PI_FWD_STORE(PADDED, WHICH, buf, dst)                                                                      \
    do {                                                                                                           \
        if constexpr (HOLD) smap.template store<PADDED, WHICH>(buf, dst);                                          \
        else persist_fwd_store<T, K, BX, BY, NT, IDLE, PADDED, WHICH>(buf, dst, g, ty0, tx0);                       \
    } while (0)
    auto PI_FWD_STRIP = [&](const T* cur, T* nxt, const T* pp, unsigned w) {
        if constexpr (PI_FWD_LOADS_FIRST != 0 && sizeof(T) == 4) fwd_strip_geo_loads_first<T, K, BX, BY>(cur, nxt, pp, w);
        else fwd_strip_geo<T, K, BX, BY>(cur, nxt, pp, w);
    };
    auto half_strip = [&](const T* cur, T* nxt, unsigned w) {
        if constexpr (HALF) fwd_half_strip_geo<T, K, BX, BY>(cur, nxt, P, w);
    };
    for (int grp = 0; grp < pa.ngroups; ++grp) {
        T* fr = frames + (long)grp * K * frame_stride;                    // this group's frame t: fr + m * frame_stride = level m
        PI_PSTAMP(0);
        // ---- P0: I_0 (b0 -> b1); the idle waves store level 4 of the previous group (= this group's level 0, complete in b0) ----
        if constexpr (HALF_PYR) {
            if (grp > 0) smap_all.template store<true, 0>(b0, fr);
            half_strip(b0, b1, gw[0]);
        } else {
            if (grp > 0) PI_FWD_STORE(true, 0, b0, fr);
            PI_FWD_STRIP(b0, b1, P, PI_FWD_GEO(0));
        }
        lds_barrier();
        PI_PSTAMP(1);
        const unsigned epoch = (unsigned)grp;
        const size_t half = (size_t)(epoch & 1u) * (size_t)ntiles * (2 * BANDU);          // granule index of this parity's half
        int gs[NGAT];
        typename GranuleIO<T>::Raw gx[NGAT];
        auto request = [&]() {
            if (grp > 0) {
#if PI_FWD_PERSIST_PAUSE
                __builtin_amdgcn_s_sleep(PI_FWD_PERSIST_PAUSE);
#endif
#pragma unroll
                for (int q = 0; q < NGAT; ++q) {
                    gs[q] = tab_gs[q * NT + tid];
                    gx[q] = gio.get(half + (size_t)gs[q]);                                 // (lanes without: granule 0)
                }
            }
        };
        if constexpr (PI_FWD_PERSIST_REQ_AFTER == 0) request();
        // ---- P1: I_1 (b1 -> b0 centre); the idle waves store the block of level 1 that I_2 will overwrite ----
        if constexpr (HALF_PYR) {
            smap_all.template store<false, 1>(b1, fr + frame_stride);
            half_strip(b1, b0, gw[1]);
        } else {
            PI_FWD_STORE(false, 1, b1, fr + frame_stride);
            PI_FWD_STRIP(b1, b0, P, PI_FWD_GEO(1));
        }
        PI_PSTAMP(2);
        if constexpr (PI_FWD_PERSIST_REQ_AFTER == 1) request();
        if constexpr (PI_FWD_I2_EARLY != 0) {
            // I_2 (waves 0-1; it needs I_1 only) while the ring is still in flight, instead of next to A_0 behind it: P2 is then the
            // annulus alone.  One more barrier; b1's centre is free (its early block of level 1 was stored during P1).
            lds_barrier();
            if (wave_id < 2) PI_FWD_STRIP(b0, b1, P, PI_FWD_GEO(2));
        }
        if (grp > 0) {
            int gl[NGAT];
#pragma unroll
            for (int q = 0; q < NGAT; ++q) gl[q] = tab_gl[q * NT + tid];
            const unsigned long long t0 = wall_clock64();
            const unsigned long long bound = grp == 1 ? pa.first_timeout_ticks : pa.timeout_ticks;
            bool failed = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NGAT; ++q)
                    if (gl[q] >= 0) ok &= GranuleIO<T>::ok(gx[q], epoch);
                if (__all(ok)) break;
                if (wall_clock64() - t0 > bound ||
                    __hip_atomic_load(pa.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int q = 0; q < NGAT; ++q)
                    if (gl[q] >= 0 && !GranuleIO<T>::ok(gx[q], epoch)) gx[q] = gio.get(half + (size_t)gs[q]);
            }
            if (failed) {
                if (threadIdx.x % WAVE == 0) *wg_abort = 1;
            } else {
                // (the ring of b0 -- level 0 -- is read by A_0 only; I_1 above wrote b0's centre)
#pragma unroll
                for (int q = 0; q < NGAT; ++q)
                    if (gl[q] >= 0) GranuleIO<T>::land(gx[q], b0 + gl[q]);
            }
        }
        lds_barrier();
        if (grp > 0 && *wg_abort) {                        // ABORT: the host re-runs the rollout with one launch per group
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(pa.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(pa.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && pa.host) {
                    __hip_atomic_store(pa.host + 1, grp - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 2, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
        PI_PSTAMP(3);
        // ---- P2: I_2 (b0 centre -> b1 centre) next to A_0 (b0 with its ring -> b1 outside I_0's square) ----
        if constexpr (HALF) {
            if (wave_id < 2) PI_FWD_STRIP(b0, b1, P, gw[2]);
            else { half_strip(b0, b1, gw[2]); half_strip(b0, b1, gw[6]); }
        } else if constexpr (PI_FWD_I2_EARLY != 0) {
            if (wave_id >= 2) PI_FWD_STRIP(b0, b1, P, PI_FWD_GEO(2));
        } else {
            PI_FWD_STRIP(b0, b1, P, PI_FWD_GEO(2));
        }
        lds_barrier();
        PI_PSTAMP(4);
        // ---- P3: A_1 (b1 -> b0); level 1 is complete in b1 outside the block stored in P1 ----
        if constexpr (HALF) smap_all.template store<false, 2>(b1, fr + frame_stride);
        else PI_FWD_STORE(false, 2, b1, fr + frame_stride);
        PI_PSTAMP(9);                                      // (debug builds: what a pass is made of -- tools/fwd_dev.hip)
        if constexpr (HALF) half_strip(b1, b0, gw[3]);
        else PI_FWD_STRIP(b1, b0, P, PI_FWD_GEO(3));
        PI_PSTAMP(10);
        lds_barrier();
        PI_PSTAMP(5);
        // ---- P4: A_2 (b0 -> b1); level 2 is complete in b0 ----
        if constexpr (HALF) { smap_all.template store<true, 0>(b0, fr + 2 * frame_stride); half_strip(b0, b1, gw[4]); }
        else { PI_FWD_STORE(true, 0, b0, fr + 2 * frame_stride); PI_FWD_STRIP(b0, b1, P, PI_FWD_GEO(4)); }
        lds_barrier();
        PI_PSTAMP(6);
        // ---- P5: I_3 + A_3 (b1 -> b0); level 3 is complete in b1 ----
        if constexpr (HALF) {
            smap_all.template store<false, 0>(b1, fr + 3 * frame_stride);
            if (PI_FWD_HALF_I3 == 0 && wave_id < 1) PI_FWD_STRIP(b1, b0, P, gw[5]);
            else half_strip(b1, b0, gw[5]);
        } else {
            PI_FWD_STORE(false, 0, b1, fr + 3 * frame_stride);
            PI_FWD_STRIP(b1, b0, P, PI_FWD_GEO(5));
        }
        lds_barrier();
        PI_PSTAMP(7);
        if (grp + 1 == pa.ngroups) {                       // the last level 4: stored by everybody
            tile_store<T, K, BX, BY, NT, true>(b0, fr + (long)K * frame_stride, g, ty0, tx0);
            break;
        }
        // ---- publish my band of level 4 (complete since the barrier) ----
        const unsigned ep1 = (unsigned)grp + 1u;
        const size_t mine = (size_t)(ep1 & 1u) * (size_t)ntiles * (2 * BANDU) + (size_t)tile * (2 * BANDU);
#pragma unroll
        for (int q = 0; q < NPUB; ++q) {
            const int pl = tab_pub[q * NT + tid];
            if (pl >= 0) gio.put(mine + (size_t)(tid + q * NT), ep1, b0 + pl);
        }
        PI_PSTAMP(8);
        // (no barrier: P0 reads b0 -- complete -- and writes b1 inside I_0's square; the store of level 3 from b1 was issued before
        // the barrier that ended P5 -- its LDS reads are done)
    }
}

// ------------------------------------------------------------------------------------------------
#undef PI_FWD_GEO
#undef PI_FWD_STORE

// PERSISTENT FORWARD for the SMALL-TILE regime and ragged grids (round 5): grids that are not whole 32 x 32
// tiles or have fewer than 16 of them -- the reference's own 100^2 x 200 rollout (train_2drd.py:162-190, :597-636) among them --
// paid one launch per four steps: 5.0 us for sub-steps that take under two.  Here the T-step rollout is ONE launch of resident
// workgroups on the machinery of pi_adj2d_persist_small_kernel: the state tile stays in LDS across groups of K steps, a tile
// publishes ALL its BX x BY values of the frame a group ends on as data-tagged granules, the 2K-wide ring is gathered through
// tables built from GLOBAL coordinates (owner tile after the periodic wrap -- ragged edge tiles, halos that span two neighbours and
// single-column grids are the same code), residency roll call / bounded waits / clean abort as everywhere.  The sub-steps are
// pi_fwd2d_tile_kernel's own device functions (fwd_substeps): the trajectory is that kernel's bit for bit.  The frame a group ends
// on is stored AFTER the tile has been published and the ring requested: the neighbours wait for the granules, nobody for it.
// ------------------------------------------------------------------------------------------------
// HALFS (round 6): half-strips on twice the lanes -- both waves of a SIMD work in every sub-step (7 | 5 | 4 | 2 waves of a 32 x 8 tile
// instead of 4 | 3 | 2 | 1), half the granule requests per lane in the hand-over.  Same operations per point: bit-identical.
template <typename T, int K, int BX, int BY, int NT, bool HALFS = false>
__global__ void __launch_bounds__(NT)
pi_fwd2d_persist_small_kernel(T* __restrict__ frames /* frame t0; t0+1 .. t0 + K * ngroups are written */, long frame_stride,
                              const T* __restrict__ P_in, TileGeom g, PersistArgs pa)
{
    static_assert(sizeof(T) == 4 && K % 2 == 0, "float32; an even number of sub-steps leaves the state in buffer 0");
    using TL = Tile<K, BX, BY>;
    constexpr int HW = 2 * K, LXW = TL::LX, LYW = TL::LY;
    constexpr int OWN = BX * BY;                                         // values per species a tile publishes
    constexpr int RINGH = LXW * LYW - OWN;                               // halo values per species
    constexpr bool PAIRS = HALFS && GranuleIO<T>::VALS == 2;            // (as in pi_adj2d_persist_small_kernel)
    constexpr int GV = PAIRS ? 2 : 1;
    static_assert(OWN % (2 * GV) == 0 && RINGH % GV == 0 && BX % GV == 0 && HW % GV == 0, "granules of x-adjacent values");
    constexpr int NPUB = (2 * OWN / GV + NT - 1) / NT, NGAT = (2 * RINGH / GV + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw) + lds_pad0<T>::value;
    T* b1 = reinterpret_cast<T*>(smem_raw) + 2 * TL::PLANE + lds_pad1<T>::value;
    const int tile = tile_of_block(blockIdx.x, g);
    const int tiles_y = (g.H + BY - 1) / BY;
    const int tyi = tile / g.tiles_x, txi = tile % g.tiles_x;
    const int ty0 = tyi * BY, tx0 = txi * BX;
    const int ntiles = g.tiles_x * tiles_y;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* outbox = (gu64*)pa.outbox;
    const GranuleIO<T> gio(pa.outbox, (size_t)2 * (size_t)ntiles * (2 * OWN) * 8);             // (PAIRS; same bytes either way)
    // LDS: state buffers | gather tables | abort word
    int* tab_gl = reinterpret_cast<int*>(smem_raw + tile_state_bytes<T, K, BX, BY>());      // [NGAT][NT]: LDS position of a halo value
    int* tab_gs = tab_gl + NGAT * NT;                                                       // [NGAT][NT]: granule index inside a parity half
    unsigned* tab_geo = reinterpret_cast<unsigned*>(tab_gs + NGAT * NT);                    // [K][NT]: the lane's strip in each sub-step
    int* wg_abort = reinterpret_cast<int*>(tab_geo + K * NT);
    // Measured and not adopted (profiles/r05_small_tile_resident_forward.txt): strip geometry from the table (GEO) together with
    // plain instead of write-through frame stores -- what pays in the 32 x 32 resident forward -- 100^2 1.075 -> 1.096 us per
    // step, 256^2 1.137 -> 1.165: these sub-steps are a single wave per SIMD with one strip, the division by a constant they save
    // is cheaper than the table read, and write-back lines of the frames compete with the granules later
    constexpr bool GEO = false, WT = true;
    static_assert(K == 4, "geometry rows below");
    if constexpr (GEO) {
        tab_geo[0 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 0, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[1 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 1, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[2 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 2, PART_FULL, 0>(g, ty0, tx0);
        tab_geo[3 * NT + (int)threadIdx.x] = persist_geo_word<K, BX, BY, NT, 3, PART_FULL, 0>(g, ty0, tx0);
    }
    if (threadIdx.x == 0) {                                                                 // residency roll call (pi_adj2d_persist_kernel)
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(pa.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)ntiles && pa.host) __hip_atomic_store(pa.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int q = 0; q < NGAT; ++q) {                       // (the tables of pi_adj2d_persist_small_kernel)
        const int r = (int)threadIdx.x + q * NT;
        int gl = -1, gs = 0;
        if (r < 2 * RINGH / GV) {
            const int sp = r / (RINGH / GV), e = (r - sp * (RINGH / GV)) * GV;       // (e: first value of the granule)
            int wy, wx;                                    // ring positions row-major over the window, skipping the centre
            if (e < HW * LXW) { wy = e / LXW; wx = e - wy * LXW; }
            else if (e < HW * LXW + BY * 2 * HW) { const int m = e - HW * LXW; wy = HW + m / (2 * HW); const int c = m % (2 * HW); wx = c < HW ? c : BX + c; }
            else { const int m = e - HW * LXW - BY * 2 * HW; wy = HW + BY + m / LXW; wx = m % LXW; }
            int gy = (ty0 + wy - HW) % g.H, gx = (tx0 + wx - HW) % g.W;                     // the global point, periodic
            gy += gy < 0 ? g.H : 0;
            gx += gx < 0 ? g.W : 0;
            const int nty = gy / BY, ntx = gx / BX;
            gl = sp * TL::PLANE + wy * LXW + wx;
            gs = ((nty * g.tiles_x + ntx) * (2 * OWN) + sp * OWN + (gy - nty * BY) * BX + (gx - ntx * BX)) / GV;
        }
        tab_gl[q * NT + (int)threadIdx.x] = gl;
        tab_gs[q * NT + (int)threadIdx.x] = gs;
    }
    // group 0 starts from frame t0 in memory (window = tile + ring), like a launch of pi_fwd2d_tile_kernel
    tile_load<T, K, BX, BY, NT>(frames, g, ty0, tx0, b0);
    __syncthreads();
    const int tid = (int)threadIdx.x;
    // (round 6, as in pi_fwd2d_persist_kernel: the parameter block held in registers -- behind the barriers' memory clobbers every
    // sub-step re-read it with scalar loads on its critical path; 85 -> ~125 of this kernel's 256 registers)
    T Ph[NPOLY];
#pragma unroll
    for (int i = 0; i < NPOLY; ++i) {
        T x = P_in[i];
        asm volatile("" : "+v"(x));
        Ph[i] = x;
    }
    const T* P = Ph;
    unsigned gw[K] = {0u, 0u, 0u, 0u};                                     // HALFS: the lane's half-strip in each sub-step
    if constexpr (HALFS) {
        gw[0] = fwd_half_word<K, BX, BY, 0, PART_FULL>(tid);
        gw[1] = fwd_half_word<K, BX, BY, 1, PART_FULL>(tid);
        gw[2] = fwd_half_word<K, BX, BY, 2, PART_FULL>(tid);
        gw[3] = fwd_half_word<K, BX, BY, 3, PART_FULL>(tid);
#pragma unroll
        for (int m = 0; m < K; ++m) asm volatile("" : "+v"(gw[m]));
    }
    for (int grp = 0; grp < pa.ngroups; ++grp) {
        T* fr = frames + (long)grp * K * frame_stride;                    // this group's frame t
        const bool last = grp + 1 == pa.ngroups;
        // frames t + 1 .. t + K - 1 go to memory as in the launch-per-group kernel (idle-wave stores included); frame t + K is
        // complete in buffer 0 after the barrier that ends sub-step K - 1
        fwd_substeps<T, POLY, K, BX, BY, NT, 0, false, GEO, WT, HALFS>(b0, b1, fr, frame_stride, g, ty0, tx0, P, HALFS ? gw : tab_geo);
        if (last) {
            tile_store<T, K, BX, BY, NT, true, WT>(b0, fr + (long)K * frame_stride, g, ty0, tx0);
            break;
        }
        // ---- hand-over: publish my tile, request my ring, THEN store the frame ----
        const unsigned epoch = (unsigned)grp + 1u;
        gu64* half = outbox + (size_t)(epoch & 1u) * (size_t)ntiles * (2 * OWN);
        gu64* mine = half + (size_t)tile * (2 * OWN);
        const size_t halfu = (size_t)(epoch & 1u) * (size_t)ntiles * (2 * OWN / GV);      // ... in granules (PAIRS)
#pragma unroll
        for (int q = 0; q < NPUB; ++q) {
            const int i = tid + q * NT;
            if (i < 2 * OWN / GV) {
                const int sp = i / (OWN / GV), e = (i - sp * (OWN / GV)) * GV, y = e / BX, x = e - y * BX;
                const T* src = b0 + sp * TL::PLANE + (HW + y) * LXW + HW + x;
                if constexpr (PAIRS) {
                    gio.put(halfu + (size_t)tile * (2 * OWN / GV) + (size_t)i, epoch, src);
                } else {
                    const unsigned v = __builtin_bit_cast(unsigned, *src);
                    __hip_atomic_store(mine + i, ((unsigned long long)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        tile_store<T, K, BX, BY, NT, true, WT>(b0, fr + (long)K * frame_stride, g, ty0, tx0);
        for (int w = 0; w < pa.pause; ++w) __builtin_amdgcn_s_sleep(1);
        int gl[NGAT], gs[NGAT];
        typename SmallGranule<T, PAIRS>::Raw gx[NGAT];
        auto fetch = [&](int idx) {
            if constexpr (PAIRS) return gio.get(halfu + (size_t)idx);
            else return __hip_atomic_load(half + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
#pragma unroll
        for (int q = 0; q < NGAT; ++q) {
            gl[q] = tab_gl[q * NT + tid];
            gs[q] = tab_gs[q * NT + tid];
            gx[q] = fetch(gs[q]);                                                                     // (lanes without: granule 0)
        }
        const unsigned long long t0 = wall_clock64();
        const unsigned long long bound = grp == 0 ? pa.first_timeout_ticks : pa.timeout_ticks;
        bool failed = false;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < NGAT; ++q)
                if (gl[q] >= 0) ok &= SmallGranule<T, PAIRS>::ok(gx[q], epoch);
            if (__all(ok)) break;
            if (wall_clock64() - t0 > bound ||
                __hip_atomic_load(pa.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int q = 0; q < NGAT; ++q)                      // only what has not arrived yet is asked for again
                if (gl[q] >= 0 && !SmallGranule<T, PAIRS>::ok(gx[q], epoch)) gx[q] = fetch(gs[q]);
        }
        if (failed) {
            if (threadIdx.x % WAVE == 0) *wg_abort = 1;
        } else {
#pragma unroll
            for (int q = 0; q < NGAT; ++q)
                if (gl[q] >= 0) SmallGranule<T, PAIRS>::land(gx[q], b0 + gl[q]);
        }
        lds_barrier();
        if (*wg_abort) {
            // ABORT: frames written so far are the launch-per-group kernel's values, but the launch reports failure and the host
            // recomputes the rollout launch by launch (deterministic)
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(pa.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(pa.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && pa.host) {
                    __hip_atomic_store(pa.host + 1, grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 2, tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(pa.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
}

}  // namespace pi
