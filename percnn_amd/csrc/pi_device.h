// pi_device.h -- device-side helpers shared by the Pi-block kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pi {

// ---- parameter block layout (documented in include/percnn_pi.h) ------------------------------
constexpr int P_DT = 0, P_COEF = 1, P_C0 = 3, P_TAPS = 4, P_W = 16;
__host__ __device__ constexpr int species_block(int hc) { return 10 * hc + 1; }
// hc == 0 selects the pre-contracted polynomial reaction: 16 header slots + 2 x 10 cubic coefficients
constexpr int POLY = -1;          // template tag for that mode
constexpr int NPOLY = P_W + 20;
// hc == -1 selects the advective polynomial block of the Stage-3 physics-based cells (60 entries, pi_adv.h)
__host__ __device__ constexpr int nparams(int hc) { return hc == 0 ? NPOLY : (hc == -1 ? 60 : P_W + 2 * species_block(hc)); }

constexpr int WAVE = 64;          // CDNA wavefront
constexpr int NXCD = 8;           // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

// 16-byte vector access: 4 x f32 or 2 x f64 per lane -> 1 KiB per wave-instruction
template <typename T> struct vec_width;
template <> struct vec_width<float>  { static constexpr int value = 4; };
template <> struct vec_width<double> { static constexpr int value = 2; };

template <typename T, int N> struct alignas(sizeof(T) * N) Pack { T v[N]; };

template <typename T, int N>
__device__ __forceinline__ Pack<T, N> ld(const T* p) { return *reinterpret_cast<const Pack<T, N>*>(p); }
template <typename T, int N>
__device__ __forceinline__ void st(T* p, const Pack<T, N>& x) { *reinterpret_cast<Pack<T, N>*>(p) = x; }

// Write-through (sc1, agent scope) 16-byte store for the frames the 2D tile kernels write.  Those frames are consumed by the
// NEXT launch (all XCDs) or much later by the backward, never again by the launch that wrote them, and a dependent kernel
// boundary costs + (dirty bytes / ~6 TB/s) for the L2 write-back at kernel end (MI355X_MICROARCH.md, row "boundary": a K = 4
// forward launch leaves 8 MiB dirty).  Written through while the kernel still computes, that drain is off the boundary:
// measured on MI355X 512^2 forward 2.00 -> 1.84 us per step, lambda-omega 512^2 fp64 3.0 -> 2.78, 2048^2 16.3 -> 15.4
// (profiles/r02_write_through_stores.txt); the direct 3D kernels do not gain (128^3 11.3 -> 11.7) and keep plain stores.
// The compiler only emits sc1 on <= 8-byte atomic stores (two 8-byte sc1 stores per lane were measured SLOWER than plain
// stores: 512^2 forward 2.28 us), hence inline asm; the s_nop covers the gfx9 hazard "VMEM store of more than 64 bits
// followed by a write of its data VGPRs", which the hazard recogniser cannot see through inline asm (without it: corrupted
// frames -- caught by the parity tests).  PI_WT_STORES = 0 restores plain stores.
#ifndef PI_WT_STORES
#define PI_WT_STORES 1
#endif
template <typename T, int N>
__device__ __forceinline__ void st_frame_wt(T* p, const Pack<T, N>& x)
{
#if PI_WT_STORES
    if constexpr (sizeof(T) * N == 16) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u v = __builtin_bit_cast(v4u, x);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        *reinterpret_cast<Pack<T, N>*>(p) = x;
    }
#else
    *reinterpret_cast<Pack<T, N>*>(p) = x;
#endif
}

// Explicit fused multiply-add; the translation unit is built with -ffp-contract=off so every
// other a*b+c keeps its two roundings (the reference rounds `coef*lap + react` and `h + res*dt`
// separately -- train_2drd.py:115-118).
__device__ __forceinline__ float  fma_(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// the 10 wave-uniform scalars of one hidden channel {w1u,w1v,b1, w2u,w2v,b2, w3u,w3v,b3, w4}
template <typename T> struct W10 { T w[10]; };
template <typename T>
__device__ __forceinline__ W10<T> load_w10(const T* __restrict__ p)
{
    W10<T> r;
#pragma unroll
    for (int i = 0; i < 10; ++i) r.w[i] = p[i];
    return r;
}

// ---- pre-contracted reaction: r_s(u,v) = sum_m c[m] phi_m, phi = {1,u,v,u2,uv,v2,u3,u2v,uv2,v3} -------
// The Hadamard product of the three 1x1 branches followed by the 1x1 aggregation
// (train_2drd.py:115-116) IS this cubic; evaluating it in Horner form costs 9 FMAs per species
// instead of 72 VALU ops at Hc = 8, independent of Hc.
template <typename T>
__device__ __forceinline__ T poly_r(const T* __restrict__ c, T u, T v)
{
    const T A0 = fma_(v, fma_(v, fma_(v, c[9], c[5]), c[2]), c[0]);
    const T A1 = fma_(v, fma_(v, c[8], c[4]), c[1]);
    const T A2 = fma_(v, c[7], c[3]);
    return fma_(u, fma_(u, fma_(u, c[6], A2), A1), A0);
}
// dr/du and dr/dv
template <typename T>
__device__ __forceinline__ void poly_dr(const T* __restrict__ c, T u, T v, T& ru, T& rv)
{
    const T A1 = fma_(v, fma_(v, c[8], c[4]), c[1]);
    const T A2x2 = fma_(v, T(2) * c[7], T(2) * c[3]);
    ru = fma_(u, fma_(u, T(3) * c[6], A2x2), A1);
    const T B0 = fma_(v, fma_(v, T(3) * c[9], T(2) * c[5]), c[2]);
    const T B1 = fma_(v, T(2) * c[8], c[4]);
    rv = fma_(u, fma_(u, c[7], B1), B0);
}

// ---- loss gradient formed inside the adjoint sweep (round 3) -------------------------------------------------------------
// A sweep kernel is handed the frame to "inject" at step t-1 as a pointer.  mode 0: that frame IS dL/dh_{t-1} (the caller
// materialised it).  For the squared-error losses of the reference (train_2drd.py:397-407: MSE against data; SURVEY 8d:
// mean(traj^2)) the frame is a function of operands the sweep reads anyway, so the caller need not materialise it:
//   mode 1:  dL/dh_{t-1} = a * h_{t-1}                 (the pointer is the state frame itself, no extra bytes: 24 B / point)
//   mode 2:  dL/dh_{t-1} = a * (h_{t-1} - target_{t-1})  (the pointer is the target frame)
// with a = scale * (dev ? *dev : 1) -- the host factor (2 / N for a mean) times the scalar autograd hands the loss, read from
// device memory so that no host synchronisation is needed.  One subtraction and one multiplication, separately rounded:
// the same values ATen's mse_loss / pow backward writes into a materialised dL/dtraj.
struct LossInj {
    double scale;
    const void* dev;        // nullable; one element of the compute type
    int mode;               // 0, 1, 2
};
template <typename T>
__device__ __forceinline__ T loss_factor(const LossInj& l)
{
    return (T)l.scale * (l.dev ? *static_cast<const T*>(l.dev) : T(1));
}
// value to add to the adjoint state: j = the value loaded through the injection pointer, h = the state at that point
template <typename T>
__device__ __forceinline__ T loss_inject(int mode, T a, T h, T j)
{
    if (mode == 0) return j;
    const T d = mode == 2 ? h - j : h;
    return a * d;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc drain the vector-memory
// counter (s_waitcnt vmcnt(0)) first, which would serialise every software-prefetched global load and
// every in-flight trajectory store behind the barrier; LDS visibility needs lgkmcnt(0) alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }
// stencil neighbours: i in [-2, n+1] and n >= 2, so one conditional add/sub is the periodic wrap.  The runtime `%`
// above is ~20 quarter-rate integer instructions; eight of them per point made the direct 3D kernels VALU-bound.
__device__ __forceinline__ int wrap_near(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// exact x / d for 0 <= x < 2^31 from host-computed (m, s): d == 1 -> m = 0;  else l = ceil(log2 d),
// m = ceil(2^(31+l) / d) (< 2^32), s = l - 1 and  x / d == umulhi(x, m) >> s   (Granlund-Montgomery, N = 31)
struct FastDiv {
    unsigned m, s;
    __device__ __forceinline__ unsigned div(unsigned x) const { return m ? (__umulhi(x, m) >> s) : x; }
    // d >= 2 known: no test of m (a wave-uniform test becomes a scalar branch, i.e. a basic-block boundary that keeps the
    // scheduler from batching the loads on either side of it)
    __device__ __forceinline__ unsigned div_nz(unsigned x) const { return __umulhi(x, m) >> s; }
};

// XCD-aware block remap: hand each XCD (private 4 MiB L2) a contiguous range of the grid so the
// axis-0 neighbours of a block's rows are served by the same L2.  Pure speed; any mapping is correct.
// Any block count: XCD x (the blocks with bid % 8 == x) gets the contiguous range starting at x*q + min(x, r), q = n / 8,
// r = n % 8 -- the first r XCDs hold one block more.  (Was: identity unless n % 8 == 0, which cost a 1800^2 grid with a
// block count of 3165 a quarter of its speed against 3600 blocks.)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks)
{
    const unsigned q = nblocks / NXCD, r = nblocks % NXCD, x = bid % NXCD;
    return x * q + min(x, r) + bid / NXCD;
}

// ---- wave-level sum (all 64 lanes) -----------------------------------------------------------
#ifndef PI_USE_DPP
#define PI_USE_DPP 1      // 0 = ds_bpermute butterflies: 6 dependent LDS-crossbar round trips per sum (measured: the
#endif                    // two fp64 sums at the end of the tile adjoint kernel cost ~1 us that way)
#if PI_USE_DPP
// DPP row shifts + row broadcasts: 6 VALU adds, no LDS crossbar traffic. Total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum_to_last(T v)
{
    v += dpp_mov<0x111, 0xF>(v);   // row_shr:1
    v += dpp_mov<0x112, 0xF>(v);   // row_shr:2
    v += dpp_mov<0x114, 0xF>(v);   // row_shr:4
    v += dpp_mov<0x118, 0xF>(v);   // row_shr:8   -> lane 15 of each row holds the row total
    v += dpp_mov<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
    v += dpp_mov<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave total
    return v;
}
constexpr int REDUCE_LANE = 63;
#else
template <typename T>
__device__ __forceinline__ T wave_sum_to_last(T v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}
constexpr int REDUCE_LANE = 0;
#endif

}  // namespace pi
