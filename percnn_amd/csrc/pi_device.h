// pi_device.h -- device-side helpers shared by the Pi-block kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pi {

// ---- parameter block layout (documented in include/percnn_pi.h) ------------------------------
constexpr int P_DT = 0, P_COEF = 1, P_C0 = 3, P_TAPS = 4, P_W = 16;
__host__ __device__ constexpr int species_block(int hc) { return 10 * hc + 1; }
__host__ __device__ constexpr int nparams(int hc) { return P_W + 2 * species_block(hc); }

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

// Explicit fused multiply-add; the translation unit is built with -ffp-contract=off so every
// other a*b+c keeps its two roundings (the reference rounds `coef*lap + react` and `h + res*dt`
// separately -- train_2drd.py:115-118).
__device__ __forceinline__ float  fma_(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ int wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }

// XCD-aware block remap: hand each XCD (private 4 MiB L2) a contiguous range of the grid so the
// axis-0 neighbours of a block's rows are served by the same L2.  Pure speed; any mapping is correct.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks)
{
    if (nblocks % NXCD) return bid;
    return (bid % NXCD) * (nblocks / NXCD) + bid / NXCD;
}

// ---- wave-level sum (all 64 lanes) -----------------------------------------------------------
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
