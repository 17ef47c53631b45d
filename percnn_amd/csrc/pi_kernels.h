// pi_kernels.h -- fused Pi-block step kernels for gfx950 (forward and adjoint), v1 "direct" family.
//
// One lane owns VEC consecutive points of a grid row (16 B: 4 x f32 / 2 x f64), so every global
// access of a wave is a 1 KiB fully-coalesced transaction.  The radius-2 star stencil is gathered
// with index arithmetic (periodic wrap = the reference's torch.cat padding, train_2drd.py:108-109,
// at zero extra bytes); neighbour rows/planes are re-read through the vector L1 / XCD L2.  All
// 2*(10*hc+1) branch weights are wave-uniform and live in SGPRs (scalar loads from the parameter
// block); the six 1x1 convolutions, their Hadamard product, the 1x1 aggregation and the Euler
// update are pure register math (train_2drd.py:115-119).
#pragma once
#include "pi_device.h"

namespace pi {

#ifdef PI_3D_TIMING
// debug builds only (tools/ubench/step3d_probe.hip): 100 MHz wall-clock stamps of every wave of the direct 3D kernels
__device__ long long pi_3d_stamps[4096 * 8 * 8];                 // [block < 4096][wave < 8][slot < 8]
#define PI_STAMP3(i) do { if (threadIdx.x % 64 == 0 && blockIdx.x < 4096)                                                  \
                              pi_3d_stamps[(blockIdx.x * 8 + threadIdx.x / 64) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define PI_STAMP3(i) do { } while (0)
#endif

struct Geom {
    int n0, n1, W;      // extents: 2D {n0=H, W}; 3D {n0=D, n1=H, W}
    int rows;           // rows of W points = n0 (2D) or n0*n1 (3D)
    long s0;            // element stride of axis 0: W (2D) or n1*W (3D)
    long ss;            // species stride (elements); slab layout: (n0+4)*s0
    long off;           // element offset of the first interior point; slab layout: 2*s0
    int wrap0;          // 1: axis 0 periodic; 0: two halo planes present on each side (slab)
    FastDiv dcpr, dn1;  // chunk id -> (row, chunk in row) and row -> (plane, row in plane) without integer division;
    int fastdiv;        // set by the launcher when the chunk count is < 2^31 (else the 64-bit path below is used)
    // block-uniform decomposition of the direct step kernels (pi_fwd_kernel / pi_bwd_kernel): a workgroup covers
    // (blockDim >> lxs) rows x (1 << lxs) 16-byte chunks of ONE plane, so plane / row-group / x-block come from
    // blockIdx by scalar arithmetic and a lane only adds its (row, chunk) inside the block
    int lxs;            // log2 of the lanes along x; -1: flat (consecutive chunks of the plane, row = chunk / chunks per row)
    int nxb, nrg;       // x-blocks per row; row groups per plane (3D) or per grid (2D)
    unsigned nblk;      // virtual blocks = nxb * nrg * (3D: n0)
    FastDiv dnxb, dnrg;
    // 3D, planes too large for the L2: the rows of a plane are cut into y-tiles of `rgt` row groups and the block order is
    // (y-tile, plane, row group in tile, x block), so that the five planes a tile's stencil touches stay resident in the
    // XCD's L2 while the blocks march along axis 0 (rgt == 0: one tile = the whole plane)
    int rgt, nlast;     // row groups per full tile / in the last tile
    unsigned per_tile;  // rgt * n0
    FastDiv dper, drgt, dlast;
    unsigned xwin;      // forward kernel: XCD-contiguous remap inside windows of this many blocks (0 = the whole grid)
    int rz;             // 3D: consecutive planes one workgroup pass computes (plane neighbours shared in registers); the
                        // "planes" of the block decomposition above are groups of rz planes
    LossInj loss;       // adjoint kernels: what the injection pointer means (pi_device.h)
};

// radius-2 star: lap[i] = c0*f(x) + sum_axes sum_t w[axis][t] * f(x + FLIP*offs[t]); FLIP=-1 is the adjoint
template <typename T, int NDIM, int VEC, int FLIP>
__device__ __forceinline__ void star(const T* __restrict__ f, const T* __restrict__ P, const Geom& g,
                                     int i0, int i1, int x0, long e, const Pack<T, VEC>& c, T (&lap)[VEC])
{
#pragma unroll
    for (int i = 0; i < VEC; ++i) lap[i] = P[P_C0] * c.v[i];
    // axis 0 (slowest): periodic or halo-backed
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        int j0 = i0 + k;
        if (g.wrap0) j0 = wrap_near(j0, g.n0);
        const Pack<T, VEC> nb = ld<T, VEC>(f + e + (long)(j0 - i0) * g.s0);
        const T w = P[P_TAPS + t];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, nb.v[i], lap[i]);
    }
    if constexpr (NDIM == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = FLIP * (t < 2 ? t - 2 : t - 1);
            const int j1 = wrap_near(i1 + k, g.n1);
            const Pack<T, VEC> nb = ld<T, VEC>(f + e + (long)(j1 - i1) * g.W);
            const T w = P[P_TAPS + 4 + t];
#pragma unroll
            for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, nb.v[i], lap[i]);
        }
    }
    // fastest axis: window x0-2 .. x0+VEC+1
    T win[VEC + 4];
    const T* row = f + e - x0;
    if constexpr (VEC == 1) {
        win[0] = row[wrap_near(x0 - 2, g.W)];
        win[1] = row[wrap_near(x0 - 1, g.W)];
        win[3] = row[wrap_near(x0 + 1, g.W)];
        win[4] = row[wrap_near(x0 + 2, g.W)];
    } else {
        const int xl = x0 >= 2 ? x0 - 2 : x0 - 2 + g.W;
        const int xr = x0 + VEC < g.W ? x0 + VEC : x0 + VEC - g.W;
        const Pack<T, 2> l = ld<T, 2>(row + xl), r = ld<T, 2>(row + xr);
        win[0] = l.v[0]; win[1] = l.v[1];
        win[VEC + 2] = r.v[0]; win[VEC + 3] = r.v[1];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) win[2 + i] = c.v[i];
    constexpr int TX = P_TAPS + 4 * (NDIM - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        const T w = P[TX + t];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, win[2 + i + k], lap[i]);
    }
}

template <int NDIM>
__device__ __forceinline__ void chunk_coords(const Geom& g, long cid, int cpr, int vec, int& i0, int& i1, int& x0, long& e)
{
    if (g.fastdiv) {
        const unsigned row = g.dcpr.div((unsigned)cid);
        x0 = (int)((unsigned)cid - row * (unsigned)cpr) * vec;
        if constexpr (NDIM == 3) {
            i0 = (int)g.dn1.div(row);
            i1 = (int)(row - (unsigned)i0 * (unsigned)g.n1);
            e = (long)i0 * g.s0 + (long)i1 * g.W + x0;
        } else {
            i0 = (int)row;
            i1 = 0;
            e = (long)i0 * g.s0 + x0;
        }
        return;
    }
    const long row = cid / cpr;
    x0 = (int)(cid - row * cpr) * vec;
    if constexpr (NDIM == 3) {
        i0 = (int)(row / g.n1);
        i1 = (int)(row - (long)i0 * g.n1);
        e = (long)i0 * g.s0 + (long)i1 * g.W + x0;
    } else {
        i0 = (int)row;
        i1 = 0;
        e = (long)i0 * g.s0 + x0;
    }
}

// ---------------------------------------------------------------------------------------------
// Direct step kernels, addressing.
//
// The first version of these kernels decomposed a flat chunk id per lane (two integer divisions), kept 64-bit
// element offsets per neighbour and re-derived every wrap per lane: the 3D forward kernel executed ~295 VALU
// instructions per wave of which ~80 were floating point (ISA count: 38 v_lshl_add_u64, 25 v_mul_lo_u32, 17
// v_mad_u64_u32, 54 v_cndmask ...), i.e. at 128^3 it was bound by INTEGER issue (2.4 M VALU instructions per launch
// = 5.9 us of the 8.5 us kernel body), not by memory.  Here
//   * the plane (3D) / row group / x block of a workgroup are scalar (SALU) functions of blockIdx,
//   * plane neighbours are scalar base pointers (+- s0 in SGPRs), both species share every vector offset,
//   * a lane owns ONE 32-bit byte offset inside the plane; row neighbours are that offset +- k*W with one
//     compare/select for the periodic wrap; loads are `global_load v, v_off, s[base]` (scalar base + 32-bit offset).
// Arithmetic and its order are unchanged (bit-identical results).
// ---------------------------------------------------------------------------------------------
// The bases come out of sgpr_ptr (an asm constraint), after which the compiler no longer knows they point to global
// memory and would emit FLAT loads (64-bit VGPR addresses, counted on the LDS counter as well): say so explicitly ->
// global_load_dwordx4 v, v_off, s[base:base+1].
#define PI_GLOBAL __attribute__((address_space(1)))
template <int BYTES> struct RawBits;
template <> struct RawBits<4>  { typedef unsigned type; };
template <> struct RawBits<8>  { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct RawBits<16> { typedef unsigned type __attribute__((ext_vector_type(4))); };
template <typename T, int N>
__device__ __forceinline__ Pack<T, N> ldb(const char* __restrict__ base, unsigned byteoff)
{
    typedef typename RawBits<sizeof(Pack<T, N>)>::type R;
    const R r = *(const PI_GLOBAL R*)((const PI_GLOBAL char*)base + byteoff);
    return __builtin_bit_cast(Pack<T, N>, r);
}
template <typename T, int N>
__device__ __forceinline__ void stb(char* __restrict__ base, unsigned byteoff, const Pack<T, N>& x)
{
    typedef typename RawBits<sizeof(Pack<T, N>)>::type R;
    *(PI_GLOBAL R*)((PI_GLOBAL char*)base + byteoff) = __builtin_bit_cast(R, x);
}

// what a lane knows about its chunk: everything but `eb`, `row`, `x0` is wave-uniform
struct Lane {
    int i0;            // plane (3D; uniform), 0 in 2D
    int row;           // row inside the plane (3D: axis 1) or the grid (2D: axis 0)
    int x0;            // first point of the chunk
    unsigned eb;       // byte offset of the chunk relative to the plane base (3D) / biased field base (2D)
    bool valid;
};

// block id -> uniform (plane, row group, x block); lane -> (row, chunk).  Invalid lanes are clamped onto the last
// valid chunk (their loads stay in bounds, callers mask their results).
template <typename T, int NDIM, int VEC>
__device__ __forceinline__ Lane locate(const Geom& g, unsigned vb)
{
    const unsigned t = g.dnxb.div(vb);
    const unsigned xb = vb - t * (unsigned)g.nxb;
    unsigned rg = t, pl = 0;
    if constexpr (NDIM == 3) {
        if (g.rgt == 0) {
            pl = g.dnrg.div(t);
            rg = t - pl * (unsigned)g.nrg;
        } else {
            const unsigned yt = g.dper.div(t);                       // all tiles before the last one are full
            const unsigned r = t - yt * g.per_tile;
            const bool last = (yt + 1u) * (unsigned)g.rgt >= (unsigned)g.nrg;
            pl = last ? g.dlast.div(r) : g.drgt.div(r);
            rg = yt * (unsigned)g.rgt + (r - pl * (unsigned)(last ? g.nlast : g.rgt));
        }
    }
    const int cpr = g.W / VEC;
    const int nrow = NDIM == 3 ? g.n1 : g.n0;
    int chunk, row;
    Lane L;
    if (g.lxs < 0) {
        // flat decomposition (widths that no power of two of lanes covers well): the workgroup takes blockDim consecutive
        // chunks of the plane, rows follow each other in memory, so a wave still reads 1 KiB contiguous pieces; the lane
        // finds its row with one multiply-shift
        const unsigned total = (unsigned)nrow * (unsigned)cpr;
        unsigned idx = rg * blockDim.x + threadIdx.x;
        L.valid = idx < total;
        idx = min(idx, total - 1u);
        const unsigned r = g.dcpr.div(idx);
        row = (int)r;
        chunk = (int)(idx - r * (unsigned)cpr);
    } else {
        const int lx = 1 << g.lxs;
        const int xi = (int)threadIdx.x & (lx - 1), ri = (int)threadIdx.x >> g.lxs;
        const int rpb = (int)blockDim.x >> g.lxs;
        chunk = (int)xb * lx + xi;
        row = (int)rg * rpb + ri;
        L.valid = chunk < cpr && row < nrow;
        chunk = min(chunk, cpr - 1);
        row = min(row, nrow - 1);
    }
    L.i0 = NDIM == 3 ? (int)pl * g.rz : 0;          // first plane of the group
    L.row = row;
    L.x0 = chunk * VEC;
    // 2D: bias of two rows keeps the offsets of rows -2, -1 (slab layout: halo rows below the first computed one)
    // non-negative in unsigned arithmetic; the callers subtract it from the scalar base
    const unsigned bias = NDIM == 2 ? 2u * (unsigned)g.W : 0u;
    L.eb = ((unsigned)row * (unsigned)g.W + (unsigned)L.x0 + bias) * (unsigned)sizeof(T);
    return L;
}

// Make the components of a loaded chunk opaque at their point of use.  The x taps pair the chunk's values with their
// neighbours ((c1,c2), (c3,r0), ...) for v_pk_fma_f32; hipcc would rather re-load such a misaligned pair from memory
// (global_load_dwordx2 ... offset:4 / offset:8 next to the dwordx4 that already brought the chunk: +6 requests per pass of
// the 3D kernels) than spend two v_mov on it.
template <typename T, int N>
__device__ __forceinline__ void keep_in_regs(Pack<T, N>& c)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm("" : "+v"(c.v[i]));
}

// write-through (sc1) 16-byte store, scalar base + 32-bit lane offset; see st_frame_wt for what it buys and for the s_nop
template <typename T, int N>
__device__ __forceinline__ void stb_wt(char* base, unsigned byteoff, const Pack<T, N>& x)
{
    static_assert(sizeof(Pack<T, N>) == 16, "16-byte lanes");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_bit_cast(v4u, x);
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(byteoff), "v"(v), "s"(base) : "memory");
}

// Pin a wave-uniform pointer into an SGPR pair.  Without it LLVM reassociates  base + delta*s0 + zext(offset)  into a
// per-lane 64-bit multiply-add (v_mad_u64_u32 + a VGPR-pair address) for every plane neighbour.
__device__ __forceinline__ const char* sgpr_ptr(const char* p)
{
    asm("" : "+s"(p));
    return p;
}

// scalar base of species plane set `f` (already offset to species / first computed plane) for this block
template <typename T, int NDIM>
__device__ __forceinline__ const char* plane_base(const T* f, const Geom& g, int i0)
{
    if constexpr (NDIM == 3) return sgpr_ptr(reinterpret_cast<const char*>(f + (long)i0 * g.s0));
    else return sgpr_ptr(reinterpret_cast<const char*>(f) - (size_t)2 * g.W * sizeof(T));
}

// in-plane part of the radius-2 star (rows, then the fastest axis) accumulated onto `lap`; pb = base of the plane
template <typename T, int NDIM, int VEC, int FLIP>
__device__ __forceinline__ void star2_inplane(const char* __restrict__ pb, const T* __restrict__ P, const Geom& g,
                                              const Lane& L, const Pack<T, VEC>& c, T (&lap)[VEC])
{
    const unsigned Wb = (unsigned)g.W * (unsigned)sizeof(T);
    {
        // rows: axis 1 of a 3D plane (always periodic) or axis 0 of a 2D grid (periodic unless slab layout)
        const int nrow = NDIM == 3 ? g.n1 : g.n0;
        const bool wrap = NDIM == 3 ? true : (g.wrap0 != 0);
        const unsigned span = (unsigned)nrow * Wb;
        constexpr int TB = NDIM == 3 ? P_TAPS + 4 : P_TAPS;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = FLIP * (t < 2 ? t - 2 : t - 1);
            const int jr = L.row + k;
            unsigned off = L.eb + (unsigned)(k * (int)Wb);
            if (wrap) off += jr < 0 ? span : (jr >= nrow ? 0u - span : 0u);
#if defined(PI_EXPERIMENT) && PI_EXPERIMENT >= 2      // timing experiments only (wrong results): no row-neighbour loads
            const Pack<T, VEC> nb = c;
            (void)off;
#else
            const Pack<T, VEC> nb = ldb<T, VEC>(pb, off);
#endif
            const T w = P[TB + t];
#pragma unroll
            for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, nb.v[i], lap[i]);
        }
    }
    // fastest axis: window x0-2 .. x0+VEC+1 of the lane's own row
    T win[VEC + 4];
    const unsigned rowb = L.eb - (unsigned)L.x0 * (unsigned)sizeof(T);
    if constexpr (VEC == 1) {
        win[0] = ldb<T, 1>(pb, rowb + (unsigned)wrap_near(L.x0 - 2, g.W) * (unsigned)sizeof(T)).v[0];
        win[1] = ldb<T, 1>(pb, rowb + (unsigned)wrap_near(L.x0 - 1, g.W) * (unsigned)sizeof(T)).v[0];
        win[3] = ldb<T, 1>(pb, rowb + (unsigned)wrap_near(L.x0 + 1, g.W) * (unsigned)sizeof(T)).v[0];
        win[4] = ldb<T, 1>(pb, rowb + (unsigned)wrap_near(L.x0 + 2, g.W) * (unsigned)sizeof(T)).v[0];
    } else {
        const int xl = L.x0 >= 2 ? L.x0 - 2 : L.x0 - 2 + g.W;
        const int xr = L.x0 + VEC < g.W ? L.x0 + VEC : L.x0 + VEC - g.W;
        const Pack<T, 2> l = ldb<T, 2>(pb, rowb + (unsigned)xl * (unsigned)sizeof(T));
        const Pack<T, 2> r = ldb<T, 2>(pb, rowb + (unsigned)xr * (unsigned)sizeof(T));
        win[0] = l.v[0]; win[1] = l.v[1];
        win[VEC + 2] = r.v[0]; win[VEC + 3] = r.v[1];
    }
    Pack<T, VEC> cc = c;
    keep_in_regs(cc);
#pragma unroll
    for (int i = 0; i < VEC; ++i) win[2 + i] = cc.v[i];
    constexpr int TX = P_TAPS + 4 * (NDIM - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = FLIP * (t < 2 ? t - 2 : t - 1);
        const T w = P[TX + t];
#pragma unroll
        for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, win[2 + i + k], lap[i]);
    }
}

// The lane's chunk in planes i0-2 .. i0+RZ+1 of one species (3D): the register window the RZ output planes of a
// workgroup pass share -- RZ + 4 plane loads (scalar base pointers) instead of 5 * RZ.  fs = species base + g.off.
template <typename T, int VEC, int RZ>
struct PlaneWindow {
    Pack<T, VEC> w[RZ + 4];
    __device__ __forceinline__ void load(const T* fs, const Geom& g, const Lane& L)
    {
        // Branch-free on purpose: `if (g.wrap0)` on these wave-uniform values became scalar branches, i.e. one basic block per
        // plane -- ~45 SALU instructions and four taken branches between two loads of the window, and instruction selection
        // (per basic block) no longer saw `base + zext(offset)` and fell back to 64-bit VGPR addresses for all loads but one.
        const int nw = g.wrap0 ? g.n0 : 0;                  // periodic: wrap by n0; slab layout: planes -2, -1 exist as halos
        const int hi = g.wrap0 ? 0x7fffffff : g.n0 + 1;     // slab layout: two halo planes beyond the computed range exist
#pragma unroll
        for (int j = 0; j < RZ + 4; ++j) {
            int jz = L.i0 - 2 + j;
            jz += jz < 0 ? nw : 0;
            jz -= jz >= g.n0 ? nw : 0;
            if (RZ > 1) jz -= jz >= g.n0 ? nw : 0;          // partial last group of a grid with fewer than RZ + 2 planes
            jz = min(jz, hi);
#if defined(PI_EXPERIMENT) && PI_EXPERIMENT >= 1      // timing experiments only (wrong results): no plane-neighbour loads
            if (j < 2 || j >= RZ + 2) { w[j] = Pack<T, VEC>{}; continue; }
#endif
            w[j] = ldb<T, VEC>(sgpr_ptr(reinterpret_cast<const char*>(fs + (long)jz * g.s0)), L.eb);
        }
    }
    // centre + axis-0 taps of output plane j (0 <= j < RZ): the first five terms of the star, in pi::star's order
    template <int FLIP>
    __device__ __forceinline__ void planes(int j, const T* __restrict__ P, T (&lap)[VEC]) const
    {
#pragma unroll
        for (int i = 0; i < VEC; ++i) lap[i] = P[P_C0] * w[j + 2].v[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = FLIP * (t < 2 ? t - 2 : t - 1);
            const T wt = P[P_TAPS + t];
#pragma unroll
            for (int i = 0; i < VEC; ++i) lap[i] = fma_(wt, w[j + 2 + k].v[i], lap[i]);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// forward: out = h + dt * (coef * Lap(h) + Wh4(Wh1(h) * Wh2(h) * Wh3(h)))
// ---------------------------------------------------------------------------------------------
template <typename T, int NDIM, int HC, int VEC, int RZ = 1>
__global__ void __launch_bounds__(256)
pi_fwd_kernel(const T* __restrict__ h, T* __restrict__ out, const T* __restrict__ P, Geom g, int hc_rt)
{
    static_assert(NDIM == 3 || RZ == 1, "plane blocking is a 3D notion");
    const int hc = HC > 0 ? HC : hc_rt;      // unused when HC == POLY
    // one virtual block (plane group, row group, x block) per workgroup, or -- option fwd_blocks -- a bounded grid of
    // workgroups that walk the virtual blocks in order (measured slower: 384^3 376 -> 416 us)
    unsigned first = xcd_remap(blockIdx.x, gridDim.x);
    if (g.xwin) {                                   // all XCDs inside one window of the grid at a time (see launch_fwd)
        const unsigned base = blockIdx.x / g.xwin * g.xwin;
        const unsigned len = min(g.xwin, gridDim.x - base);
        first = base + xcd_remap(blockIdx.x - base, len);
    }
    const T dt = P[P_DT];
    PI_STAMP3(0);
    for (unsigned vb = first; vb < g.nblk; vb += gridDim.x) {
        const Lane L = locate<T, NDIM, VEC>(g, vb);
        if (!L.valid) continue;
        const T* hs[2] = {h + g.off, h + g.ss + g.off};
        // 3D: the lane's chunk in planes i0-2 .. i0+RZ+1, both species, requested up front (RZ + 4 loads per species
        // serve RZ output planes)
        PlaneWindow<T, VEC, NDIM == 3 ? RZ : 1> win[2];
        if constexpr (NDIM == 3) {
            win[0].load(hs[0], g, L);
            win[1].load(hs[1], g, L);
        }
#pragma unroll
        for (int j = 0; j < RZ; ++j) {
            const int iz = L.i0 + j;
            if (NDIM == 3 && iz >= g.n0) break;          // partial last plane group (block-uniform)
            const char* pu = plane_base<T, NDIM>(hs[0], g, iz);
            const char* pv = plane_base<T, NDIM>(hs[1], g, iz);
            Pack<T, VEC> cu, cv;
            T lap[2][VEC];
            if constexpr (NDIM == 3) {
                cu = win[0].w[j + 2];
                cv = win[1].w[j + 2];
                win[0].template planes<+1>(j, P, lap[0]);
                win[1].template planes<+1>(j, P, lap[1]);
            } else {
                cu = ldb<T, VEC>(pu, L.eb);
                cv = ldb<T, VEC>(pv, L.eb);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { lap[0][i] = P[P_C0] * cu.v[i]; lap[1][i] = P[P_C0] * cv.v[i]; }
            }
            star2_inplane<T, NDIM, VEC, +1>(pu, P, g, L, cu, lap[0]);
            star2_inplane<T, NDIM, VEC, +1>(pv, P, g, L, cv, lap[1]);

            // The species / hidden-channel loops stay ROLLED on purpose: a fully unrolled body is several KiB
            // of straight-line code that every wave executes exactly once, and at one wave per SIMD the
            // kernel then runs at instruction-fetch speed (measured ~16 cycles per VALU op).  The rolled
            // body is ~40 instructions, I$-resident, with next channel's 10 scalars prefetched into SGPRs.
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
                    for (int jj = 0; jj < hc; ++jj) {
                        const W10<T> c = nx;
                        if (jj + 1 < hc) nx = load_w10(W + 10 * (jj + 1));
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
                    const T hv = s == 0 ? cu.v[i] : cv.v[i];
                    const T lp = s == 0 ? lap[0][i] : lap[1][i];
                    const T res = coef * lp + rr[i];            // two roundings (train_2drd.py:115)
                    const T inc = res * dt;                     // two roundings (train_2drd.py:117)
                    o.v[i] = hv + inc;
                }
                char* po = const_cast<char*>(plane_base<T, NDIM>(out + s * g.ss + g.off, g, iz));
                stb<T, VEC>(po, L.eb, o);
            }
            PI_STAMP3(4 + (j > 0));
        }
    }
    PI_STAMP3(7);
}

// ---------------------------------------------------------------------------------------------
// adjoint of one step.
//   G    = dL/d(step output)            (stencil-read, halo-backed in slab mode)
//   Gp   = G + coef*dt*LapT(G) + dt*J_react(h)^T G (+ inj)
//   partials[block][np] += this block's parameter-gradient sums (double, owner-block RMW)
// Gradient of the diffusion coefficient uses sum_x g*Lap(h) == sum_x LapT(g)*h, so the forward
// Laplacian is never recomputed.
// ---------------------------------------------------------------------------------------------
template <typename T, int NDIM, int HC, int VEC, bool WGRAD, int RZ = 1>
__global__ void __launch_bounds__(256)
pi_bwd_kernel(const T* __restrict__ h, const T* __restrict__ G, const T* __restrict__ inj, T* __restrict__ Gp,
              double* __restrict__ partials, const T* __restrict__ P, Geom g, int hc_rt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);           // [nwaves][np] running sums of this block

    const int hc = HC == POLY ? 0 : (HC > 0 ? HC : hc_rt);
    const int np = nparams(hc);
    const int nwaves = blockDim.x / WAVE;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    T* myred = red + wave * np;
    double* redc = reinterpret_cast<double*>(smem_raw + ((size_t)nwaves * np * sizeof(T) + 15) / 16 * 16);   // [nwaves][2]
    for (int i = threadIdx.x; i < nwaves * np; i += blockDim.x) red[i] = T(0);
    if (threadIdx.x < 2 * nwaves) redc[threadIdx.x] = 0.0;
    // running partial of this workgroup's row: requested NOW, needed at the very end (a dependent load -> add -> store
    // in the tail of every launch otherwise; same fix as in the tile sweep)
    auto carries_grad = [&](int idx) {
        if (idx >= np || idx == P_DT || (idx >= P_C0 && idx < P_W)) return false;   // dt, frozen stencil: no gradient
        return WGRAD || idx < P_W;                                                  // sweep-only flavour: coefficients only
    };
    double* const prow = partials + (long)blockIdx.x * np;
    const double pold = carries_grad((int)threadIdx.x) ? prow[threadIdx.x] : 0.0;
    __syncthreads();

    const T dt = P[P_DT];

    double lane_c[2] = {0.0, 0.0};
    // poly mode with fused gradients: the 20 coefficient moments stay in registers over all chunks of the lane and are
    // reduced across lanes once per launch (was: 22 wave reductions per chunk)
    constexpr bool LANE_MOM = WGRAD && HC == POLY;
    T macc[LANE_MOM ? 2 : 1][LANE_MOM ? 10 : 1];
    if constexpr (LANE_MOM) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) macc[s][m] = T(0);
    }

    // virtual blocks (plane, row group, x block) of this workgroup: block-uniform trip count (wave-level reductions
    // inside need whole waves, which a uniform loop guarantees); addressing as in the forward kernel
    for (unsigned vb = xcd_remap(blockIdx.x, gridDim.x); vb < g.nblk; vb += gridDim.x) {
        const Lane L = locate<T, NDIM, VEC>(g, vb);
        const bool valid = L.valid;
        // 3D: the adjoint state of planes i0-2 .. i0+RZ+1 in registers, shared by the RZ output planes of this pass
        PlaneWindow<T, VEC, NDIM == 3 ? RZ : 1> win[2];
        if constexpr (NDIM == 3) {
            win[0].load(G + g.off, g, L);
            win[1].load(G + g.ss + g.off, g, L);
        }
#pragma unroll
        for (int jz = 0; jz < RZ; ++jz) {
        const int iz = L.i0 + jz;
        if (NDIM == 3 && iz >= g.n0) break;              // partial last plane group (block-uniform)
        const char* phu = plane_base<T, NDIM>(h + g.off, g, iz);
        const char* phv = plane_base<T, NDIM>(h + g.ss + g.off, g, iz);
        const char* pgu = plane_base<T, NDIM>(G + g.off, g, iz);
        const char* pgv = plane_base<T, NDIM>(G + g.ss + g.off, g, iz);
        const Pack<T, VEC> u = ldb<T, VEC>(phu, L.eb), v = ldb<T, VEC>(phv, L.eb);
        Pack<T, VEC> gc[2];
        T dl[2][VEC];
        if constexpr (NDIM == 3) {
            gc[0] = win[0].w[jz + 2];
            gc[1] = win[1].w[jz + 2];
            win[0].template planes<-1>(jz, P, dl[0]);
            win[1].template planes<-1>(jz, P, dl[1]);
        } else {
            gc[0] = ldb<T, VEC>(pgu, L.eb);
            gc[1] = ldb<T, VEC>(pgv, L.eb);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { dl[0][i] = P[P_C0] * gc[0].v[i]; dl[1][i] = P[P_C0] * gc[1].v[i]; }
        }
        star2_inplane<T, NDIM, VEC, -1>(pgu, P, g, L, gc[0], dl[0]);
        star2_inplane<T, NDIM, VEC, -1>(pgv, P, g, L, gc[1], dl[1]);
        const T live = valid ? T(1) : T(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                dl[s][i] = (dl[s][i] * dt) * live;
                gc[s].v[i] *= live;
            }

        T du[VEC], dv[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) du[i] = dv[i] = T(0);

        if constexpr (HC == POLY) {
            // monomials shared by both species (only needed for the moment sums)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const T* c = P + P_W + 10 * s;
                const int gbase = P_W + 10 * s;
                const Pack<T, VEC>& hs = s == 0 ? u : v;
                (void)gbase;
                double acc_c = 0.0;                      // heavily cancelling sum (stencil row-sum ~ 0): keep it in fp64
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const T gr = gc[s].v[i] * dt;
                    acc_c += (double)(dl[s][i] * hs.v[i]);
                    T ru, rv;
                    poly_dr(c, u.v[i], v.v[i], ru, rv);
                    du[i] = fma_(gr, ru, du[i]);
                    dv[i] = fma_(gr, rv, dv[i]);
                    if constexpr (WGRAD) {
                        T (&acc)[10] = macc[s];
                        const T uu = u.v[i], vv = v.v[i];
                        const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
                        acc[0] += gr;
                        acc[1] = fma_(gr, uu, acc[1]); acc[2] = fma_(gr, vv, acc[2]);
                        acc[3] = fma_(gr, u2, acc[3]); acc[4] = fma_(gr, uv, acc[4]); acc[5] = fma_(gr, v2, acc[5]);
                        acc[6] = fma_(gr, u2 * uu, acc[6]); acc[7] = fma_(gr, u2 * vv, acc[7]);
                        acc[8] = fma_(gr, uu * v2, acc[8]); acc[9] = fma_(gr, v2 * vv, acc[9]);
                    }
                }
                lane_c[s] += acc_c;                      // one cross-lane reduction per launch, not per chunk
            }
        } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const T* W = P + P_W + s * species_block(hc);
            const int gbase = P_W + s * species_block(hc);
            const Pack<T, VEC>& hs = s == 0 ? u : v;
            T gr[VEC];
            double acc_c = 0.0;                          // heavily cancelling sum: fp64
            T acc_b4 = T(0);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                gr[i] = gc[s].v[i] * dt;
                acc_c += (double)(dl[s][i] * hs.v[i]);
                acc_b4 += gr[i];
            }
            if constexpr (WGRAD) {
                acc_c = wave_sum_to_last(acc_c);
                acc_b4 = wave_sum_to_last(acc_b4);
                if (lane == REDUCE_LANE) {
                    redc[wave * 2 + s] += acc_c;
                    myred[gbase + 10 * hc] += acc_b4;
                }
            } else {
                lane_c[s] += acc_c;
            }
            auto channel = [&](int j) {
                const T* w = W + 10 * j;
                const T w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], w5 = w[5], w6 = w[6], w7 = w[7],
                        w8 = w[8], w9 = w[9];
                T acc[10];
#pragma unroll
                for (int m = 0; m < 10; ++m) acc[m] = T(0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const T a1 = fma_(w0, u.v[i], fma_(w1, v.v[i], w2));
                    const T a2 = fma_(w3, u.v[i], fma_(w4, v.v[i], w5));
                    const T a3 = fma_(w6, u.v[i], fma_(w7, v.v[i], w8));
                    const T p12 = a1 * a2;
                    const T gw = gr[i] * w9;
                    const T q1 = gw * (a2 * a3), q2 = gw * (a1 * a3), q3 = gw * p12;
                    if constexpr (WGRAD) {
                        acc[9] += gr[i] * (p12 * a3);
                        acc[0] += q1 * u.v[i]; acc[1] += q1 * v.v[i]; acc[2] += q1;
                        acc[3] += q2 * u.v[i]; acc[4] += q2 * v.v[i]; acc[5] += q2;
                        acc[6] += q3 * u.v[i]; acc[7] += q3 * v.v[i]; acc[8] += q3;
                    }
                    du[i] = fma_(q1, w0, fma_(q2, w3, fma_(q3, w6, du[i])));
                    dv[i] = fma_(q1, w1, fma_(q2, w4, fma_(q3, w7, dv[i])));
                }
                if constexpr (WGRAD) {
#pragma unroll
                    for (int m = 0; m < 10; ++m) acc[m] = wave_sum_to_last(acc[m]);
                    if (lane == REDUCE_LANE) {
#pragma unroll
                        for (int m = 0; m < 10; ++m) myred[gbase + 10 * j + m] += acc[m];
                    }
                }
                        };
            if constexpr (HC > 0) {                      // compile-time width: fully unrolled
#pragma unroll
                for (int j = 0; j < HC; ++j) channel(j);
            } else {
                for (int j = 0; j < hc; ++j) channel(j);
            }
        }

        }

        if (valid) {
            Pack<T, VEC> ou, ov;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const T tu = P[P_COEF + 0] * dl[0][i] + du[i];
                const T tv = P[P_COEF + 1] * dl[1][i] + dv[i];
                ou.v[i] = gc[0].v[i] + tu;
                ov.v[i] = gc[1].v[i] + tv;
            }
            if (inj) {
                Pack<T, VEC> ju = u, jv = v;
                if (g.loss.mode != 1) {                                  // mode 1 injects a function of the state alone
                    ju = ldb<T, VEC>(plane_base<T, NDIM>(inj + g.off, g, iz), L.eb);
                    jv = ldb<T, VEC>(plane_base<T, NDIM>(inj + g.ss + g.off, g, iz), L.eb);
                }
                const T la = g.loss.mode ? loss_factor<T>(g.loss) : T(0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    ou.v[i] += loss_inject(g.loss.mode, la, u.v[i], ju.v[i]);
                    ov.v[i] += loss_inject(g.loss.mode, la, v.v[i], jv.v[i]);
                }
            }
            stb<T, VEC>(const_cast<char*>(plane_base<T, NDIM>(Gp + g.off, g, iz)), L.eb, ou);
            stb<T, VEC>(const_cast<char*>(plane_base<T, NDIM>(Gp + g.ss + g.off, g, iz)), L.eb, ov);
        }
        }   // planes of the group
    }

    if constexpr (!WGRAD || LANE_MOM) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const double r = wave_sum_to_last(lane_c[s]);
            if (lane == REDUCE_LANE) redc[wave * 2 + s] += r;
        }
    }
    if constexpr (LANE_MOM) {
        // Block-wide sums of the 20 per-lane moments through an LDS transpose: every thread writes its 20 values, then 8
        // lanes per moment add NT/8 values each and fold with three DPP steps.  The earlier form (20 six-step DPP wave
        // reductions per wave) cost 1.6 us of a 21.6 us launch at 128^3 -- measured by removing it (timing experiment) --
        // because every wave runs it in the tail of the launch, when nothing is left to overlap it with.
        const int NT = (int)blockDim.x, RS = NT + 8;                 // row stride: + 8 floats -> 8 rows cover all banks
        T* scr = reinterpret_cast<T*>(smem_raw + (((size_t)nwaves * np * sizeof(T) + 15) / 16 * 16) +
                                      (size_t)nwaves * 2 * sizeof(double));
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) scr[(10 * s + m) * RS + (int)threadIdx.x] = macc[s][m];
        __syncthreads();
        for (int base = 0; base < 160; base += NT) {                 // uniform trip count: whole waves run the DPP steps
            const int task = base + (int)threadIdx.x;
            const int mm = min(task, 159) >> 3, part = task & 7;
            T a = T(0);
            if (task < 160) {
                // NT / 8 = 8 .. 32 values per lane, NT a multiple of 64: eight loads in flight, four partial sums
                T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
                const T* row = scr + mm * RS + part;
                for (int k = 0; k < NT; k += 64) {
                    const T v0 = row[k], v1 = row[k + 8], v2 = row[k + 16], v3 = row[k + 24];
                    const T v4 = row[k + 32], v5 = row[k + 40], v6 = row[k + 48], v7 = row[k + 56];
                    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
                    a0 += v4; a1 += v5; a2 += v6; a3 += v7;
                }
                a = (a0 + a1) + (a2 + a3);
            }
            a += dpp_mov<0x111, 0xF>(a);                             // row_shr:1, :2, :4 -> lane 7 of each group of 8
            a += dpp_mov<0x112, 0xF>(a);
            a += dpp_mov<0x114, 0xF>(a);
            if (task < 160 && part == 7) red[P_W + mm] = a;          // wave 0's row of `red` (the others stay zero)
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < np; idx += blockDim.x) {
        if (!carries_grad(idx)) continue;
        double s = 0.0;
        if (idx == P_COEF || idx == P_COEF + 1)
            for (int w = 0; w < nwaves; ++w) s += redc[w * 2 + idx - P_COEF];
        else
            for (int w = 0; w < nwaves; ++w) s += (double)red[w * np + idx];
        prow[idx] = (idx == (int)threadIdx.x ? pold : prow[idx]) + s;
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradients of the whole rollout in ONE launch (the "wgrad" kernel).
//
// The reverse sweep is sequential in time, but once the adjoint trajectory adj[t] = dL/dh_t is
// stored, the parameter gradients  sum_t sum_x (...)  are an embarrassingly parallel reduction over
// all T*P (step, point) pairs.  Every lane streams (h_{t-1}, adj_t) chunks with 16-byte loads,
// keeps its NS*(10*JC+1) running sums in registers for the whole kernel and the cross-lane
// reduction happens exactly once at the end -- no per-step reductions, no per-step launch
// latency, full chip.  NS = 2: one workgroup differentiates both species' branches (small hc, the
// state is read once); NS = 1: blockIdx.y picks the species.  Hidden channels [j0, j0+JC).
// ---------------------------------------------------------------------------------------------
template <typename T, int JC, int NS, int VEC>
__global__ void __launch_bounds__(256)
pi_wgrad_kernel(const T* __restrict__ traj, const T* __restrict__ adj, double* __restrict__ partials,
                const T* __restrict__ P, long n, long ss, long off, int t_lo, int t_hi, int hc, int j0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);            // [nwaves][NS*(10*JC+1)]
    constexpr int NA = 10 * JC + 1;
    const int np = nparams(hc);
    const T dt = P[P_DT];
    // frames are [2][ss] with the n interior points of a species starting at `off` (slab layout: halo planes
    // skipped; plain layout: ss = n, off = 0)
    const long frame = 2 * ss;
    const long cpf = n / VEC;                            // chunks per frame
    const long nsteps = t_hi - t_lo;
    const long stride = (long)gridDim.x * blockDim.x;
    const long stride_t = stride / cpf, stride_x = stride - stride_t * cpf;

    T acc[NS][JC][10];
    T acc_b4[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        acc_b4[q] = T(0);
#pragma unroll
        for (int jj = 0; jj < JC; ++jj)
#pragma unroll
            for (int m = 0; m < 10; ++m) acc[q][jj][m] = T(0);
    }

    const long c0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tt = c0 / cpf, xc = c0 - tt * cpf;
    while (tt < nsteps) {
        const long t = t_lo + 1 + tt;                    // step t maps frame t-1 -> frame t
        const long x = off + xc * VEC;
        const Pack<T, VEC> u = ld<T, VEC>(traj + (t - 1) * frame + x);
        const Pack<T, VEC> v = ld<T, VEC>(traj + (t - 1) * frame + ss + x);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int s = NS == 2 ? q : (int)blockIdx.y;
            const T* W = P + P_W + s * species_block(hc) + 10 * j0;
            const Pack<T, VEC> a = ld<T, VEC>(adj + t * frame + s * ss + x);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const T gr = a.v[i] * dt;
                acc_b4[q] += gr;
#pragma unroll
                for (int jj = 0; jj < JC; ++jj) {
                    const T* w = W + 10 * jj;
                    const T a1 = fma_(w[0], u.v[i], fma_(w[1], v.v[i], w[2]));
                    const T a2 = fma_(w[3], u.v[i], fma_(w[4], v.v[i], w[5]));
                    const T a3 = fma_(w[6], u.v[i], fma_(w[7], v.v[i], w[8]));
                    const T p12 = a1 * a2;
                    const T gw = gr * w[9];
                    const T q1 = gw * (a2 * a3), q2 = gw * (a1 * a3), q3 = gw * p12;
                    T* A = acc[q][jj];
                    A[9] = fma_(gr, p12 * a3, A[9]);
                    A[0] = fma_(q1, u.v[i], A[0]); A[1] = fma_(q1, v.v[i], A[1]); A[2] += q1;
                    A[3] = fma_(q2, u.v[i], A[3]); A[4] = fma_(q2, v.v[i], A[4]); A[5] += q2;
                    A[6] = fma_(q3, u.v[i], A[6]); A[7] = fma_(q3, v.v[i], A[7]); A[8] += q3;
                }
            }
        }
        xc += stride_x;
        tt += stride_t;
        if (xc >= cpf) { xc -= cpf; ++tt; }
    }

    const int nwaves = blockDim.x / WAVE;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
        for (int jj = 0; jj < JC; ++jj)
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                const T r = wave_sum_to_last(acc[q][jj][m]);
                if (lane == REDUCE_LANE) red[(wave * NS + q) * NA + 10 * jj + m] = r;
            }
        const T r = wave_sum_to_last(acc_b4[q]);
        if (lane == REDUCE_LANE) red[(wave * NS + q) * NA + 10 * JC] = r;
    }
    __syncthreads();
    const long row = (long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int k = threadIdx.x; k < NS * NA; k += blockDim.x) {
        const int q = k / NA, idx = k - q * NA;
        if (idx == 10 * JC && j0 != 0) continue;         // the Wh4 bias is accumulated by the j0 == 0 pass only
        const int s = NS == 2 ? q : (int)blockIdx.y;
        const int gbase = P_W + s * species_block(hc);
        T sum = T(0);
        for (int w = 0; w < nwaves; ++w) sum += red[(w * NS + q) * NA + idx];
        const int col = idx == 10 * JC ? gbase + 10 * hc : gbase + 10 * j0 + idx;
        partials[row * np + col] += (double)sum;
    }
}

// ---------------------------------------------------------------------------------------------
// pre-contracted mode: gradients w.r.t. the 2 x 10 cubic coefficients are the "moments"
//   dL/dc[s][m] = sum_t sum_x (adj_t[s](x) * dt) * phi_m(h_{t-1}(x))
// again one time-parallel streaming reduction over all (step, point) pairs (16 B per point-step read,
// ~30 VALU ops): HBM-bound.  The map back to the branch weights (dc/dW, multilinear) is a
// 20 x (2*(10*hc+1)) chain rule done by the caller.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
pi_moments_kernel(const T* __restrict__ traj, const T* __restrict__ adj, double* __restrict__ partials,
                  const T* __restrict__ P, long n, long ss, long off, int t_lo, int t_hi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* red = reinterpret_cast<T*>(smem_raw);            // [nwaves][20]
    const T dt = P[P_DT];
    const long frame = 2 * ss;
    const long cpf = n / VEC;
    const long nsteps = t_hi - t_lo;
    const long stride = (long)gridDim.x * blockDim.x;
    const long stride_t = stride / cpf, stride_x = stride - stride_t * cpf;
    T acc[2][10];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 10; ++m) acc[s][m] = T(0);

    const long c0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long tt = c0 / cpf, xc = c0 - tt * cpf;
    while (tt < nsteps) {
        const long t = t_lo + 1 + tt;
        const long x = off + xc * VEC;
        const Pack<T, VEC> u = ld<T, VEC>(traj + (t - 1) * frame + x);
        const Pack<T, VEC> v = ld<T, VEC>(traj + (t - 1) * frame + ss + x);
        const Pack<T, VEC> au = ld<T, VEC>(adj + t * frame + x);
        const Pack<T, VEC> av = ld<T, VEC>(adj + t * frame + ss + x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const T uu = u.v[i], vv = v.v[i];
            const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
            const T phi[10] = {T(1), uu, vv, u2, uv, v2, u2 * uu, u2 * vv, uu * v2, v2 * vv};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const T gr = (s == 0 ? au.v[i] : av.v[i]) * dt;
                acc[s][0] += gr;
#pragma unroll
                for (int m = 1; m < 10; ++m) acc[s][m] = fma_(gr, phi[m], acc[s][m]);
            }
        }
        xc += stride_x;
        tt += stride_t;
        if (xc >= cpf) { xc -= cpf; ++tt; }
    }
    const int nwaves = blockDim.x / WAVE;
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 10; ++m) {
            const T r = wave_sum_to_last(acc[s][m]);
            if (lane == REDUCE_LANE) red[wave * 20 + 10 * s + m] = r;
        }
    __syncthreads();
    if (threadIdx.x < 20) {
        T sum = T(0);
        for (int w = 0; w < nwaves; ++w) sum += red[w * 20 + threadIdx.x];
        partials[(long)blockIdx.x * NPOLY + P_W + threadIdx.x] += (double)sum;
    }
}

// Frame-parallel launches (grid = chunk blocks x frames), XCD-aware: workgroup b runs on XCD b % 8, and with the plain
// (blockIdx.x, blockIdx.y) = (chunk block, frame) grid the 8 XCDs interleave the ROWS of a frame -- every XCD's L2 then fetches
// the four neighbour rows of each of its rows from the fabric itself (5x the algorithmic reads: the lambda-omega 512^2 loss
// pass ran at 1.1 TB/s).  Here the launch is 1-D, XCD x owns the contiguous chunk blocks [x * per, (x + 1) * per) of EVERY frame
// (per = ceil(gx / 8)) and walks them frame by frame: neighbour rows come from the XCD's own L2 except at its two seams.
struct FrameGrid { unsigned gx, gy, per; };          // chunk blocks per frame, frame slots, chunk blocks per frame and XCD
__host__ __device__ inline unsigned frame_grid_blocks(const FrameGrid& fg) { return NXCD * fg.per * fg.gy; }
struct FrameBlock { unsigned bx, by; bool live; };
__device__ __forceinline__ FrameBlock frame_block(const FrameGrid& fg)
{
    const unsigned b = blockIdx.x, x = b % NXCD, j = b / NXCD;
    const unsigned by = j / fg.per, bx = x * fg.per + (j - by * fg.per);
    return FrameBlock{bx, by, bx < fg.gx};
}

// ---------------------------------------------------------------------------------------------
// physics residual of a polynomial reaction-diffusion equation over a whole trajectory, time-parallel
// (reference: loss_generator.get_phy_Loss, train_2drd.py:270-329, train_3drd.py:287-323,
// percnn_LO_eqn.py:283-341 -- a 5x5(x5) Laplacian conv over all frames + a permute/reshape/Conv1d time
// difference; here ONE launch, blockIdx.y = frame):
//   R_s(f, x) = coef_s * Lap(h_f)_s + r_s(h_f) - (h_{f+1,s} - h_{f,s}) / dt
// Q is a pre-contracted block holding the TRUE equation (coefficients of the PDE, not of the model).
// ---------------------------------------------------------------------------------------------
template <typename T, int NDIM, int VEC>
__global__ void __launch_bounds__(256)
pi_residual_kernel(const T* __restrict__ traj, T* __restrict__ R, const T* __restrict__ Q, Geom g, FrameGrid fg)
{
    const FrameBlock fb = frame_block(fg);
    const int cpr = g.W / VEC;
    const long nchunks = (long)g.rows * cpr;
    const long cid = (long)fb.bx * blockDim.x + threadIdx.x;
    if (!fb.live || cid >= nchunks) return;
    int i0, i1, x0;
    long e;
    chunk_coords<NDIM>(g, cid, cpr, VEC, i0, i1, x0, e);
    const long frame = 2 * g.ss;
    const T* h = traj + (long)fb.by * frame;
    const T* hn = h + frame;
    const Pack<T, VEC> cu = ld<T, VEC>(h + e), cv = ld<T, VEC>(h + g.ss + e);
    T lap[2][VEC];
    star<T, NDIM, VEC, +1>(h, Q, g, i0, i1, x0, e, cu, lap[0]);
    star<T, NDIM, VEC, +1>(h + g.ss, Q, g, i0, i1, x0, e, cv, lap[1]);
    const T dt = Q[P_DT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const Pack<T, VEC> nx = ld<T, VEC>(hn + s * g.ss + e);
        const T* c = Q + P_W + 10 * s;
        Pack<T, VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const T hs = s == 0 ? cu.v[i] : cv.v[i];
            const T rhs = Q[P_COEF + s] * lap[s][i] + poly_r(c, cu.v[i], cv.v[i]);
            o.v[i] = rhs - (nx.v[i] - hs) / dt;
        }
        st<T, VEC>(R + (long)fb.by * frame + s * g.ss + e, o);
    }
}

// The residual LOSS without a materialised residual (round 3; reference: loss_gen, train_2drd.py:340-353, percnn_LO_eqn.py:
// 343-357 -- MSE of f_u plus MSE of f_v).  The reference pads 2 cells low / 3 high, i.e. it evaluates on an (N+1)^d grid in which
// index 0 of every axis appears twice: weight w(x) = 2^(number of zero coordinates) (ResLoss::weighted), else 1.
//   GRAD = false: partials[block] = sum over this block's chunks and its frames f = blockIdx.y, += gridDim.y of w * R^2 (double)
//   GRAD = true : G(f, x) = a * w * R(f, x),  a = 2 * scale * (upstream gradient, read from the device) -- what
//                 pi_residual_adj_kernel<FULL> turns into dL/dtraj
struct ResLoss {
    double scale;          // 1 / (frames * prod(n + 1))  (weighted)  or  1 / (frames * prod(n))
    const void* g_dev;     // GRAD: device pointer to the upstream scalar gradient in the compute type (nullptr = 1)
    int weighted;
};

template <typename T, int NDIM, int VEC, bool GRAD>
__global__ void __launch_bounds__(256)
pi_residual_sq_kernel(const T* __restrict__ traj, T* __restrict__ G, double* __restrict__ partials,
                      const T* __restrict__ Q, Geom g, int nframes, ResLoss rl, FrameGrid fg)
{
    __shared__ double red[256 / WAVE];
    const FrameBlock fb = frame_block(fg);
    if (!fb.live) return;
    const int cpr = g.W / VEC;
    const long nchunks = (long)g.rows * cpr;
    const long cid = (long)fb.bx * blockDim.x + threadIdx.x;
    const bool live = cid < nchunks;
    int i0 = 0, i1 = 0, x0 = 0;
    long e = 0;
    if (live) chunk_coords<NDIM>(g, cid, cpr, VEC, i0, i1, x0, e);
    const long frame = 2 * g.ss;
    const T dt = Q[P_DT];
    const T wrow = rl.weighted ? T((i0 == 0 ? 2 : 1) * ((NDIM == 3 && i1 == 0) ? 2 : 1)) : T(1);
    T a = T(1);
    if constexpr (GRAD) a = (T)(2.0 * rl.scale) * (rl.g_dev ? *static_cast<const T*>(rl.g_dev) : T(1));
    double acc = 0.0;
    for (int f = (int)fb.by; f < nframes && live; f += (int)fg.gy) {
        const T* h = traj + (long)f * frame;
        const T* hn = h + frame;
        const Pack<T, VEC> cu = ld<T, VEC>(h + e), cv = ld<T, VEC>(h + g.ss + e);
        T lap[2][VEC];
        star<T, NDIM, VEC, +1>(h, Q, g, i0, i1, x0, e, cu, lap[0]);
        star<T, NDIM, VEC, +1>(h + g.ss, Q, g, i0, i1, x0, e, cv, lap[1]);
        T part = T(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const Pack<T, VEC> nx = ld<T, VEC>(hn + s * g.ss + e);
            const T* c = Q + P_W + 10 * s;
            Pack<T, VEC> o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const T hs = s == 0 ? cu.v[i] : cv.v[i];
                const T rhs = Q[P_COEF + s] * lap[s][i] + poly_r(c, cu.v[i], cv.v[i]);
                const T r = rhs - (nx.v[i] - hs) / dt;                     // the residual, as pi_residual_kernel forms it
                const T w = (rl.weighted && x0 + i == 0) ? wrow * T(2) : wrow;
                if constexpr (GRAD) o.v[i] = (a * w) * r;
                else part = fma_(w * r, r, part);
            }
            if constexpr (GRAD) st<T, VEC>(G + (long)f * frame + s * g.ss + e, o);
        }
        acc += (double)part;
    }
    if constexpr (!GRAD) {
        acc = wave_sum_to_last(acc);
        if (threadIdx.x % WAVE == REDUCE_LANE) red[threadIdx.x / WAVE] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < 256 / WAVE; ++w) t += red[w];
            partials[(long)fb.by * fg.gx + fb.bx] = t;
        }
    }
}

// (dR_f/dh_f)^T G  =  coef * LapT(G) + J_r(h_f)^T G + G / dt        (the -G/dt part w.r.t. h_{f+1} is pointwise)
// FULL: `out` has nout >= nframes + 1 frames and is written completely -- frame f < nframes as above minus G_{f-1} / dt
// (the pointwise part of step f - 1), frame nframes = -G_{nframes-1} / dt, later frames zero: dL/dtraj of a loss over
// R_0 .. R_{nframes-1} in ONE launch (was: zero-fill, this kernel, a full-trajectory division and a subtraction).
template <typename T, int NDIM, int VEC, bool FULL = false>
__global__ void __launch_bounds__(256)
pi_residual_adj_kernel(const T* __restrict__ traj, const T* __restrict__ G, T* __restrict__ out,
                       const T* __restrict__ Q, Geom g, FrameGrid fg, int nframes = 0)
{
    const FrameBlock fb = frame_block(fg);
    if (!fb.live) return;
    if constexpr (FULL) {
        const int f = (int)fb.by;
        if (f >= nframes) {
            const int cprF = g.W / VEC;
            const long cidF = (long)fb.bx * blockDim.x + threadIdx.x;
            if (cidF >= (long)g.rows * cprF) return;
            int j0, j1, y0;
            long eF;
            chunk_coords<NDIM>(g, cidF, cprF, VEC, j0, j1, y0, eF);
            const long frameF = 2 * g.ss;
            const T dtF = Q[P_DT];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                Pack<T, VEC> o;
                if (f == nframes) {
                    const Pack<T, VEC> gp = ld<T, VEC>(G + (long)(f - 1) * frameF + s * g.ss + eF);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o.v[i] = T(0) - gp.v[i] / dtF;
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) o.v[i] = T(0);
                }
                st<T, VEC>(out + (long)f * frameF + s * g.ss + eF, o);
            }
            return;
        }
    }
    const int cpr = g.W / VEC;
    const long nchunks = (long)g.rows * cpr;
    const long cid = (long)fb.bx * blockDim.x + threadIdx.x;
    if (cid >= nchunks) return;
    int i0, i1, x0;
    long e;
    chunk_coords<NDIM>(g, cid, cpr, VEC, i0, i1, x0, e);
    const long frame = 2 * g.ss;
    const T* h = traj + (long)fb.by * frame;
    const T* Gf = G + (long)fb.by * frame;
    const Pack<T, VEC> u = ld<T, VEC>(h + e), v = ld<T, VEC>(h + g.ss + e);
    const Pack<T, VEC> gu = ld<T, VEC>(Gf + e), gv = ld<T, VEC>(Gf + g.ss + e);
    T lg[2][VEC];
    star<T, NDIM, VEC, -1>(Gf, Q, g, i0, i1, x0, e, gu, lg[0]);
    star<T, NDIM, VEC, -1>(Gf + g.ss, Q, g, i0, i1, x0, e, gv, lg[1]);
    const T dt = Q[P_DT];
    Pack<T, VEC> ou, ov;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        T ruu, ruv, rvu, rvv;                            // d r_u / d(u,v), d r_v / d(u,v)
        poly_dr(Q + P_W, u.v[i], v.v[i], ruu, ruv);
        poly_dr(Q + P_W + 10, u.v[i], v.v[i], rvu, rvv);
        ou.v[i] = fma_(Q[P_COEF + 0], lg[0][i], fma_(gu.v[i], ruu, gv.v[i] * rvu)) + gu.v[i] / dt;
        ov.v[i] = fma_(Q[P_COEF + 1], lg[1][i], fma_(gu.v[i], ruv, gv.v[i] * rvv)) + gv.v[i] / dt;
    }
    if constexpr (FULL) {
        if (fb.by > 0) {
            const Pack<T, VEC> pu = ld<T, VEC>(Gf - frame + e), pv = ld<T, VEC>(Gf - frame + g.ss + e);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                ou.v[i] -= pu.v[i] / dt;
                ov.v[i] -= pv.v[i] / dt;
            }
        }
    }
    st<T, VEC>(out + (long)fb.by * frame + e, ou);
    st<T, VEC>(out + (long)fb.by * frame + g.ss + e, ov);
}

// ---------------------------------------------------------------------------------------------
// squared-error losses over a trajectory without a materialised dL/dtraj (LossInj, pi_device.h)
// ---------------------------------------------------------------------------------------------
// one frame of the loss gradient: out = a * (h - target) (target == nullptr: a * h).  The sweep starts from it (frame T has
// no later step that could inject it) and the fall-back paths materialise whole trajectories of it.  n = elements.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
pi_loss_grad_kernel(const T* __restrict__ h, const T* __restrict__ target, T* __restrict__ out, long n, LossInj l)
{
    const T a = loss_factor<T>(l);
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += (long)gridDim.x * blockDim.x * VEC) {
        const Pack<T, VEC> x = ld<T, VEC>(h + i);
        Pack<T, VEC> y = x;
        if (target) {
            const Pack<T, VEC> t = ld<T, VEC>(target + i);
#pragma unroll
            for (int k = 0; k < VEC; ++k) y.v[k] = x.v[k] - t.v[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) y.v[k] = a * y.v[k];
        st<T, VEC>(out + i, y);
    }
}

// sum_x (h - target)^2 over `n` consecutive elements (a run of whole frames), one partial per workgroup (double).
// Streaming: 4 (8 with a target) bytes per element, read once; four 16-byte loads in flight per lane and operand.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
pi_sqerr_kernel(const T* __restrict__ traj, const T* __restrict__ target, long n, double* __restrict__ partials)
{
    __shared__ double red[256 / WAVE];
    double acc = 0.0;
    const long nchunks = n / VEC, stride = (long)gridDim.x * blockDim.x;
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; c + 3 * stride < nchunks; c += 4 * stride) {
        Pack<T, VEC> x[4], t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = ld<T, VEC>(traj + (c + q * stride) * VEC);
        if (target) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = ld<T, VEC>(target + (c + q * stride) * VEC);
        }
        T part = T(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const T d = target ? x[q].v[k] - t[q].v[k] : x[q].v[k];
                part = fma_(d, d, part);
            }
        acc += (double)part;
    }
    for (; c < nchunks; c += stride) {
        const Pack<T, VEC> x = ld<T, VEC>(traj + c * VEC);
        T part = T(0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T d = target ? x.v[k] - target[c * VEC + k] : x.v[k];
            part = fma_(d, d, part);
        }
        acc += (double)part;
    }
    acc = wave_sum_to_last(acc);
    if (threadIdx.x % WAVE == REDUCE_LANE) red[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 256 / WAVE; ++w) s += red[w];
        partials[blockIdx.x] += s;                       // slots are zeroed by the host; runs beyond the 64th share slots (stream-ordered)
    }
}

// loss = scale * sum of the partials, written in the compute type (one wave, fixed order)
template <typename T>
__global__ void __launch_bounds__(64)
pi_sqerr_finish_kernel(const double* __restrict__ partials, int n, double scale, T* __restrict__ out)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < n; b += WAVE) s += partials[b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, WAVE);
    if (threadIdx.x == 0) out[0] = (T)(s * scale);
}

// param_grad[idx] += sum_b partials[b][idx]; one wave per parameter, fixed order -> deterministic
__global__ void __launch_bounds__(64)
pi_reduce_partials_kernel(const double* __restrict__ partials, int nblocks, int np, double* __restrict__ param_grad)
{
    const int idx = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += WAVE) s += partials[(long)b * np + idx];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, WAVE);
    if (threadIdx.x == 0) param_grad[idx] += s;
}

}  // namespace pi
