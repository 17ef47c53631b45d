// pi_brick3d.h -- "brick" step kernels for 3D grids (forward and adjoint): one launch = one time step, built around what the
// device timeline of the direct kernels showed at 128^3 (tools/ubench/step3d_probe.hip, DESIGN.md "Brick kernels"):
//   * the load phase is bound by the bytes REQUESTED through the vector L1 (~64 B/clk per CU), not by what reaches the
//     L2: a pass that asks for every stencil neighbour separately spends 2.3 us just issuing its loads;
//   * the compute phase is bound by VALU ISSUE (~2 ns per instruction and SIMD whatever its type), and two thirds of the
//     instructions of the generic direct kernels were integer overhead (per-lane 64-bit addresses, wraps, spilled SGPRs).
// So: a workgroup of 256 lanes owns a BRICK = 256 consecutive 16-byte chunks of RZ consecutive planes (whole rows at the
// usual widths; any width works, rows follow each other in memory);
//   * z neighbours: the lane's own chunk in planes i0-2 .. i0+RZ+1, RZ + 4 loads per species with SCALAR plane bases and
//     ONE 32-bit lane offset (no per-lane address arithmetic at all);
//   * y / x neighbours: LDS.  Per (plane, species) the brick's chunks plus two rows of chunks before and after them (the
//     periodic wrap of the plane resolved when they are staged) sit in one window of fixed stride, so every neighbour is a
//     ds_read at lane address + immediate; the four halo rows are fetched by whole waves (wave-uniform task -> scalar
//     base, lane = chunk), at most 2 * RZ requests per lane;
//   * ONE barrier per pass; stores with scalar bases.
// Requests per lane and pass: 2 * (RZ + 4) + <= 2 * RZ  (RZ = 2: 14-16, against 36 in the direct kernel), VALU
// instructions per pass roughly halved.  Arithmetic and its order are those of pi::star / pi_fwd_kernel / pi_bwd_kernel:
// results are bit-identical (tests/test_hip_parity.py::test_brick3d_bitwise).
#pragma once
#include "pi_device.h"
#include "pi_kernels.h"
#include "pi_peer.h"

namespace pi {

constexpr int BRICK_NT = 256;                        // lanes per workgroup = own chunks per plane of a brick (NT = 512: wide rows)
constexpr int BRICK_CPR_MAX = 64;                    // 256-lane bricks: rows of up to 64 chunks (W <= 256 float32 / 128 float64)
// 512-lane bricks take rows of up to 128 chunks (W <= 512 float32: 384^3 ran on the direct kernels at 0.41 of HBM, round 4)
__host__ __device__ constexpr int brick_cpr_max(int nt) { return nt >= 512 ? 2 * BRICK_CPR_MAX : BRICK_CPR_MAX; }
// chunks / bytes per LDS window (fixed stride -> immediate offsets): 8 KiB per (plane, species) with 256 lanes, 16 KiB with 512
__host__ __device__ constexpr int brick_wb(int nt) { return (nt + 4 * brick_cpr_max(nt)) * 16; }
constexpr int BRICK_WB = brick_wb(BRICK_NT);

struct BrickGeom {
    int n0, n1;            // planes this call computes, rows per plane
    int cpr, total;        // chunks per row, per plane
    int nrg;               // bricks per plane group = ceil(total / lanes per workgroup)
    unsigned nblk;         // nrg * ceil(n0 / RZ)
    int wrap0;             // 1: axis 0 periodic, 0: slab layout (two halo planes on either side of the computed range)
    long s0, ss, off;      // plane stride, species stride, first computed point (elements), as in Geom
    FastDiv dnrg, dcpr;
    int nseg, ntask;       // halo segments of 64 chunks per (plane, species) = ceil(4 cpr / 64); ntask = 2 * RZ * nseg
    FastDiv dnseg;
    // XCD regions: workgroup b runs on XCD b % 8 (private L2).  xny == 0: every XCD takes a contiguous range of bricks, i.e. a
    // range of plane groups -- its two halo plane groups on either side are fetched by two L2s (128^3: 25 % of an XCD's reads).
    // xny > 0: the 8 XCDs tile the (plane group, brick row) space as (8 / xny) x xny rectangles of xpg x xrg bricks: shorter in z,
    // split in y -- 128^3: 32 planes x half a plane each, halo share 18.75 %.  Pure placement; any map is correct.
    int xny, xpg, xrg;
    FastDiv dxrg;
    LossInj loss;          // adjoint kernel: what the injection pointer means (pi_device.h)
    int wt;                // 1: the output frame is stored write-through (sc1).  Plain stores leave the whole frame dirty in the
                           // L2s and the kernel boundary then waits for its write-back (16 MiB at 128^3: ~1.5 us of a 9.4 us
                           // step, tools/ubench/step3d_probe.hip); written through, that traffic overlaps the waves still
                           // loading / computing.  (The direct kernels of round 2 did not gain from it: their boundary was
                           // hidden behind a longer body.)
};

// what a lane of a brick knows: `eb`, `lo`, row-neighbour / x-halo LDS addresses are per lane, the rest is wave-uniform
// NT lanes per workgroup: 256, or 512 for rows of more than 32 chunks -- the halo is four ROWS, so a 256-lane brick of 64-chunk
// rows (W = 256 float32: the 32 x 256^2 slabs of the 8-GPU 256^3 problem) fetches as many halo chunks as it owns; 512 lanes
// halve that share
template <typename T, int RZ, int NT = BRICK_NT>
struct Brick {
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int NW = NT / 64;
    static constexpr int WB = brick_wb(NT);
    static constexpr int MH = (2 * RZ * (4 * brick_cpr_max(NT) / 64) + NW - 1) / NW;   // halo tasks per wave: 2 RZ nseg / NW waves, nseg <= 4 (8: wide rows)
    int i0, cb, nown;                                // first plane of the group; first own chunk of the plane; how many
    bool valid;
    unsigned eb;                                     // byte offset of the lane's chunk inside a plane
    unsigned lo, ym2, ym1, yp1, yp2, xl, xr;         // LDS byte addresses (window 0): own chunk, row neighbours, x-halo pairs
    Pack<T, VEC> hreg[MH];
    unsigned hoff[MH];

    // r: raw brick id with r % 8 = the XCD it runs on (blockIdx.x, + multiples of a grid that is a multiple of 8)
    __device__ __forceinline__ void locate(const BrickGeom& g, unsigned r, unsigned ngrid, unsigned lds_base)
    {
        unsigned pg, rg;
        if (g.xny > 0) {
            const unsigned x = r % NXCD, i = r / NXCD;
            const unsigned zx = x / (unsigned)g.xny, yx = x - zx * (unsigned)g.xny;
            unsigned pl = g.dxrg.div(i);
            const unsigned rl = i - pl * (unsigned)g.xrg;
            // the regions of the upper half walk their planes from the top down: both ends of the plane range are computed
            // FIRST -- in the slab layout those are the faces a fused put (pi_peer.h) is waiting for
            if (2u * zx >= (unsigned)(NXCD / g.xny)) pl = (unsigned)g.xpg - 1u - pl;
            pg = zx * (unsigned)g.xpg + pl;
            rg = yx * (unsigned)g.xrg + rl;
        } else {
            // contiguous range per XCD within the first `ngrid` ids, the same again for every further pass of a persistent grid
            const unsigned pass = r / ngrid;
            const unsigned vb = pass * ngrid + xcd_remap(r - pass * ngrid, min(ngrid, g.nblk - pass * ngrid));
            pg = g.dnrg.div(vb);
            rg = vb - pg * (unsigned)g.nrg;
        }
        i0 = (int)pg * RZ;
        cb = (int)rg * NT;
        nown = min(NT, g.total - cb);
        const int tid = (int)threadIdx.x;
        valid = tid < nown;
        const int own = min(tid, nown - 1);                            // idle lanes shadow the last chunk (never stored)
        const unsigned idx = (unsigned)(cb + own);
        eb = idx * 16u;
        const unsigned row = g.dcpr.div(idx), xc = idx - row * (unsigned)g.cpr;
        const unsigned Wb = (unsigned)g.cpr * 16u;
        lo = lds_base + (unsigned)(2 * g.cpr + own) * 16u;
        ym2 = lo - 2u * Wb; ym1 = lo - Wb; yp1 = lo + Wb; yp2 = lo + 2u * Wb;
        xl = xc > 0u ? lo - 2u * (unsigned)sizeof(T) : lo + Wb - 2u * (unsigned)sizeof(T);
        xr = xc + 1u < (unsigned)g.cpr ? lo + 16u : lo + 16u - Wb;
    }

    // the four halo rows of every (plane, species) of the brick: whole waves take 64-chunk segments
    __device__ __forceinline__ void request_halo(const T* __restrict__ f, const BrickGeom& g, unsigned lds_base)
    {
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
        const int cpr = g.cpr;
#pragma unroll
        for (int m = 0; m < MH; ++m) {
            const int task = wave + NW * m;
            hoff[m] = ~0u;
            if (task < g.ntask) {                                      // wave-uniform
                const int c = (int)g.dnseg.div((unsigned)task), sg = task - c * g.nseg;    // c = 2 * plane + species
                int hidx = sg * 64 + lane;
                const bool ok = hidx < 4 * cpr;
                hidx = min(hidx, 4 * cpr - 1);
                const int wp = hidx + (hidx >= 2 * cpr ? nown : 0);    // chunk of the window
                int pc = cb - 2 * cpr + wp;                            // chunk of the plane, periodic
                pc += pc < 0 ? g.total : (pc >= g.total ? -g.total : 0);
                const int pl = min(i0 + (c >> 1), g.n0 - 1);           // partial last group: stay inside the field
                const char* base = sgpr_ptr(reinterpret_cast<const char*>(f + ((c & 1) ? g.ss : 0L) + g.off + (long)pl * g.s0));
                hreg[m] = ldb<T, VEC>(base, (unsigned)pc * 16u);
                hoff[m] = ok ? lds_base + (unsigned)c * (unsigned)WB + (unsigned)wp * 16u : ~0u;
            }
        }
    }

    __device__ __forceinline__ void commit(unsigned char* smem, const PlaneWindow<T, VEC, RZ> (&win)[2]) const
    {
#pragma unroll
        for (int j = 0; j < RZ; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                *reinterpret_cast<Pack<T, VEC>*>(smem + lo + (2 * j + s) * WB) = win[s].w[j + 2];
#pragma unroll
        for (int m = 0; m < MH; ++m)
            if (hoff[m] != ~0u) *reinterpret_cast<Pack<T, VEC>*>(smem + hoff[m]) = hreg[m];
    }

    // rows, then the fastest axis, onto `lap` (pi::star2_inplane's order); window c = 2 * plane + species
    template <int FLIP>
    __device__ __forceinline__ void inplane(const unsigned char* smem, int c, const T* __restrict__ P,
                                            const Pack<T, VEC>& ctr, T (&lap)[VEC]) const
    {
        const unsigned yo[4] = {FLIP > 0 ? ym2 : yp2, FLIP > 0 ? ym1 : yp1, FLIP > 0 ? yp1 : ym1, FLIP > 0 ? yp2 : ym2};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const Pack<T, VEC> nb = *reinterpret_cast<const Pack<T, VEC>*>(smem + yo[t] + c * WB);
            const T w = P[P_TAPS + 4 + t];
#pragma unroll
            for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, nb.v[i], lap[i]);
        }
        T win[VEC + 4];
        const Pack<T, 2> l = *reinterpret_cast<const Pack<T, 2>*>(smem + xl + c * WB);
        const Pack<T, 2> r = *reinterpret_cast<const Pack<T, 2>*>(smem + xr + c * WB);
        win[0] = l.v[0]; win[1] = l.v[1];
        win[VEC + 2] = r.v[0]; win[VEC + 3] = r.v[1];
        Pack<T, VEC> cc = ctr;
        keep_in_regs(cc);
#pragma unroll
        for (int i = 0; i < VEC; ++i) win[2 + i] = cc.v[i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = FLIP * (t < 2 ? t - 2 : t - 1);
            const T w = P[P_TAPS + 8 + t];
#pragma unroll
            for (int i = 0; i < VEC; ++i) lap[i] = fma_(w, win[2 + i + k], lap[i]);
        }
    }
};

// the window of plane neighbours reads Geom; the brick kernels carry the few fields it needs
__device__ __forceinline__ Geom brick_as_geom(const BrickGeom& b)
{
    Geom g;
    g.n0 = b.n0; g.wrap0 = b.wrap0; g.s0 = b.s0;
    return g;
}

// ---------------------------------------------------------------------------------------------
// forward: out = h + dt * (coef * Lap(h) + react(h))      (one brick per workgroup: gridDim.x == g.nblk)
// ---------------------------------------------------------------------------------------------
template <typename T, int HC, int RZ, int NT = BRICK_NT>
__global__ void __launch_bounds__(NT)
pi_fwd3d_brick_kernel(const T* __restrict__ h, T* __restrict__ out, const T* __restrict__ P, BrickGeom g, int hc_rt,
                      PeerPutFused put)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // slab layout, peer-mailbox ring: the lowest `put.nput` workgroups carry the faces of the frame being written into the
    // neighbours' mailboxes as soon as the bricks that hold them have stored them (pi_peer.h "put fused into the step kernel")
    if (put.nput && (int)blockIdx.x < put.nput) {
        if (put.vec16) peer_put_block<true>(put, (int)blockIdx.x);
        else peer_put_block<false>(put, (int)blockIdx.x);
        return;
    }
    const int hc = HC > 0 ? HC : hc_rt;
    PI_STAMP3(0);
    Brick<T, RZ, NT> B;
    B.locate(g, blockIdx.x - (unsigned)put.nput, gridDim.x - (unsigned)put.nput, 0u);
    Lane L;
    L.i0 = B.i0; L.eb = B.eb;
    Geom gg = brick_as_geom(g);
    const T* hs[2] = {h + g.off, h + g.ss + g.off};
    PlaneWindow<T, VEC, RZ> win[2];
    win[0].load(hs[0], gg, L);
    win[1].load(hs[1], gg, L);
    B.request_halo(h, g, 0u);
    PI_STAMP3(1);
    B.commit(smem_raw, win);
    PI_STAMP3(2);
    lds_barrier();
    PI_STAMP3(3);
    const T dt = P[P_DT];
#pragma unroll
    for (int j = 0; j < RZ; ++j) {
        const int iz = B.i0 + j;
        if (iz >= g.n0) break;                               // partial last plane group (block-uniform)
        const Pack<T, VEC> cu = win[0].w[j + 2], cv = win[1].w[j + 2];
        T lap[2][VEC];
        win[0].template planes<+1>(j, P, lap[0]);
        win[1].template planes<+1>(j, P, lap[1]);
        B.template inplane<+1>(smem_raw, 2 * j, P, cu, lap[0]);
        B.template inplane<+1>(smem_raw, 2 * j + 1, P, cv, lap[1]);
#pragma clang loop unroll(disable)
        for (int s = 0; s < 2; ++s) {
            T rr[VEC];
#if defined(PI_BRICK_SKELETON)      // access-pattern ceiling (tools/gpu_brick_skeleton.sh; WRONG results): loads, LDS traffic, stencil
            if constexpr (true) {   // taps and stores of the step, without the reaction term
#pragma unroll
                for (int i = 0; i < VEC; ++i) rr[i] = T(0);
            } else
#endif
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
                const T res = coef * lp + rr[i];            // two roundings (train_3drd.py:133-134)
                const T inc = res * dt;
                o.v[i] = hv + inc;
            }
            char* po = const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(out + s * g.ss + g.off + (long)iz * g.s0)));
            if (B.valid) {
                if (g.wt) stb_wt<T, VEC>(po, B.eb, o);      // wave-uniform choice
                else stb<T, VEC>(po, B.eb, o);
            }
        }
        PI_STAMP3(4 + (j > 0));
    }
    if (put.nput) peer_face_stored(put, B.i0, min(B.i0 + RZ, g.n0));
    PI_STAMP3(7);
}

// ---------------------------------------------------------------------------------------------
// physics-residual LOSS of a 3D trajectory on the brick machinery (pi_residual_sq_kernel<GRAD = false> is the generic flavour;
// reference: loss_gen / get_phy_Loss, train_3drd.py:287-346): per frame f the residual
//   R_s = coef_s * Lap(h_f)_s + r_s(h_f) - (h_{f+1,s} - h_{f,s}) / dt          (Q: pre-contracted block of the TRUE equation)
// is formed exactly as the forward brick kernel forms `res` (same taps, same order: bit-identical to the generic kernel's R) and
// w * R^2 is summed per workgroup in double; workgroup (x, y) walks frames y, y + gridDim.y, ... of brick x.
// The generic kernel asks the vector L1 for 24 sixteen-byte pieces per chunk and frame (12x the bytes that reach HBM): 128^3,
// 200 frames 2.7 ms = 0.15 of HBM; here the plane neighbours come from the register window and the in-plane ones from LDS.
// ---------------------------------------------------------------------------------------------
template <typename T, int RZ, int NT = BRICK_NT>
__global__ void __launch_bounds__(NT)
pi_res3d_brick_kernel(const T* __restrict__ traj, double* __restrict__ partials, const T* __restrict__ Q, BrickGeom g,
                      long frame_stride, int nframes, int weighted)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[NT / WAVE];
    Brick<T, RZ, NT> B;
    B.locate(g, blockIdx.x, gridDim.x, 0u);
    Lane L;
    L.i0 = B.i0; L.eb = B.eb;
    Geom gg = brick_as_geom(g);
    // weights of the reference's padded evaluation grid: index 0 of every axis counts twice
    const unsigned idx = (unsigned)(B.cb + min((int)threadIdx.x, B.nown - 1));
    const unsigned row = g.dcpr.div(idx), xc = idx - row * (unsigned)g.cpr;
    const T wyx = weighted ? T(row == 0u ? 2 : 1) : T(1);
    const T dt = Q[P_DT];
    double acc = 0.0;
    for (int f = (int)blockIdx.y; f < nframes; f += (int)gridDim.y) {
        const T* h = traj + (long)f * frame_stride;
        const T* hs[2] = {h + g.off, h + g.ss + g.off};
        PlaneWindow<T, VEC, RZ> win[2];
        win[0].load(hs[0], gg, L);
        win[1].load(hs[1], gg, L);
        B.request_halo(h, g, 0u);
        Pack<T, VEC> nx[RZ][2];                              // the same chunk of frame f + 1: requested with the window
#pragma unroll
        for (int j = 0; j < RZ; ++j) {
            const int iz = min(B.i0 + j, g.n0 - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                nx[j][s] = ldb<T, VEC>(sgpr_ptr(reinterpret_cast<const char*>(hs[s] + frame_stride + (long)iz * g.s0)), B.eb);
        }
        B.commit(smem_raw, win);
        lds_barrier();
        T part = T(0);
#pragma unroll
        for (int j = 0; j < RZ; ++j) {
            const int iz = B.i0 + j;
            if (iz >= g.n0) break;                           // partial last plane group (block-uniform)
            const Pack<T, VEC> cu = win[0].w[j + 2], cv = win[1].w[j + 2];
            T lap[2][VEC];
            win[0].template planes<+1>(j, Q, lap[0]);
            win[1].template planes<+1>(j, Q, lap[1]);
            B.template inplane<+1>(smem_raw, 2 * j, Q, cu, lap[0]);
            B.template inplane<+1>(smem_raw, 2 * j + 1, Q, cv, lap[1]);
            const T wz = (weighted && iz == 0) ? wyx * T(2) : wyx;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const T* c = Q + P_W + 10 * s;
                const T coef = Q[P_COEF + s];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const T hv = s == 0 ? cu.v[i] : cv.v[i];
                    const T rhs = coef * lap[s][i] + poly_r(c, cu.v[i], cv.v[i]);
                    const T r = rhs - (nx[j][s].v[i] - hv) / dt;
                    const T w = (weighted && xc == 0u && i == 0) ? wz * T(2) : wz;
                    part = fma_(w * r, r, part);
                }
            }
        }
        if (B.valid) acc += (double)part;
        lds_barrier();                                       // the next frame's window overwrites this one
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

// ---------------------------------------------------------------------------------------------
// adjoint of one step (see pi_bwd_kernel): Gp = G + coef*dt*LapT(G) + dt*J_react(h)^T G (+ inj), diffusion-coefficient
// sums always, and -- MOM, pre-contracted blocks -- the 20 coefficient moments carried per lane over all bricks of the
// workgroup and reduced once per launch.  partials: one row of np doubles per workgroup (owner-block read-modify-write).
// ---------------------------------------------------------------------------------------------
// LOSS: the injected frame is a squared-error loss gradient formed here (pi::LossInj modes 1 / 2) instead of a materialised
// dL/dout (a template parameter, not a run-time switch: the kernel sits at the edge of its 128-register budget and of the 102
// SGPRs -- the generic form spilled)
// (LOSS = pi::LossInj::mode, 0 / 1 / 2)
// PUT: the slab sweep's flavour that carries the faces of the adjoint frame it writes into the neighbours' mailboxes itself
// (pi_peer.h "put fused into the step kernel"; the lowest put.nput block ids).  A template parameter, and the argument an empty
// struct otherwise: as a plain extra argument it pushed three flavours of this kernel, which sits at its 128-VGPR / 102-SGPR
// edge, into scratch.  Through the rank's OWN mailbox it buys little (a put to self is 2 MiB of uncached stores into the same
// HBM the sweep streams from); across xGMI the same bytes are ~16 us of wire per step that would otherwise sit between two
// sweep launches.
template <typename T> using BV2 = T __attribute__((ext_vector_type(2)));
template <typename T, bool PAIR> struct BrickMomAcc {
    using type = T;
    static __device__ __forceinline__ T total(T a) { return a; }
};
template <typename T> struct BrickMomAcc<T, true> {
    using type = BV2<T>;
    static __device__ __forceinline__ T total(BV2<T> a) { return a.x + a.y; }
};
#ifndef PI_BRICK_MOM_PAIR_MIN_RZ
#define PI_BRICK_MOM_PAIR_MIN_RZ 2     // one-plane bricks hold 128 registers for four waves per SIMD: 20 more spill (24.5 vs 17.7 us at 128^3)
#endif
struct NoPut {};
template <bool PUT> struct AdjPutArg { using type = NoPut; };
template <> struct AdjPutArg<true> { using type = PeerPutFused; };

template <typename T, int HC, int RZ, bool MOM, int LOSS = 0, int NT = BRICK_NT, bool PUT = false>
#ifndef PI_ADJ_RZ2_WAVES
#define PI_ADJ_RZ2_WAVES 2
#endif
__global__ void __launch_bounds__(NT, (RZ == 1 && HC == POLY && LOSS != 2 && sizeof(T) == 4 && !PUT) ? 4 : (RZ == 2 && HC == POLY && LOSS == 0 && sizeof(T) == 4 && NT == 256 && !PUT ? PI_ADJ_RZ2_WAVES : 2))   // one-plane bricks of pre-contracted float32
pi_adj3d_brick_kernel(const T* __restrict__ h, const T* __restrict__ G, const T* __restrict__ inj, T* __restrict__ Gp,
                      double* __restrict__ partials, const T* __restrict__ P, BrickGeom g, int hc_rt,
                      typename AdjPutArg<PUT>::type put)
{
    static_assert(!MOM || HC == POLY, "fused moments are those of the pre-contracted block");
    unsigned bid = blockIdx.x, nwg = gridDim.x;              // this workgroup among the brick workgroups
    if constexpr (PUT) {
        if ((int)blockIdx.x < put.nput) {
            if (put.vec16) peer_put_block<true>(put, (int)blockIdx.x);
            else peer_put_block<false>(put, (int)blockIdx.x);
            return;
        }
        bid -= (unsigned)put.nput;
        nwg -= (unsigned)put.nput;
    }
    constexpr int VEC = 16 / (int)sizeof(T), NW = NT / WAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int hc = HC == POLY ? 0 : (HC > 0 ? HC : hc_rt);
    const int np = nparams(hc);
    // LDS: [NW][2] coefficient sums (double) | windows, overlaid after the last pass by the moment transpose scratch
    double* redc = reinterpret_cast<double*>(smem_raw);
    constexpr unsigned WIN0 = NW * 2 * sizeof(double);
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    // this workgroup's partial row: requested now, needed at the very end
    auto slot_of = [&](int k) { return k < 2 ? P_COEF + k : P_W + k - 2; };
    double* const prow = partials + (long)bid * np;
    const int nsum = MOM ? 22 : 2;
    const double pold = (int)threadIdx.x < nsum ? prow[slot_of((int)threadIdx.x)] : 0.0;

    const T dt = P[P_DT];
    double lane_c[2] = {0.0, 0.0};
    // moment accumulators.  PAIR (float32, bricks of two or more planes): 2-vectors over the point pairs (0,1) / (2,3) of the
    // lane's chunk -- operands that sit in adjacent registers as loaded, so the sums issue as v_pk_fma_f32 without the
    // v_mov_b32 shuffles hipcc's own packing of the scalar form paid (round 5: 314 of 1037 VALU instructions of the
    // two-plane kernel were moves; profiles/r05_brick_adjoint_packed_moments.txt)
    constexpr bool PAIR = MOM && sizeof(T) == 4 && VEC == 4 && RZ >= PI_BRICK_MOM_PAIR_MIN_RZ;
    using MAcc = typename BrickMomAcc<T, PAIR>::type;
    MAcc macc[MOM ? 2 : 1][MOM ? 10 : 1];
    if constexpr (MOM) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) macc[s][m] = MAcc{};
    }
    Geom gg = brick_as_geom(g);
    bool staged = false;
    PI_STAMP3(0);
    for (unsigned vb = bid; vb < g.nblk; vb += nwg) {
        Brick<T, RZ, NT> B;
        B.locate(g, vb, nwg, WIN0);
        Lane L;
        L.i0 = B.i0; L.eb = B.eb;
        PlaneWindow<T, VEC, RZ> win[2];
        win[0].load(G + g.off, gg, L);
        win[1].load(G + g.ss + g.off, gg, L);
        B.request_halo(G, g, WIN0);
        // pointwise operands of all planes of the pass: requested with the window, before the barrier (HBM-cold)
        Pack<T, VEC> hu[RZ], hv[RZ], ju[RZ], jv[RZ];
#pragma unroll
        for (int j = 0; j < RZ; ++j) {
            const int iz = min(B.i0 + j, g.n0 - 1);
            hu[j] = ldb<T, VEC>(sgpr_ptr(reinterpret_cast<const char*>(h + g.off + (long)iz * g.s0)), B.eb);
            hv[j] = ldb<T, VEC>(sgpr_ptr(reinterpret_cast<const char*>(h + g.ss + g.off + (long)iz * g.s0)), B.eb);
            if (LOSS != 1 && inj) {                          // LOSS mode 1: a function of the state, nothing more to read
                ju[j] = ldb<T, VEC>(sgpr_ptr(reinterpret_cast<const char*>(inj + g.off + (long)iz * g.s0)), B.eb);
                jv[j] = ldb<T, VEC>(sgpr_ptr(reinterpret_cast<const char*>(inj + g.ss + g.off + (long)iz * g.s0)), B.eb);
            }
        }
        PI_STAMP3(1);
        if (staged) lds_barrier();                           // a further brick of this workgroup: the last one's reads are done
        B.commit(smem_raw, win);
        PI_STAMP3(2);
        lds_barrier();
        PI_STAMP3(3);
        staged = true;
        const T live = B.valid ? T(1) : T(0);
#pragma unroll
        for (int j = 0; j < RZ; ++j) {
            const int iz = B.i0 + j;
            if (iz >= g.n0) break;
            const Pack<T, VEC> u = hu[j], v = hv[j];
            Pack<T, VEC> gc[2] = {win[0].w[j + 2], win[1].w[j + 2]};
            T dl[2][VEC];
            win[0].template planes<-1>(j, P, dl[0]);
            win[1].template planes<-1>(j, P, dl[1]);
            B.template inplane<-1>(smem_raw, 2 * j, P, gc[0], dl[0]);
            B.template inplane<-1>(smem_raw, 2 * j + 1, P, gc[1], dl[1]);
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
#if defined(PI_BRICK_SKELETON)      // access-pattern ceiling: no Jacobian, no moments, no coefficient sums (the operands are
            if constexpr (true) {   // still loaded and consumed by one add each)
#pragma unroll
                for (int i = 0; i < VEC; ++i) { du[i] = u.v[i]; dv[i] = v.v[i]; }
            } else
#endif
            if constexpr (HC == POLY) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const T* c = P + P_W + 10 * s;
                    const Pack<T, VEC>& hs = s == 0 ? u : v;
                    double acc_c = 0.0;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        const T gr = gc[s].v[i] * dt;
                        acc_c += (double)(dl[s][i] * hs.v[i]);
                        T ru, rv;
                        poly_dr(c, u.v[i], v.v[i], ru, rv);
                        du[i] = fma_(gr, ru, du[i]);
                        dv[i] = fma_(gr, rv, dv[i]);
                        if constexpr (MOM && !PAIR) {
                            MAcc (&acc)[10] = macc[s];
                            const T uu = u.v[i], vv = v.v[i];
                            const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
                            acc[0] += gr;
                            acc[1] = fma_(gr, uu, acc[1]); acc[2] = fma_(gr, vv, acc[2]);
                            acc[3] = fma_(gr, u2, acc[3]); acc[4] = fma_(gr, uv, acc[4]); acc[5] = fma_(gr, v2, acc[5]);
                            acc[6] = fma_(gr, u2 * uu, acc[6]); acc[7] = fma_(gr, u2 * vv, acc[7]);
                            acc[8] = fma_(gr, uu * v2, acc[8]); acc[9] = fma_(gr, v2 * vv, acc[9]);
                        }
                    }
                    lane_c[s] += acc_c;
                }
                if constexpr (PAIR) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const BV2<T> uu{u.v[2 * hh], u.v[2 * hh + 1]}, vv{v.v[2 * hh], v.v[2 * hh + 1]};
                        const BV2<T> u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
                        const BV2<T> u3 = u2 * uu, u2v = u2 * vv, uv2 = uu * v2, v3 = v2 * vv;
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const BV2<T> gr = BV2<T>{gc[s].v[2 * hh], gc[s].v[2 * hh + 1]} * BV2<T>{dt, dt};
                            MAcc (&acc)[10] = macc[s];
                            acc[0] += gr;
                            acc[1] = __builtin_elementwise_fma(gr, uu, acc[1]); acc[2] = __builtin_elementwise_fma(gr, vv, acc[2]);
                            acc[3] = __builtin_elementwise_fma(gr, u2, acc[3]); acc[4] = __builtin_elementwise_fma(gr, uv, acc[4]);
                            acc[5] = __builtin_elementwise_fma(gr, v2, acc[5]);
                            acc[6] = __builtin_elementwise_fma(gr, u3, acc[6]); acc[7] = __builtin_elementwise_fma(gr, u2v, acc[7]);
                            acc[8] = __builtin_elementwise_fma(gr, uv2, acc[8]); acc[9] = __builtin_elementwise_fma(gr, v3, acc[9]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const T* W = P + P_W + s * species_block(hc);
                    const Pack<T, VEC>& hs = s == 0 ? u : v;
                    T gr[VEC];
                    double acc_c = 0.0;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        gr[i] = gc[s].v[i] * dt;
                        acc_c += (double)(dl[s][i] * hs.v[i]);
                    }
                    lane_c[s] += acc_c;
                    W10<T> nx = load_w10(W);
#pragma clang loop unroll(disable)
                    for (int jj = 0; jj < hc; ++jj) {
                        const W10<T> c = nx;
                        if (jj + 1 < hc) nx = load_w10(W + 10 * (jj + 1));
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const T a1 = fma_(c.w[0], u.v[i], fma_(c.w[1], v.v[i], c.w[2]));
                            const T a2 = fma_(c.w[3], u.v[i], fma_(c.w[4], v.v[i], c.w[5]));
                            const T a3 = fma_(c.w[6], u.v[i], fma_(c.w[7], v.v[i], c.w[8]));
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
                ou.v[i] = gc[0].v[i] + tu;
                ov.v[i] = gc[1].v[i] + tv;
            }
            if (inj) {
                if constexpr (LOSS != 0) {
                    const T la = loss_factor<T>(g.loss);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        ou.v[i] += la * (LOSS == 2 ? u.v[i] - ju[j].v[i] : u.v[i]);
                        ov.v[i] += la * (LOSS == 2 ? v.v[i] - jv[j].v[i] : v.v[i]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) { ou.v[i] += ju[j].v[i]; ov.v[i] += jv[j].v[i]; }
                }
            }
            if (B.valid) {
                char* pu = const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(Gp + g.off + (long)iz * g.s0)));
                char* pv = const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(Gp + g.ss + g.off + (long)iz * g.s0)));
                if (g.wt) { stb_wt<T, VEC>(pu, B.eb, ou); stb_wt<T, VEC>(pv, B.eb, ov); }
                else { stb<T, VEC>(pu, B.eb, ou); stb<T, VEC>(pv, B.eb, ov); }
            }
            PI_STAMP3(4 + (j > 0));
        }
        if constexpr (PUT) {                                 // block-uniform: only bricks that hold face planes count themselves
            if (peer_holds_face(put, B.i0, min(B.i0 + RZ, g.n0))) peer_face_stored(put, B.i0, min(B.i0 + RZ, g.n0));
        }
    }
    PI_STAMP3(6);

    // ---- per-launch reductions --------------------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double r = wave_sum_to_last(lane_c[s]);
        if (lane == REDUCE_LANE) redc[wave * 2 + s] = r;
    }
    T* mred = reinterpret_cast<T*>(smem_raw + WIN0);                      // [20] block sums of the moments, then scratch
    if constexpr (MOM) {
        // LDS transpose (see pi_bwd_kernel): every thread writes its 20 values, 8 lanes per moment add NT / 8 of them
        constexpr int RS = NT + 8;
        T* scr = mred + 32;
        lds_barrier();                                                   // the scratch overlays the windows
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) scr[(10 * s + m) * RS + (int)threadIdx.x] = BrickMomAcc<T, PAIR>::total(macc[s][m]);
        __syncthreads();
        {
            const int task = (int)threadIdx.x;                           // NT = 256 >= 160 tasks: one trip
            const int mm = min(task, 159) >> 3, part = task & 7;
            T a = T(0);
            if (task < 160) {
                T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
                const T* row = scr + mm * RS + part;
#pragma unroll
                for (int k = 0; k < NT; k += 64) {
                    const T v0 = row[k], v1 = row[k + 8], v2 = row[k + 16], v3 = row[k + 24];
                    const T v4 = row[k + 32], v5 = row[k + 40], v6 = row[k + 48], v7 = row[k + 56];
                    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
                    a0 += v4; a1 += v5; a2 += v6; a3 += v7;
                }
                a = (a0 + a1) + (a2 + a3);
            }
            a += dpp_mov<0x111, 0xF>(a);
            a += dpp_mov<0x112, 0xF>(a);
            a += dpp_mov<0x114, 0xF>(a);
            if (task < 160 && part == 7) mred[mm] = a;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nsum) {
        double s = 0.0;
        if (threadIdx.x < 2) {
#pragma unroll
            for (int w = 0; w < NW; ++w) s += redc[w * 2 + threadIdx.x];
        } else {
            s = (double)mred[threadIdx.x - 2];
        }
        prow[slot_of((int)threadIdx.x)] = pold + s;
    }
    PI_STAMP3(7);
}

}  // namespace pi
