// pi_res3d.h -- RESIDENT 3D rollouts (round 6): the whole T-step forward rollout / reverse sweep of a 3D grid in ONE launch,
// the state kept in LDS for the whole rollout.
//
// Why (DESIGN.md "3D resident"): at 128^3 the launch-per-step brick kernels sat at 0.48 of HBM for three rounds,
// within 6-8 % of their own access pattern: the DESIGN was the limit -- every step re-reads the 16 MiB state through the vector
// L1 (3.5x its size in requests) and pays a ~1.9 us kernel boundary on an 8.3 us step.  But that state is 16 MiB and the chip
// has 256 CUs x 160 KB of LDS = 40 MB: it FITS ON CHIP.  One resident workgroup per CU owns a block of BZ x BY x BX = 16 x 16 x
// 32 points (both species: 64 KB) inside an LDS window with a two-point halo on every side; per step it computes the block
// from LDS, stores the frame (forward: the only HBM traffic left, write-only) and hands the two-deep faces of its block to its
// six neighbours as DATA-TAGGED GRANULES through a per-block outbox in device memory (the protocol of the resident 2D kernels,
// pi_tile2d.h, double-buffered by step parity: a block publishes step e + 2 only after it gathered e + 1 from every neighbour,
// which they published after THEIR gather of e -- the last read of the slot about to be overwritten; a star stencil has face
// neighbours only).
//
// What the first versions measured (tools/res3d_dev.hip, profiles/r06_resident3d_*.txt) and what the design answers:
//  * a CU moves only ~11-13 bytes per clock to or from anything beyond its XCD's L2, and a 16-byte {u, tag, v, tag} granule per
//    halo point is 160 KB per CU and step on top of the 64 KB frame: 11.5 us per step, all of it queueing.  Hence
//    (a) XCD REGIONS: workgroup b runs on XCD b % 8, so it takes a block of region b % 8 (2 x 2 x 2 regions of 4 x 4 x 2 blocks at
//        128^3): 70 % of all faces have both sides under ONE L2 and are published with plain stores (the line stays in the L2;
//        the reader's sc1 load is served from it), only region faces are written through;
//    (b) ONE-BIT TAGS: a granule carries TWO points {u0, v0', u1, v1'}; each 8-byte half validates itself through the lowest
//        mantissa bit of v', which holds the step's tag bit (a slot only ever holds step e or e - 2, one bit tells them apart);
//        the 2 x 64 displaced bits of a wave's store travel as ballots in one small mask granule per wave and step with full
//        32-bit tags.  8.25 bytes per point instead of 16, 5 + 1 requests per lane and hand-over instead of 10, values
//        bit-exact after the gather puts the bits back;
//  * publishing strip by strip from registers issued 36 sparse store instructions per wave (a face is a few lanes of many
//    waves) and each costs the memory pipeline a full slot: the shell phase took 3.6 us instead of 2.2.  Now the faces are
//    published DENSELY from the LDS window after the write-back: lane r stores granule r, five stores per lane, all lanes;
//  * hipcc (ROCm 7.2) emits no wait state between a buffer_store_dwordx4 whose soffset is an SGPR and a rewrite of its data
//    registers (GCNHazardRecognizer::createsVALUHazard exempts that form), gfx950 needs one: the next granule's v_mov landed in
//    the store's dword 0 for the upper lanes.  Every granule store here has the constant 0 as soffset (the compiler then
//    inserts the s_nop) and the parity / block offset inside the buffer descriptor.
// Per step:  interior strips (768 of 2048: no halo needed) | halo lands | the other strips | write back + frame store |
// publish + request -- the hand-over's round trip sits under the interior strips of the NEXT step.
// Arithmetic and its order are pi::star's / pi_fwd3d_brick_kernel's: every frame is bit-identical to the brick kernels'.
#pragma once
#include "pi_device.h"

namespace pi {
namespace r3d {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int BZ = 16, BY = 16, BX = 32, XS = BX / 4;
// LDS window: [2 species][BZ + 4 planes][BY + 4 rows][ROWB bytes]; a row = x = -2, -1 | own 32 floats | x = 32, 33 with no padding,
// the window itself 8 bytes into the allocation: own strips are 16-byte aligned and every stencil neighbour of every point is
// lane address + immediate.  Behind it the STAGING area: the faces of the state being computed, in outbox order.
constexpr int ROWB = 144, OWN0 = 8, YS = ROWB, ZS = (BY + 4) * ROWB, SP = (BZ + 4) * ZS;
constexpr int STG0 = (8 + 2 * SP + 15) / 16 * 16;                       // 115216
constexpr int NI = (BZ - 4) * (BY - 4) * (XS - 2);                      // 864 strips that need no halo
constexpr int NSTRIP = BZ * BY * XS, NSH = NSTRIP - NI;                 // 2048 strips, 1184 that need the halo
// NT lanes per workgroup (512: two waves per SIMD, four strips per lane and step; 1024: four waves, two strips).
// Phase S (needs the halo; its faces are what the neighbours wait for) = the 1184 shell strips + the last 96 interior strips =
// five wave-slots per SIMD; phase I = the first 768 interior strips = three per SIMD, with the hand-over travelling under it.
constexpr int NS = NSTRIP - 768;                                        // phase S: 1280 strips = five wave-slots per SIMD
static_assert(NS >= NSH && NS % WAVE == 0, "task schedule");
template <int NT> struct Shape {
    static constexpr int NW = NT / WAVE, SLOTS = NSTRIP / NT;
    static constexpr int NQ = (2560 + NT - 1) / NT;                     // granule stores / loads per lane and hand-over
    static constexpr int NMASK = NW * NQ * 2;                           // mask granule (wave, q, point) = F_MASK + (wave * NQ + q) * 2 + point
    static constexpr int BOX_GRAN = 2560 + NMASK, BOX_BYTES = BOX_GRAN * 16;
    static_assert(NT == 256 || NT == 512 || NT == 1024, "one, two or four waves per SIMD");
};
// a block's outbox per parity: six face buffers of 16-byte granules (two x-adjacent points each), then the mask granules.
// z faces [2 planes][BY][BX / 2], y faces [BZ][2 rows][BX / 2], x faces [BZ][BY] (the two columns of a face are one granule)
constexpr int GZ = 2 * BY * BX / 2, GY = 2 * BZ * BX / 2, GX = BZ * BY;
constexpr int F_ZLO = 0, F_ZHI = GZ, F_YLO = 2 * GZ, F_YHI = 2 * GZ + GY, F_XLO = 2 * GZ + 2 * GY, F_XHI = F_XLO + GX;
constexpr int NGRAN = 2 * GZ + 2 * GY + 2 * GX;                         // 2560 data granules
constexpr int F_MASK = NGRAN;
constexpr int LDS_BYTES = STG0 + NGRAN * 16 + 64 + 16 * 22 * 8;           // 159056 of the CU's 163840 (abort word, the adjoint's sums)
static_assert(NGRAN == 2560 && GZ % WAVE == 0 && GX % WAVE == 0, "the lane map of the hand-over assumes these face sizes");
// granule g of a box (0 <= g < NGRAN) -> face, granule inside the face
__device__ __forceinline__ int face_of(int g) { return g < F_YLO ? g / GZ : (g < F_XLO ? 2 + (g - F_YLO) / GY : 4 + (g - F_XLO) / GX); }
__device__ __forceinline__ int face_base(int f) { return f < 2 ? f * GZ : (f < 4 ? F_YLO + (f - 2) * GY : F_XLO + (f - 4) * GX); }

struct Args {
    int n0, n1, n2;                // grid (axis 0 slowest), whole blocks
    int gz, gy, gx;                // blocks per axis; gridDim.x == gz * gy * gx
    int rz, ry, rx;                // XCD regions per axis (rz * ry * rx == 8: workgroup b, which runs on XCD b % 8, takes a block of
                                   // region b % 8), or 1, 1, 1: linear block order, every face written through
    long ss, frame_stride;         // species stride, frame stride (elements)
    void* outbox;                  // [2][blocks][BOX_GRAN] granules, zeroed before the launch
    unsigned* sync;                // device words, zeroed: [0] roll call, [1] abort flag, [2] time-outs
    int* host;                     // host-mapped status (nullable): [0] roll call complete, [3] aborted, [1] step, [2] workgroup
    int nsteps;
    int skip;                      // TIMING EXPERIMENTS ONLY (wrong results): bit 0 = local faces are not handed over, bit 1 = remote ones
    int pause;                     // units of 64 clocks between the publish and the ring request
    unsigned long long timeout_ticks, first_timeout_ticks;
};

struct Task { int z, y, xs; };
__device__ __forceinline__ Task interior_task(int i)
{
    Task t;
    constexpr int PER = (BY - 4) * (XS - 2);
    t.z = 2 + i / PER;
    const int r = i % PER;
    t.y = 2 + r / (XS - 2);
    t.xs = 1 + r % (XS - 2);
    return t;
}
__device__ __forceinline__ Task shell_task(int i)
{
    Task t;
    constexpr int SLAB = 2 * BY * XS;                                    // strips of the two bottom / top planes
    constexpr int MID = BY * XS - (BY - 4) * (XS - 2);                   // shell strips of a middle plane (56)
    if (i < SLAB) { t.z = i / (BY * XS); t.y = (i / XS) % BY; t.xs = i % XS; return t; }
    if (i >= SLAB + (BZ - 4) * MID) { const int k = i - SLAB - (BZ - 4) * MID; t.z = BZ - 2 + k / (BY * XS); t.y = (k / XS) % BY; t.xs = k % XS; return t; }
    const int m = i - SLAB;
    t.z = 2 + m / MID;
    const int w = m % MID;
    if (w < 2 * XS) { t.y = w / XS; t.xs = w % XS; }
    else if (w >= MID - 2 * XS) { const int k = w - (MID - 2 * XS); t.y = BY - 2 + k / XS; t.xs = k % XS; }
    else { const int k = w - 2 * XS; t.y = 2 + k / 2; t.xs = (k & 1) * (XS - 1); }
    return t;
}
// strip number k of the schedule: phase S = shell strips, then interior strips 768..863; phase I = interior strips 0..767.
// Slot s of lane tid is strip s * NT + tid, wave-slot s * NW + wave belongs to phase S iff it is below NS / 64.
__device__ __forceinline__ Task task_k(int k) { return k < NSH ? shell_task(k) : (k < NS ? interior_task(768 + k - NSH) : interior_task(k - NS)); }
template <int NT> __device__ __forceinline__ Task task_of(int s, int tid) { return task_k(s * NT + tid); }
__device__ __forceinline__ unsigned lds_of(const Task& t) { return (unsigned)((t.z + 2) * ZS + (t.y + 2) * YS + OWN0 + 16 * t.xs); }

__device__ __forceinline__ v4f lds4(const unsigned char* p) { return *reinterpret_cast<const v4f*>(p); }
__device__ __forceinline__ v2f lds2(const unsigned char* p) { return *reinterpret_cast<const v2f*>(p); }

// one strip (4 points along x, both species) of one forward step, operands from the LDS window; pi::star's order: centre,
// axis 0 (-2, -1, +1, +2), axis 1, axis 2; coef * lap + react and h + res * dt keep their two roundings (train_3drd.py:123-139)
// `fr` (nullable, wave-uniform): frame of the state being READ -- the strip's own values are stored there on the way.  The frame
// of step t thus leaves during the strips of step t + 1, two stores at a time between arithmetic, instead of as one burst of 64 KB
// per CU that the ring request would queue behind (a CU drains ~12 B per clock to memory: 2.2 us of a step).
typedef __attribute__((address_space(1))) v4f gv4f;
typedef __attribute__((address_space(1))) char gchar;
__device__ __forceinline__ void fwd_strip(const unsigned char* smem, unsigned lo, const float* __restrict__ P, v4f& ou, v4f& ov,
                                          float* fr, long ss, unsigned go)
{
    v4f c[2], lap[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned char* b = smem + s * SP + lo;
        c[s] = lds4(b);
#ifndef R3D_NO_FRAME
        if (fr) *(gv4f*)((gchar*)(fr + s * ss) + go) = c[s];
#endif
        v4f l;
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = P[P_C0] * c[s][i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = t < 2 ? t - 2 : t - 1;
            const v4f nb = lds4(b + k * ZS);
            const float w = P[P_TAPS + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = fma_(w, nb[i], l[i]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = t < 2 ? t - 2 : t - 1;
            const v4f nb = lds4(b + k * YS);
            const float w = P[P_TAPS + 4 + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = fma_(w, nb[i], l[i]);
        }
        const v2f xl = lds2(b - 8), xr = lds2(b + 16);
        const float win[8] = {xl[0], xl[1], c[s][0], c[s][1], c[s][2], c[s][3], xr[0], xr[1]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = t < 2 ? t - 2 : t - 1;
            const float w = P[P_TAPS + 8 + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = fma_(w, win[2 + i + k], l[i]);
        }
        lap[s] = l;
    }
    const float dt = P[P_DT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* cf = P + P_W + 10 * s;
        const float coef = P[P_COEF + s];
        v4f o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float rr = poly_r(cf, c[0][i], c[1][i]);
            const float res = coef * lap[s][i] + rr;
            const float inc = res * dt;
            o[i] = c[s][i] + inc;
        }
        if (s == 0) ou = o; else ov = o;
    }
}

// the face granules of a computed strip into the staging area (payload only, outbox order): two x-adjacent points per granule
__device__ __forceinline__ void stage_strip(unsigned char* smem, const Task& t, const v4f& u, const v4f& v)
{
    const v4f g0 = {u[0], v[0], u[1], v[1]}, g1 = {u[2], v[2], u[3], v[3]};
    if (t.z < 2 || t.z >= BZ - 2) {
        const int r = ((t.z < 2 ? t.z : t.z - (BZ - 2)) * BY + t.y) * (BX / 2) + 2 * t.xs;
        v4f* d = reinterpret_cast<v4f*>(smem + STG0 + ((t.z < 2 ? F_ZLO : F_ZHI) + r) * 16);
        d[0] = g0; d[1] = g1;
    }
    if (t.y < 2 || t.y >= BY - 2) {
        const int r = (t.z * 2 + (t.y < 2 ? t.y : t.y - (BY - 2))) * (BX / 2) + 2 * t.xs;
        v4f* d = reinterpret_cast<v4f*>(smem + STG0 + ((t.y < 2 ? F_YLO : F_YHI) + r) * 16);
        d[0] = g0; d[1] = g1;
    }
    if (t.xs == 0) *reinterpret_cast<v4f*>(smem + STG0 + (F_XLO + t.z * BY + t.y) * 16) = g0;
    if (t.xs == XS - 1) *reinterpret_cast<v4f*>(smem + STG0 + (F_XHI + t.z * BY + t.y) * 16) = g1;
}

// ---- the hand-over: what this lane publishes and gathers (fixed per launch) ----------------------------------------------------
// store / load q of lane tid = box granule g = q * NT + tid (nothing beyond NGRAN): a granule of face face_of(g).  My LOW halo is
// the neighbour's HIGH face and the other way round, at the same granule number inside the face.
template <int NT> struct Ring {
    static constexpr int NQ = Shape<NT>::NQ;
    unsigned gl[NQ];               // ... of the halo pair it fills from load q
    int gsoff[NQ];                 // byte offset of the neighbour's box inside a parity half (wave-uniform per q)
    int gvoff[NQ];                 // byte offset of the granule this lane loads inside that box
    int moff;                      // lanes 0 .. 2 NQ - 1: byte offset inside a parity half of the mask granule of load lane / 2, point lane & 1
    unsigned local;                // bit f: the reader of my face f (= the owner of my halo f) shares this XCD
};

__device__ __forceinline__ unsigned pair_lds(int face, int r, bool halo)
{
    // own point pair of face granule r, or (halo) the halo pair the OPPOSITE face of the neighbour fills
    if (face < 2) {             // z faces: [2 planes][BY][BX / 2]
        const int p = r / (BY * BX / 2), y = (r / (BX / 2)) % BY, xp = r % (BX / 2);
        const int plane = halo ? (face == 0 ? p : BZ + 2 + p) : (face == 0 ? 2 + p : BZ + p);
        return (unsigned)(plane * ZS + (y + 2) * YS + OWN0 + 8 * xp);
    }
    if (face < 4) {             // y faces: [BZ][2 rows][BX / 2]
        const int z = r / BX, rr = (r / (BX / 2)) % 2, xp = r % (BX / 2);
        const int row = halo ? (face == 2 ? rr : BY + 2 + rr) : (face == 2 ? 2 + rr : BY + rr);
        return (unsigned)((z + 2) * ZS + row * YS + OWN0 + 8 * xp);
    }
    const int z = r / BY, y = r % BY;   // x faces: [BZ][BY]
    const int col = halo ? (face == 4 ? OWN0 - 8 : OWN0 + 4 * BX) : (face == 4 ? OWN0 : OWN0 + 4 * (BX - 2));
    return (unsigned)((z + 2) * ZS + (y + 2) * YS + col);
}

template <int NT>
__device__ __forceinline__ void ring_setup(Ring<NT>& R, const Args& a, int bz, int by, int bx, unsigned local)
{
    constexpr int NQ = Shape<NT>::NQ, BOX_GRAN = Shape<NT>::BOX_GRAN;
    const int tid = (int)threadIdx.x;
    auto blk = [&](int z, int y, int x) { return ((z + a.gz) % a.gz * a.gy + (y + a.gy) % a.gy) * a.gx + (x + a.gx) % a.gx; };
    const int nb[6] = {blk(bz - 1, by, bx), blk(bz + 1, by, bx), blk(bz, by - 1, bx), blk(bz, by + 1, bx), blk(bz, by, bx - 1), blk(bz, by, bx + 1)};
    R.local = local;
    R.moff = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = min(q * NT + tid, NGRAN - 1);                     // (lanes beyond the last granule never store / load)
        const int f = face_of(g), r = g - face_base(f);                 // my face this lane publishes AND the halo it fills
        R.gl[q] = pair_lds(f, r, true);
        // the neighbour published what I load as granule gn of ITS box (the opposite face), from lane gn % NT in store gn / NT
        const int gn = face_base(f ^ 1) + r;
        R.gsoff[q] = __builtin_amdgcn_readfirstlane(nb[f] * BOX_GRAN * 16);
        R.gvoff[q] = gn * 16;
        if (tid % WAVE / 2 == q) R.moff = (nb[f] * BOX_GRAN + F_MASK + ((gn % NT / WAVE) * NQ + gn / NT) * 2 + (tid & 1)) * 16;
    }
}

__device__ __forceinline__ unsigned tag_bit(unsigned ep) { return ((ep + 1u) >> 1) & 1u; }

// publish the faces in the staging area: NQ data granules per lane, then the wave's mask granules
template <int NT>
__device__ __forceinline__ void publish(const unsigned char* smem, const Ring<NT>& R, __amdgpu_buffer_rsrc_t mine, unsigned ep, unsigned skipf)
{
    constexpr int NQ = Shape<NT>::NQ;
    const int tid = (int)threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    const unsigned tb = tag_bit(ep);
    unsigned mlo = 0, mhi = 0;                                          // lane 2 q + p ends up holding the ballot of (q, p)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int g = q * NT + tid;
        if (q * NT + (tid / WAVE) * WAVE >= NGRAN) continue;            // wave-uniform: this wave has no granule in store q
        const v4u sg = *reinterpret_cast<const v4u*>(smem + STG0 + g * 16);
        const v2u u = {sg.x, sg.z}, v = {sg.y, sg.w};
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64((v[0] & 1u) != 0u), b1 = __builtin_amdgcn_ballot_w64((v[1] & 1u) != 0u);
        mlo = lane == 2 * q ? (unsigned)b0 : (lane == 2 * q + 1 ? (unsigned)b1 : mlo);
        mhi = lane == 2 * q ? (unsigned)(b0 >> 32) : (lane == 2 * q + 1 ? (unsigned)(b1 >> 32) : mhi);
        const v4u w = {u[0], (v[0] & ~1u) | tb, u[1], (v[1] & ~1u) | tb};
        const int f = face_of(g);
        if ((skipf >> f) & 1u) continue;
        // wave-uniform choice (the cache policy is an immediate): plain = the line stays in this XCD's L2 for a reader under the
        // same L2, sc1 = written through for a reader on another XCD
        if ((R.local >> f) & 1u) __builtin_amdgcn_raw_buffer_store_b128(w, mine, g * 16, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(w, mine, g * 16, 0, /*sc1*/ 16);
    }
    if (lane < 2 * NQ) {
        const v4u m = {mlo, ep, mhi, ep};
        // the masks of one wave serve readers on either kind of XCD: written through (a same-XCD reader's sc1 load finds them too)
        __builtin_amdgcn_raw_buffer_store_b128(m, mine, (F_MASK + wave * NQ * 2 + lane) * 16, 0, /*sc1*/ 16);
    }
}

template <int NT> struct Landing { v4u g[Shape<NT>::NQ]; v4u m; };

template <int NT>
__device__ __forceinline__ void request(Landing<NT>& L, const Ring<NT>& R, __amdgpu_buffer_rsrc_t box, int half, unsigned skipf)
{
    constexpr int NQ = Shape<NT>::NQ;
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q * NT + (tid / WAVE) * WAVE >= NGRAN) continue;
        const int f = face_of(q * NT + tid);
        if (!((skipf >> f) & 1u)) L.g[q] = __builtin_amdgcn_raw_buffer_load_b128(box, R.gvoff[q], half + R.gsoff[q], 16);
    }
    if (tid % WAVE < 2 * NQ) L.m = __builtin_amdgcn_raw_buffer_load_b128(box, R.moff, half, 16);
}

// true once every granule of this wave carries step `want`; asks again for those that do not
template <int NT>
__device__ __forceinline__ bool landed(Landing<NT>& L, const Ring<NT>& R, __amdgpu_buffer_rsrc_t box, int half, unsigned want, unsigned skipf)
{
    constexpr int NQ = Shape<NT>::NQ;
    const int tid = (int)threadIdx.x, lane = tid % WAVE;
    const unsigned tb = tag_bit(want);
    bool ok = true;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q * NT + (tid / WAVE) * WAVE >= NGRAN) continue;
        const int f = face_of(q * NT + tid);
        const bool okq = ((skipf >> f) & 1u) || ((L.g[q].y & 1u) == tb && (L.g[q].w & 1u) == tb);
        if (!okq) L.g[q] = __builtin_amdgcn_raw_buffer_load_b128(box, R.gvoff[q], half + R.gsoff[q], 16);
        ok &= okq;
    }
    if (lane < 2 * NQ && (lane / 2) * NT + (tid / WAVE) * WAVE < NGRAN) {
        const bool okm = L.m.y == want && L.m.w == want;
        if (!okm) L.m = __builtin_amdgcn_raw_buffer_load_b128(box, R.moff, half, 16);
        ok &= okm;
    }
    return __all(ok);
}

// the landed ring into the LDS window, the displaced bits put back
template <int NT>
__device__ __forceinline__ void unpack(unsigned char* smem, const Landing<NT>& L, const Ring<NT>& R, unsigned skipf)
{
    constexpr int NQ = Shape<NT>::NQ;
    const int tid = (int)threadIdx.x, lane = tid % WAVE;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q * NT + (tid / WAVE) * WAVE >= NGRAN) continue;
        const int f = face_of(q * NT + tid);
        unsigned bit[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)L.m.x, 2 * q + p), hi = (unsigned)__builtin_amdgcn_readlane((int)L.m.z, 2 * q + p);
            bit[p] = ((lane < 32 ? lo : hi) >> (lane & 31)) & 1u;
        }
        if ((skipf >> f) & 1u) continue;
        *reinterpret_cast<v2u*>(smem + R.gl[q]) = v2u{L.g[q].x, L.g[q].z};
        *reinterpret_cast<v2u*>(smem + SP + R.gl[q]) = v2u{(L.g[q].y & ~1u) | bit[0], (L.g[q].w & ~1u) | bit[1]};
    }
}

__device__ __forceinline__ unsigned long long ticks() { return wall_clock64(); }       // 100 MHz

#ifndef PI_R3D_STAMPS
#define R3D_STAMP(k)
#else
#define R3D_STAMP(k) do { if (stamps && blockIdx.x == PI_R3D_STAMPS && threadIdx.x == 0 && t == 20) stamps[k] = wall_clock64(); } while (0)
#endif

// block coordinates of workgroup b and which of its faces have their reader under the same L2
__device__ __forceinline__ unsigned locate_block(const Args& a, int b, int& bz, int& by, int& bx)
{
    const int dz = a.gz / a.rz, dy = a.gy / a.ry, dx = a.gx / a.rx;
    unsigned local = 0;
    if (a.rz * a.ry * a.rx == NXCD) {
        const int x = b % NXCD, i = b / NXCD;
        const int ix = i % dx, iy = (i / dx) % dy, iz = i / (dx * dy);
        bx = (x % a.rx) * dx + ix; by = ((x / a.rx) % a.ry) * dy + iy; bz = (x / (a.rx * a.ry)) * dz + iz;
        // a neighbour is in the same region unless this block sits on the region's face (a region that spans the whole axis
        // wraps onto itself)
        local = (unsigned)(iz > 0 || a.rz == 1) | (unsigned)(iz < dz - 1 || a.rz == 1) << 1 | (unsigned)(iy > 0 || a.ry == 1) << 2 |
                (unsigned)(iy < dy - 1 || a.ry == 1) << 3 | (unsigned)(ix > 0 || a.rx == 1) << 4 | (unsigned)(ix < dx - 1 || a.rx == 1) << 5;
    } else {
        bx = b % a.gx; by = (b / a.gx) % a.gy; bz = b / (a.gx * a.gy);
    }
    return local;
}

// forward rollout: frames[0] is the initial state (read), frames[1 .. nsteps] are written.
// Per step:  S: the strips that read the halo, their faces staged | publish | I: the interior strips, the ring requested after the
// first of them | write-back | ring lands, halo unpacked -- publish to need is the whole of I + write-back (~2.5 us).
template <int NT>
__global__ void __launch_bounds__(NT, 1)
pi_fwd3d_resident_kernel(float* __restrict__ frames, const float* __restrict__ P, Args a, unsigned long long* stamps)
{
    using S = Shape<NT>;
    constexpr int SLOTS = S::SLOTS, NW = S::NW, BOX_BYTES = S::BOX_BYTES;
    constexpr int WS = NS / WAVE;                                       // wave-slots of phase S
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* smem = smem_raw + 8;                                 // the window (see ROWB)
    int* wg_abort = reinterpret_cast<int*>(smem + STG0 + NGRAN * 16);
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int nblk = a.gz * a.gy * a.gx;
    const int b = (int)blockIdx.x;
    int bx, by, bz;
    const unsigned local = locate_block(a, b, bz, by, bx);
    const unsigned skipf = ((a.skip & 1) ? local : 0u) | ((a.skip & 2) ? (~local & 63u) : 0u);
    const int me = (bz * a.gy + by) * a.gx + bx;                        // outbox slot = block number (not workgroup number)
    if (tid == 0) {
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)nblk && a.host) __hip_atomic_store(a.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int half_bytes = nblk * BOX_BYTES;
    const __amdgpu_buffer_rsrc_t box = __builtin_amdgcn_make_buffer_rsrc(a.outbox, 0, 2 * half_bytes, 0x00020000);

    // ---- fixed geometry of this lane -------------------------------------------------------------------------------------
    unsigned lo[SLOTS], go[SLOTS], tk[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const Task t = task_of<NT>(s, tid);
        lo[s] = lds_of(t);
        tk[s] = (unsigned)t.z | (unsigned)t.y << 8 | (unsigned)t.xs << 16;
        go[s] = (unsigned)((((bz * BZ + t.z) * a.n1 + by * BY + t.y) * a.n2 + bx * BX + 4 * t.xs) * 4);
    }
    Ring<NT> R;
    ring_setup<NT>(R, a, bz, by, bx, local);

    // ---- the window of frame 0: own block and halo straight from memory ---------------------------------------------------
    for (int i = tid; i < 2 * (BZ + 4) * (BY + 4) * (BX + 4); i += NT) {
        const int x = i % (BX + 4), y = (i / (BX + 4)) % (BY + 4), z = (i / ((BX + 4) * (BY + 4))) % (BZ + 4), s = i / ((BX + 4) * (BY + 4) * (BZ + 4));
        const int g0 = (bz * BZ + z - 2 + a.n0) % a.n0, g1 = (by * BY + y - 2 + a.n1) % a.n1, g2 = (bx * BX + x - 2 + a.n2) % a.n2;
        const float v = frames[s * a.ss + ((long)g0 * a.n1 + g1) * a.n2 + g2];
        *reinterpret_cast<float*>(smem + s * SP + z * ZS + y * YS + OWN0 - 8 + 4 * x) = v;
    }
    __syncthreads();

    Landing<NT> L;
    bool failed = false;
    for (int t = 0; t < a.nsteps; ++t) {
        const unsigned ep = (unsigned)t + 1u;                           // the state this step produces
        const bool more = t + 1 < a.nsteps, ring = more && !(a.skip & 4);
        float* fr = t > 0 ? frames + (long)t * a.frame_stride : nullptr;      // frame 0 is the caller's
        v4f ou[SLOTS], ov[SLOTS];
        R3D_STAMP(0);
        // ---- phase S: the strips that read the halo; what the neighbours wait for goes to the staging area ----
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s * NW + wave < WS) {                                   // wave-uniform
                fwd_strip(smem, lo[s], P, ou[s], ov[s], fr, a.ss, go[s]);
                const Task tt = {(int)(tk[s] & 255u), (int)(tk[s] >> 8 & 255u), (int)(tk[s] >> 16)};
                stage_strip(smem, tt, ou[s], ov[s]);
            }
        R3D_STAMP(1);
        lds_barrier();                                                  // faces staged; nobody reads the halo of state t any more
        R3D_STAMP(2);
        if (ring) {
            const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(
                static_cast<char*>(a.outbox) + ((size_t)(ep & 1u) * (size_t)half_bytes + (size_t)me * BOX_BYTES), 0, BOX_BYTES, 0x00020000);
            publish<NT>(smem, R, mine, ep, skipf);
        }
        R3D_STAMP(3);
        // ---- phase I: interior strips; the ring of state t + 1 is requested after the first of them ----
        bool asked = !ring;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s * NW + wave >= WS) {
                fwd_strip(smem, lo[s], P, ou[s], ov[s], fr, a.ss, go[s]);
                if (!asked) {
                    for (int i = 0; i < a.pause; ++i) __builtin_amdgcn_s_sleep(1);
                    request<NT>(L, R, box, (int)(ep & 1u) * half_bytes, skipf);
                    asked = true;
                }
            }
        R3D_STAMP(4);
        lds_barrier();                                                  // every read of state t is done
        R3D_STAMP(5);
        // ---- state t + 1 into the window ----
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            *reinterpret_cast<v4f*>(smem + lo[s]) = ou[s];
            *reinterpret_cast<v4f*>(smem + SP + lo[s]) = ov[s];
        }
        if (!more) {                                                    // the last frame has no next step to carry it
#ifndef R3D_NO_FRAME
            float* out = frames + (long)(t + 1) * a.frame_stride;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                *(gv4f*)((gchar*)out + go[s]) = ou[s];
                *(gv4f*)((gchar*)(out + a.ss) + go[s]) = ov[s];
            }
#endif
            break;
        }
        R3D_STAMP(6);
        // ---- the halo of state t + 1 lands ----
        if (ring) {
            const int rd = (int)(ep & 1u) * half_bytes;
            const unsigned long long t0 = ticks();
            const unsigned long long bound = t == 0 ? a.first_timeout_ticks : a.timeout_ticks;
            while (!landed<NT>(L, R, box, rd, ep, skipf)) {
                if (ticks() - t0 > bound || __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed) {
                if (tid % WAVE == 0) *wg_abort = 1;
            } else {
                unpack<NT>(smem, L, R, skipf);
            }
        }
        R3D_STAMP(7);
        lds_barrier();
        if (*wg_abort) {
            // ABORT: the host learns it from its status slot and runs the launch-per-step kernels instead; frames already
            // written are the correct ones, nothing of the ones not yet written is
            if (tid == 0) {
                __hip_atomic_fetch_add(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && a.host) {
                    __hip_atomic_store(a.host + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(a.host + 2, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(a.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// REVERSE SWEEP, resident: the adjoint state G lives in the LDS window for the whole sweep; a step reads only its pointwise
// operands from memory (the state h_{t-1} and, where that frame carries a gradient, dL/dtraj[t-1]: 16 + 16 instead of the brick
// sweep's 48 + 16 bytes per point and step), always one step ahead of their use, and nothing is written until dL/dh0.
// Arithmetic = pi_adj3d_brick_kernel's (pre-contracted float32 block, fused moments): the adjoint state is bit-identical, the 22
// sums differ in summation order only.
// ---------------------------------------------------------------------------------------------------------------------------
struct AdjArgs {
    const float* traj;             // state trajectory; frame f at traj + f * frame_stride
    const float* gtraj;            // dL/dtraj, same layout (only frames whose bit is set in `frames` are read)
    const float* gtop;             // the adjoint state the sweep starts from: dL/dh_{t_top} (a whole frame)
    float* gout;                   // dL/dh_{t_top - nsteps}
    double* partials;              // [blocks][np] partial rows, added to
    int np, t_top;
    unsigned frames[128];          // bit f: frame f carries a gradient (f < 4096)
};

typedef float av2 __attribute__((ext_vector_type(2)));
struct AdjOps { v4f u, v; };

__device__ __forceinline__ void adj_load(AdjOps& o, const float* hfr, long ss, unsigned go)
{
    o.u = *(const gv4f*)((const gchar*)hfr + go);
    o.v = *(const gv4f*)((const gchar*)(hfr + ss) + go);
}

// one strip of one adjoint step: Gp = G + coef * (dt * LapT(G)) + dt * J_react(h)^T G (+ inj); pi_adj3d_brick_kernel's order
// `jfr` (nullable, wave-uniform): the frame of dL/dtraj this step injects; its strip is requested at the top and used at the bottom
__device__ __forceinline__ void adj_strip_whole(const unsigned char* smem, unsigned lo, const float* __restrict__ P, const AdjOps& o,
                                          const float* jfr, long ss, unsigned go, v4f& ou, v4f& ov, float (&mom)[2][10], double (&lane_c)[2])
{
    v4f ju, jv;
    if (jfr) {
        ju = *(const gv4f*)((const gchar*)jfr + go);
        jv = *(const gv4f*)((const gchar*)(jfr + ss) + go);
    }
    v4f gc[2], dl[2];
    const float dt = P[P_DT];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned char* b = smem + s * SP + lo;
        gc[s] = lds4(b);
        v4f l;
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = P[P_C0] * gc[s][i];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = -(t < 2 ? t - 2 : t - 1);                      // the transposed stencil: taps at mirrored offsets
            const v4f nb = lds4(b + k * ZS);
            const float w = P[P_TAPS + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = fma_(w, nb[i], l[i]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = -(t < 2 ? t - 2 : t - 1);
            const v4f nb = lds4(b + k * YS);
            const float w = P[P_TAPS + 4 + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = fma_(w, nb[i], l[i]);
        }
        const v2f xl = lds2(b - 8), xr = lds2(b + 16);
        const float win[8] = {xl[0], xl[1], gc[s][0], gc[s][1], gc[s][2], gc[s][3], xr[0], xr[1]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = -(t < 2 ? t - 2 : t - 1);
            const float w = P[P_TAPS + 8 + t];
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = fma_(w, win[2 + i + k], l[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dl[s][i] = l[i] * dt;
    }
    v4f du = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* c = P + P_W + 10 * s;
        const v4f& hs = s == 0 ? o.u : o.v;
        double acc_c = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gr = gc[s][i] * dt;
            acc_c += (double)(dl[s][i] * hs[i]);
            float ru, rv;
            poly_dr(c, o.u[i], o.v[i], ru, rv);
            du[i] = fma_(gr, ru, du[i]);
            dv[i] = fma_(gr, rv, dv[i]);
        }
        lane_c[s] += acc_c;
    }
    // the 20 coefficient moments: one float32 sum per lane and moment (pairs would issue half the instructions and hold twice the
    // registers: with them the sweep spilled 64 registers per lane to scratch -- 22.5 us per step)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uu = o.u[i], vv = o.v[i];
        const float u2 = uu * uu, uv = uu * vv, v2 = vv * vv;
        const float u3 = u2 * uu, u2v = u2 * vv, uv2 = uu * v2, v3 = v2 * vv;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float gr = gc[s][i] * dt;
            float (&acc)[10] = mom[s];
            acc[0] += gr;
            acc[1] = fma_(gr, uu, acc[1]); acc[2] = fma_(gr, vv, acc[2]);
            acc[3] = fma_(gr, u2, acc[3]); acc[4] = fma_(gr, uv, acc[4]); acc[5] = fma_(gr, v2, acc[5]);
            acc[6] = fma_(gr, u3, acc[6]); acc[7] = fma_(gr, u2v, acc[7]); acc[8] = fma_(gr, uv2, acc[8]); acc[9] = fma_(gr, v3, acc[9]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float tu = P[P_COEF + 0] * dl[0][i] + du[i];
        const float tv = P[P_COEF + 1] * dl[1][i] + dv[i];
        ou[i] = gc[0][i] + tu;
        ov[i] = gc[1][i] + tv;
    }
    if (jfr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ou[i] += ju[i]; ov[i] += jv[i]; }
    }
}

// ---- the strip as TWO HALF-STRIPS on explicit pairs (round 6, late) ----
// The four-point strip above (kept below as adj_strip_whole, R3D_ADJ_HALVES = 0) is scalar source that hipcc's SLP vectoriser packs
// by itself: 418 vector instructions per strip where ~200 packed ones would do (118 v_mov building operand pairs), 256 registers and
// 128 bytes of scratch.  Written as two halves of two x-adjacent points each on 2-vectors -- every LDS read is an 8-byte pair that
// IS the operand of a v_pk_fma_f32 -- the same IEEE operations in the same order per point (adjoint state bit-identical) need no
// pair building, and a half's temporaries are half a strip's.  R3D_MOM_PAIRS: the 20 moment sums as pairs (40 registers, one
// packed FMA per moment and half) or as scalars (20 registers, two FMAs).
// MEASURED AND NOT ADOPTED (tools/res3d, 128^3, dense injection, us per step): whole strips 16.3 | halves 21.9 (pairs) / 22.3
// (scalar sums) | halves without the scheduling barrier between them 20.0 | whole strips with the parameter block re-read per strip
// (R3D_P_RELOAD) 24.1.  All bit-identical.  The halves do drop ~50 v_mov per strip (264 v_mov_b32 against 590 + 100 v_pk_mov in the
// kernel) but the allocator answers with MORE spills (216 bytes of scratch against 128, 136 scalar registers spilled to lanes
// against 122): the kernel's wall is the 32 held outputs + 20 sums + operand ring next to ~50 live constants, not the strip's shape.
#ifndef R3D_ADJ_HALVES
#define R3D_ADJ_HALVES 0
#endif
#ifndef R3D_MOM_PAIRS
#define R3D_MOM_PAIRS 1
#endif
#ifndef R3D_HALF_BARRIER
#define R3D_HALF_BARRIER 1
#endif
#if R3D_ADJ_HALVES && R3D_MOM_PAIRS
typedef v2f MomSum;
__device__ __forceinline__ float mom_total(v2f a) { return a.x + a.y; }
__device__ __forceinline__ v2f mom_zero() { return v2f{0.f, 0.f}; }
#else
typedef float MomSum;
__device__ __forceinline__ float mom_total(float a) { return a; }
__device__ __forceinline__ float mom_zero() { return 0.f; }
#endif
__device__ __forceinline__ v2f bc2(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void mom_add(v2f& acc, v2f gr, v2f phi) { acc = fma2(gr, phi, acc); }
__device__ __forceinline__ void mom_add(float& acc, v2f gr, v2f phi) { acc = fma_(gr.x, phi.x, acc); acc = fma_(gr.y, phi.y, acc); }
__device__ __forceinline__ void mom_add1(v2f& acc, v2f gr) { acc += gr; }
__device__ __forceinline__ void mom_add1(float& acc, v2f gr) { acc += gr.x; acc += gr.y; }
// pi::poly_dr on pairs (same operations in the same order)
__device__ __forceinline__ void poly_dr2(const float* __restrict__ c, v2f u, v2f v, v2f& ru, v2f& rv)
{
    const v2f A1 = fma2(v, fma2(v, bc2(c[8]), bc2(c[4])), bc2(c[1]));
    const v2f A2x2 = fma2(v, bc2(2.f * c[7]), bc2(2.f * c[3]));
    ru = fma2(u, fma2(u, bc2(3.f * c[6]), A2x2), A1);
    const v2f B0 = fma2(v, fma2(v, bc2(3.f * c[9]), bc2(2.f * c[5])), bc2(c[2]));
    const v2f B1 = fma2(v, bc2(2.f * c[8]), bc2(c[4]));
    rv = fma2(u, fma2(u, bc2(c[7]), B1), B0);
}
// the two points at window byte offset `lo` (8-byte aligned); hu / hv: their state operands
template <typename MOM>
__device__ __forceinline__ void adj_half(const unsigned char* smem, unsigned lo, const float* __restrict__ P, v2f hu, v2f hv,
                                         v2f& ou, v2f& ov, MOM (&mom)[2][10], double (&lane_c)[2])
{
    v2f gc[2], dl[2];
    const v2f dt = bc2(P[P_DT]);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned char* b = smem + s * SP + lo;
        const v2f W[3] = {lds2(b - 8), lds2(b), lds2(b + 8)};
        gc[s] = W[1];
        v2f l = bc2(P[P_C0]) * W[1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = -(t < 2 ? t - 2 : t - 1);                      // the transposed stencil: taps at mirrored offsets
            l = fma2(bc2(P[P_TAPS + t]), lds2(b + k * ZS), l);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = -(t < 2 ? t - 2 : t - 1);
            l = fma2(bc2(P[P_TAPS + 4 + t]), lds2(b + k * YS), l);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = -(t < 2 ? t - 2 : t - 1);
            const int idx = 2 + k;                                       // element of the six-point row window W
            const v2f nb = (idx & 1) ? v2f{W[idx / 2].y, W[idx / 2 + 1].x} : W[idx / 2];
            l = fma2(bc2(P[P_TAPS + 8 + t]), nb, l);
        }
        dl[s] = l * dt;
    }
    v2f du = {0.f, 0.f}, dv = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* c = P + P_W + 10 * s;
        const v2f gr = gc[s] * dt;
        const v2f ch = dl[s] * (s == 0 ? hu : hv);
        double acc_c = 0.0;
        acc_c += (double)ch.x;
        acc_c += (double)ch.y;
        lane_c[s] += acc_c;
        v2f ru, rv;
        poly_dr2(c, hu, hv, ru, rv);
        du = fma2(gr, ru, du);
        dv = fma2(gr, rv, dv);
    }
    {
        const v2f u2 = hu * hu, uv = hu * hv, v2 = hv * hv;
        const v2f u3 = u2 * hu, u2v = u2 * hv, uv2 = hu * v2, v3 = v2 * hv;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const v2f gr = gc[s] * dt;
            MOM (&acc)[10] = mom[s];
            mom_add1(acc[0], gr);
            mom_add(acc[1], gr, hu); mom_add(acc[2], gr, hv);
            mom_add(acc[3], gr, u2); mom_add(acc[4], gr, uv); mom_add(acc[5], gr, v2);
            mom_add(acc[6], gr, u3); mom_add(acc[7], gr, u2v); mom_add(acc[8], gr, uv2); mom_add(acc[9], gr, v3);
        }
    }
    const v2f tu = bc2(P[P_COEF + 0]) * dl[0] + du;
    const v2f tv = bc2(P[P_COEF + 1]) * dl[1] + dv;
    ou = gc[0] + tu;
    ov = gc[1] + tv;
}

// one strip of one adjoint step: Gp = G + coef * (dt * LapT(G)) + dt * J_react(h)^T G (+ inj); pi_adj3d_brick_kernel's order
// `jfr` (nullable, wave-uniform): the frame of dL/dtraj this step injects; its strip is requested at the top and used at the bottom
template <typename MOM>
__device__ __forceinline__ void adj_strip(const unsigned char* smem, unsigned lo, const float* __restrict__ P, const AdjOps& o,
                                          const float* jfr, long ss, unsigned go, v4f& ou, v4f& ov, MOM (&mom)[2][10], double (&lane_c)[2])
{
    v4f ju, jv;
    if (jfr) {
        ju = *(const gv4f*)((const gchar*)jfr + go);
        jv = *(const gv4f*)((const gchar*)(jfr + ss) + go);
    }
    v2f au, av, bu, bv;
    adj_half(smem, lo, P, v2f{o.u[0], o.u[1]}, v2f{o.v[0], o.v[1]}, au, av, mom, lane_c);
#if R3D_HALF_BARRIER
    __builtin_amdgcn_sched_barrier(0);                                  // one half at a time (interleaved, their temporaries add up)
#endif
    adj_half(smem, lo + 8, P, v2f{o.u[2], o.u[3]}, v2f{o.v[2], o.v[3]}, bu, bv, mom, lane_c);
    ou = v4f{au.x, au.y, bu.x, bu.y};
    ov = v4f{av.x, av.y, bv.x, bv.y};
    if (jfr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ou[i] += ju[i]; ov[i] += jv[i]; }
    }
}

// R3D_P_RELOAD: the parameter block is read again for every strip (scalar loads from the constant cache) instead of living in ~50
// scalar registers for the whole sweep, from where the allocator spilled it to vector lanes
#ifndef R3D_P_RELOAD
#define R3D_P_RELOAD 0
#endif
__device__ __forceinline__ const float* r3d_launder(const float* p) { asm volatile("" : "+s"(p)); return p; }
#if R3D_P_RELOAD
#define R3D_P(P) r3d_launder(P)
#else
#define R3D_P(P) (P)
#endif
#if R3D_ADJ_HALVES
#define R3D_ADJ_STRIP adj_strip
#else
#define R3D_ADJ_STRIP adj_strip_whole
#endif
template <int NT>
__global__ void __launch_bounds__(NT, 1)
pi_adj3d_resident_kernel(const float* __restrict__ P, Args a, AdjArgs aa, unsigned long long* stamps)
{
    using S = Shape<NT>;
    constexpr int SLOTS = S::SLOTS, NW = S::NW, BOX_BYTES = S::BOX_BYTES;
    constexpr int WS = NS / WAVE;
    constexpr int FOLD = 8;                                             // steps between folds of the float32 moment sums into doubles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* smem = smem_raw + 8;
    int* wg_abort = reinterpret_cast<int*>(smem + STG0 + NGRAN * 16);
    double* msum = reinterpret_cast<double*>(smem + STG0 + NGRAN * 16 + 56);    // [NW][20] doubles (8-byte aligned: smem is 8 mod 16)
    const int tid = (int)threadIdx.x, lane = tid % WAVE;
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int nblk = a.gz * a.gy * a.gx;
    const int b = (int)blockIdx.x;
    int bx, by, bz;
    const unsigned local = locate_block(a, b, bz, by, bx);
    const unsigned skipf = ((a.skip & 1) ? local : 0u) | ((a.skip & 2) ? (~local & 63u) : 0u);
    const int me = (bz * a.gy + by) * a.gx + bx;
    if (tid == 0) {
        *wg_abort = 0;
        const unsigned n = __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (n == (unsigned)nblk && a.host) __hip_atomic_store(a.host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane < 20) msum[wave * 20 + lane] = 0.0;
    const int half_bytes = nblk * BOX_BYTES;
    const __amdgpu_buffer_rsrc_t box = __builtin_amdgcn_make_buffer_rsrc(a.outbox, 0, 2 * half_bytes, 0x00020000);

    // per strip only its packed coordinates are kept; LDS address and frame offset are a few integer operations each time (the
    // sweep holds 32 outputs, 40 moment sums, 32 operands in flight and 24 words of landing ring per lane: registers are what it lacks)
    unsigned tk[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const Task t = task_of<NT>(s, tid);
        tk[s] = (unsigned)t.z | (unsigned)t.y << 8 | (unsigned)t.xs << 16;
    }
    const unsigned gbase = (unsigned)((((bz * BZ) * a.n1 + by * BY) * a.n2 + bx * BX) * 4);
    auto task = [&](int s) { return Task{(int)(tk[s] & 255u), (int)(tk[s] >> 8 & 255u), (int)(tk[s] >> 16)}; };
    auto lo_of = [&](int s) { return lds_of(task(s)); };
    auto go_of = [&](int s) { const Task t = task(s); return gbase + (unsigned)(((t.z * a.n1 + t.y) * a.n2 + 4 * t.xs) * 4); };
    Ring<NT> R;
    ring_setup<NT>(R, a, bz, by, bx, local);

    auto has = [&](int f) { return ((aa.frames[f >> 5] >> (f & 31)) & 1u) != 0u; };
    // operands of the first step: requested before the window is filled
    // (a ring of two: strip s reads ops[s & 1] and, done, asks for the operands of the strip two further on -- all four strips'
    // operands a whole step ahead cost 64 registers and the kernel 396 bytes of scratch)
    static_assert(SLOTS % 2 == 0, "the operand ring alternates two buffers");
    AdjOps ops[2];
    {
        const int f = aa.t_top - 1;
        const float* hfr = aa.traj + (long)f * a.frame_stride;
        adj_load(ops[0], hfr, a.ss, go_of(0));
        adj_load(ops[1], hfr, a.ss, go_of(1));
    }
    // the window of the top frame's adjoint state: own block and halo straight from memory
    for (int i = tid; i < 2 * (BZ + 4) * (BY + 4) * (BX + 4); i += NT) {
        const int x = i % (BX + 4), y = (i / (BX + 4)) % (BY + 4), z = (i / ((BX + 4) * (BY + 4))) % (BZ + 4), s = i / ((BX + 4) * (BY + 4) * (BZ + 4));
        const int g0 = (bz * BZ + z - 2 + a.n0) % a.n0, g1 = (by * BY + y - 2 + a.n1) % a.n1, g2 = (bx * BX + x - 2 + a.n2) % a.n2;
        const float v = aa.gtop[s * a.ss + ((long)g0 * a.n1 + g1) * a.n2 + g2];
        *reinterpret_cast<float*>(smem + s * SP + z * ZS + y * YS + OWN0 - 8 + 4 * x) = v;
    }
    __syncthreads();

    MomSum mom[2][10];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 10; ++m) mom[s][m] = mom_zero();
    double lane_c[2] = {0.0, 0.0};
    auto fold = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                const float tot = wave_sum_to_last(mom_total(mom[s][m]));
                if (lane == REDUCE_LANE) msum[wave * 20 + 10 * s + m] += (double)tot;
                mom[s][m] = mom_zero();
            }
    };

    Landing<NT> L;
    bool failed = false;
    for (int k = 0; k < a.nsteps; ++k) {
        const int t = k;                                                 // (the stamps' name for the step)
        const unsigned ep = (unsigned)k + 1u;
        const bool more = k + 1 < a.nsteps, ring = more && !(a.skip & 4);
        const int f = aa.t_top - 1 - k;                                 // this step: G_{f + 1} -> G_f with the operands of frame f
        const bool inj = has(f);
        const float* hcu = aa.traj + (long)f * a.frame_stride;
        const float* jcu = inj ? aa.gtraj + (long)f * a.frame_stride : nullptr;
        const float* hnx = more ? aa.traj + (long)(f - 1) * a.frame_stride : nullptr;
        auto next_ops = [&](int s) {                                    // strip s is done: its buffer takes the state of strip s + 2
            if (s + 2 < SLOTS) adj_load(ops[s & 1], hcu, a.ss, go_of(s + 2));
            else if (more) adj_load(ops[s & 1], hnx, a.ss, go_of(s + 2 - SLOTS));
        };
        v4f ou[SLOTS], ov[SLOTS];
        R3D_STAMP(0);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s * NW + wave < WS) {
                R3D_ADJ_STRIP(smem, lo_of(s), R3D_P(P), ops[s & 1], jcu, a.ss, go_of(s), ou[s], ov[s], mom, lane_c);
                next_ops(s);
                __builtin_amdgcn_sched_barrier(0);                      // one strip at a time: interleaved strips spill
                stage_strip(smem, task(s), ou[s], ov[s]);
            }
        R3D_STAMP(1);
        lds_barrier();
        R3D_STAMP(2);
        if (ring) {
            const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(
                static_cast<char*>(a.outbox) + ((size_t)(ep & 1u) * (size_t)half_bytes + (size_t)me * BOX_BYTES), 0, BOX_BYTES, 0x00020000);
            publish<NT>(smem, R, mine, ep, skipf);
        }
        R3D_STAMP(3);
        bool asked = !ring;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s * NW + wave >= WS) {
                R3D_ADJ_STRIP(smem, lo_of(s), R3D_P(P), ops[s & 1], jcu, a.ss, go_of(s), ou[s], ov[s], mom, lane_c);
                next_ops(s);
                __builtin_amdgcn_sched_barrier(0);                      // one strip at a time: interleaved strips spill
            }
        R3D_STAMP(4);
        if (!asked) {                                                   // (after the strips: the landing ring is 24 registers the strips need)
            for (int i = 0; i < a.pause; ++i) __builtin_amdgcn_s_sleep(1);
            request<NT>(L, R, box, (int)(ep & 1u) * half_bytes, skipf);
        }
        if ((k % FOLD) == FOLD - 1 || !more) fold();
        lds_barrier();
        R3D_STAMP(5);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            *reinterpret_cast<v4f*>(smem + lo_of(s)) = ou[s];
            *reinterpret_cast<v4f*>(smem + SP + lo_of(s)) = ov[s];
        }
        if (!more) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                *(gv4f*)((gchar*)aa.gout + go_of(s)) = ou[s];
                *(gv4f*)((gchar*)(aa.gout + a.ss) + go_of(s)) = ov[s];
            }
            break;
        }
        R3D_STAMP(6);
        if (ring) {
            const int rd = (int)(ep & 1u) * half_bytes;
            const unsigned long long t0 = ticks();
            const unsigned long long bound = k == 0 ? a.first_timeout_ticks : a.timeout_ticks;
            while (!landed<NT>(L, R, box, rd, ep, skipf)) {
                if (ticks() - t0 > bound || __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (failed) {
                if (tid % WAVE == 0) *wg_abort = 1;
            } else {
                unpack<NT>(smem, L, R, skipf);
            }
        }
        R3D_STAMP(7);
        lds_barrier();
        if (*wg_abort) {
            // ABORT: neither dL/dh0 nor a partial row is written; the host runs the launch-per-step sweep instead
            if (tid == 0) {
                __hip_atomic_fetch_add(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_exchange(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && a.host) {
                    __hip_atomic_store(a.host + 1, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(a.host + 2, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(a.host + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
    // ---- once per sweep: the 22 sums of this block into its partial row ----
    double* csum = msum + NW * 20;                                      // [NW][2]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const double r = wave_sum_to_last(lane_c[s]);
        if (lane == REDUCE_LANE) csum[wave * 2 + s] = r;
    }
    __syncthreads();
    if (tid < 22) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += tid < 2 ? csum[w * 2 + tid] : msum[w * 20 + tid - 2];
        const int slot = tid < 2 ? P_COEF + tid : P_W + tid - 2;
        aa.partials[(long)b * aa.np + slot] += tot;
    }
}

}  // namespace r3d
}  // namespace pi
